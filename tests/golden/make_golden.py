#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container, where the read-only reference checkout is mounted at
/root/reference.  It imports the reference's pure-numpy host code under permissive stub modules
(`wgpu`, `pynbody` are not installed) and records *data only*:

  * inputs:   TestDataLoader arrays (reference src/topsy/loader.py:241-332)
  * cameras:  SPH._get_transform_params outputs (reference src/topsy/sph.py:268-299)
  * LUTs:     Colormap._generate_mapping_rgba_f32 (reference src/topsy/colormap/implementation.py:235-238)
  * params:   Colormap._update_parameter_buffer / _autorange_using_values outputs (:381-453)
  * driver:   RenderProgression(+WithCells) block sequences (reference src/topsy/progressive_render.py)
  * cells:    CellLayout.from_positions outputs (reference src/topsy/cell_layout.py:63-113)
  * bands:    PynbodyDataInMemory.get_rgb_masses outputs on a stub snapshot of seeded SSP band magnitudes
              (reference src/topsy/loader.py:112-121); `--bands-only` regenerates just this fixture
  * KATs:     the literal expected-output arrays held by the reference's own tests
              (tests/test_render_output.py, tests/test_colormap.py), extracted with `ast`.

Nothing from the reference's source text is copied; the .npz/.json files hold numbers.
The GPU box never runs this script (no /root/reference there).
"""
import ast
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        s = _Stub(self.__name__ + "." + name)
        setattr(self, name, s)
        return s

    def __call__(self, *a, **k):
        return _Stub("call")


def _import_reference():
    for n in ["wgpu", "pynbody", "pynbody.filt", "pynbody.filt.geometry_selection"]:
        sys.modules[n] = _Stub(n)
    sys.path.insert(0, os.path.join(REF, "src"))


def _literals_from_test(path, wanted):
    """Return {func_name.var_name: np.ndarray} for list literals assigned inside test functions."""
    tree = ast.parse(open(path).read())
    out = {}
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if not isinstance(node, ast.Assign) or len(node.targets) != 1:
                continue
            tgt = node.targets[0]
            if not isinstance(tgt, ast.Name) or tgt.id not in wanted:
                continue
            val = node.value
            if isinstance(val, ast.Call):  # np.array([...], dtype=...)
                val = val.args[0]
            if not isinstance(val, ast.List):
                continue
            out[f"{fn.name}.{tgt.id}"] = np.array(ast.literal_eval(val))
    return out


class _RecordingQueue:
    def __init__(self):
        self.last = None

    def write_buffer(self, buf, offset, data):
        self.last = np.array(data).copy()

    def write_texture(self, *a, **k):
        pass


class _RecordingDevice:
    def __init__(self):
        self.queue = _RecordingQueue()

    def __getattr__(self, name):
        return lambda *a, **k: _Stub("obj")


def make_band_fixture():
    """SSP band magnitudes -> rgb masses through the reference's PynbodyDataInMemory.get_rgb_masses
    (reference src/topsy/loader.py:112-121) on a stub snapshot: seeded I/V/U magnitudes (float64, with NaNs, +-inf
    and extreme values) and a seeded particle order.  Pins oracle_np.band_contraction and tsp_upload_band_magnitudes."""
    _import_reference()
    from topsy.loader import PynbodyDataInMemory

    class _Snapshot(dict):
        def __len__(self):
            return len(self["I_mag"])

    rs = np.random.RandomState(20260)
    n = 4096
    snap = _Snapshot()
    for band, centre in (("I", 4.0), ("V", 5.0), ("U", 6.5)):
        mag = centre + 3.0 * rs.normal(size=n)
        mag[rs.choice(n, 40, replace=False)] = np.nan                 # stars without a magnitude in this band
        mag[rs.choice(n, 8, replace=False)] = rs.choice([-60.0, 90.0, 110.0, 130.0], size=8)   # float32 overflow / underflow
        mag[rs.choice(n, 4, replace=False)] = [np.inf, -np.inf, 0.0, -0.0]
        snap[band + "_mag"] = mag
    loader = PynbodyDataInMemory.__new__(PynbodyDataInMemory)
    loader.snapshot = snap
    loader._particle_order = rs.permutation(n)
    with np.errstate(all="ignore"):
        rgb = loader.get_rgb_masses()
    assert rgb.dtype == np.float32 and rgb.shape == (n, 3)
    np.savez_compressed(os.path.join(OUT, "band_magnitudes.npz"), I_mag=snap["I_mag"], V_mag=snap["V_mag"], U_mag=snap["U_mag"],
                        particle_order=loader._particle_order.astype(np.int64), rgb=rgb)
    print("wrote band_magnitudes.npz", rgb.shape, "non-finite:", int((~np.isfinite(rgb)).sum()), "zeros:", int((rgb == 0).sum()))


def main():
    if "--bands-only" in sys.argv:
        make_band_fixture()
        return
    make_band_fixture()
    _import_reference()
    from topsy.loader import TestDataLoader
    from topsy import sph, progressive_render, cell_layout, config
    from topsy.drawreason import DrawReason
    from topsy.colormap.implementation import Colormap, RGBColormap

    # ---------------------------------------------------------------- inputs
    for n in (1, 100, 1000):
        dl = TestDataLoader(None, n)
        ps = dl.get_pos_smooth()
        np.savez_compressed(
            os.path.join(OUT, f"testdata_n{n}.npz"),
            pos_smooth=ps.astype(np.float32),
            mass=dl.get_mass().astype(np.float32),
            qty=dl.get_named_quantity("test-quantity").astype(np.float32),
            rgb=dl.get_rgb_masses().astype(np.float32),
            smooth_f64=np.asarray(dl.get_smooth(), dtype=np.float64),
        )
    dlc = TestDataLoader(None, 1000, with_cells=True)
    np.savez_compressed(
        os.path.join(OUT, "testdata_n1000_cells.npz"),
        pos_smooth=dlc.get_pos_smooth().astype(np.float32),
        mass=dlc.get_mass().astype(np.float32),
        qty=dlc.get_named_quantity("test-quantity").astype(np.float32),
        cell_offsets=dlc._cell_layout._offsets, cell_lengths=dlc._cell_layout._lengths,
        cell_centres=dlc._cell_layout._centres,
    )

    # ---------------------------------------------------------------- cameras
    class _Vis:
        periodicity_scale = None

    class _Tex:
        width = height = 200

    def rot_x(a):  # reference visualizer.py:347-357 (named "_x_rotation_matrix": rotates about y)
        return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])

    def rot_y(a):
        return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])

    cams = {
        "identity_200": (np.eye(3), np.zeros(3), 200.0),
        "rot0_0p4_20": (rot_x(0.0) @ rot_y(0.4) @ np.eye(3), np.zeros(3), 20.0),
        "rot0_0p5_20": (rot_x(0.0) @ rot_y(0.5) @ np.eye(3), np.zeros(3), 20.0),
        "rot90_200": (np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]]), np.zeros(3), 200.0),
        "depthcam_20": (np.array([[1.0, 0, 0], [0, 0, 1], [0, -1, 0]]), np.zeros(3), 20.0),
        "offset_50": (rot_x(0.3) @ rot_y(-0.2), np.array([1.5, -2.0, 0.25]), 50.0),
    }
    cam_out = {}
    for name, (rot, off, scale) in cams.items():
        s = sph.SPH.__new__(sph.SPH)
        s._visualizer = _Vis()
        s._render_texture = _Tex()
        s.rotation_matrix, s.position_offset, s.scale = rot, off, scale
        s.min_pixels, s.max_pixels = 0.0, np.inf
        tp = s._get_transform_params()
        cam_out[name + ".rotation"] = rot
        cam_out[name + ".offset"] = off
        cam_out[name + ".scale"] = np.float64(scale)
        cam_out[name + ".transform"] = np.array(tp["transform"], dtype=np.float32)
        cam_out[name + ".scale_factor"] = np.array(tp["scale_factor"], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **cam_out)

    # ---------------------------------------------------------------- colormap LUTs + parameters
    luts = {}
    for cname in ("twilight_shifted", "viridis"):
        c = Colormap.__new__(Colormap)
        c._params = {"colormap_name": cname}
        luts[cname] = c._generate_mapping_rgba_f32(config.COLORMAP_NUM_SAMPLES)
    np.savez_compressed(os.path.join(OUT, "colormap_luts.npz"), **luts)

    params_out = []
    for log in (True, False):
        for weighted in (False, True):
            for S in (1.0, 6.67, 100.0):
                c = Colormap.__new__(Colormap)
                c._device = _RecordingDevice()
                c._parameter_buffer = None
                c._params = dict(Colormap._default_params, vmin=-3.25, vmax=1.5, log=log,
                                 weighted_average=weighted)
                c._update_parameter_buffer(640, 480, S)
                rec = c._device.queue.last
                params_out.append({
                    "log": log, "weighted": weighted, "S": S, "vmin_in": -3.25, "vmax_in": 1.5,
                    "vmin": float(rec["vmin"][0]), "vmax": float(rec["vmax"][0]),
                    "density_vmin": float(rec["density_vmin"][0]), "density_vmax": float(rec["density_vmax"][0]),
                    "window_aspect_ratio": float(rec["window_aspect_ratio"][0]), "gamma": float(rec["gamma"][0]),
                })

    # autorange on seeded value sets
    rng = np.random.RandomState(42)
    auto_cases = {
        "lognormal_4000": np.exp(rng.normal(size=4000)).astype(np.float32),
        "with_zeros_1000": np.concatenate([np.zeros(100), np.exp(rng.normal(size=900))]).astype(np.float32),
        "signed_5000": rng.normal(size=5000).astype(np.float32),
        "few_50": np.exp(rng.normal(size=50)).astype(np.float32),
        "const_300": np.full(300, 2.5, dtype=np.float32),
    }
    auto_out = {}
    for k, vals in auto_cases.items():
        c = Colormap.__new__(Colormap)
        c._params = dict(Colormap._default_params)
        c.update_parameters = lambda p, c=c: c._params.update(p)
        with np.errstate(all="ignore"):
            c._autorange_using_values(vals.copy())
        auto_out[k] = {"vmin": float(c._params["vmin"]), "vmax": float(c._params["vmax"]),
                       "log": bool(c._params["log"]),
                       "ui_range_linear": [float(x) for x in c._params["ui_range_linear"]],
                       "ui_range_log": [float(x) for x in c._params["ui_range_log"]]}
        r = RGBColormap.__new__(RGBColormap)
        r._params = dict(RGBColormap._default_params)
        with np.errstate(all="ignore"):
            r.autorange_vmin_vmax(np.abs(vals).reshape(-1, 1).repeat(3, axis=1))
        auto_out[k]["rgb_vmin"] = float(r._params["vmin"])
        auto_out[k]["rgb_vmax"] = float(r._params["vmax"])
    np.savez_compressed(os.path.join(OUT, "autorange_inputs.npz"), **auto_cases)

    mags = {
        "log_to_mag(1.0)": float(RGBColormap._log_output_to_mag_per_arcsec2(1.0)),
        "log_to_mag(2.0)": float(RGBColormap._log_output_to_mag_per_arcsec2(2.0)),
        "mag_to_log(1.0)": float(RGBColormap._mag_per_arcsec2_to_log_output(1.0)),
        "mag_to_log(2.0)": float(RGBColormap._mag_per_arcsec2_to_log_output(2.0)),
        "mag_to_log(38.0)": float(RGBColormap._mag_per_arcsec2_to_log_output(38.0)),
        "mag_to_log(40.0)": float(RGBColormap._mag_per_arcsec2_to_log_output(40.0)),
    }

    # ---------------------------------------------------------------- render progression sequences
    def run_script(rp, script):
        """script: list of (draw_reason_name, [elapsed after each block ...]) frames."""
        frames = []
        for reason, times in script:
            clear = rp.start_frame(DrawReason[reason])
            blocks = []
            t = 0.0
            it = iter(times)
            while True:
                b = rp.get_block(t)
                if not b:
                    break
                blocks.append([[int(x) for x in b[0]], [int(x) for x in b[1]]])
                try:
                    t = next(it)
                except StopIteration:
                    t = t + 1.0
                rp.end_block(t)
            if blocks:
                sf = rp.end_frame_get_scalefactor()
            else:
                sf = None
                rp._current_draw_reason = None
            frames.append({"reason": reason, "clear": bool(clear), "blocks": blocks, "scalefactor": sf,
                           "needs_refine": bool(rp.needs_refine())})
        return frames

    script = [("INITIAL_UPDATE", [0.02]), ("REFINE", [0.01]), ("REFINE", [0.05]), ("CHANGE", [0.2]),
              ("REFINE", [0.03]), ("EXPORT", [0.5, 1.0, 1.5, 2.0]), ("PRESENTATION_CHANGE", []),
              ("CHANGE", [0.004]), ("REFINE", [0.004]), ("REFINE", [0.004])]
    prog = {
        "script": script,
        "plain_1e6": run_script(progressive_render.RenderProgression(1000000), script),
        "plain_1e8": run_script(progressive_render.RenderProgression(100000000), script),
        "plain_777": run_script(progressive_render.RenderProgression(777, 100), script),
    }
    np.random.seed(1337)
    pos = np.random.uniform(0.0, 1.0, (20000, 3))
    cl, order = cell_layout.CellLayout.from_positions(pos, 0.0, 1.0, 6)
    rpc = progressive_render.RenderProgressionWithCells(cl, len(pos), 500)
    prog["cells_20000"] = run_script(rpc, script[:5])
    rpc.select_sphere((0.5, 0.5, 0.5), 0.2)
    prog["cells_20000_sphere"] = run_script(rpc, [("CHANGE", [0.01]), ("REFINE", [0.01])])
    prog["cells_20000_sphere_selected"] = [int(x) for x in rpc._selected_cells]
    prog["cells_20000_fraction"] = float(rpc.get_fraction_volume_selected())
    np.savez_compressed(os.path.join(OUT, "cells_20000.npz"), pos=pos, order=order, offsets=cl._offsets,
                        lengths=cl._lengths, centres=cl._centres,
                        in_sphere=cl.cells_in_sphere((0.5, 0.5, 0.5), 0.2))

    # ---------------------------------------------------------------- the reference tests' own KATs
    kats = _literals_from_test(os.path.join(REF, "tests", "test_render_output.py"),
                               {"reference_result", "result_ref", "expect", "expect_den", "expect_qty",
                                "expect_rgba"})
    np.savez_compressed(os.path.join(OUT, "reference_kats.npz"), **kats)

    with open(os.path.join(OUT, "host_params.json"), "w") as f:
        json.dump({"colormap_params": params_out, "autorange": auto_out, "mags": mags,
                   "progression": prog,
                   "config": {k: getattr(config, k) for k in dir(config) if k.isupper()}}, f, indent=1)
    print("wrote fixtures to", OUT)
    for k, v in kats.items():
        print("  KAT", k, v.shape)


if __name__ == "__main__":
    main()

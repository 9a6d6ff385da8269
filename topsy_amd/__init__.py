"""topsy_amd -- MI355X-native SPH particle-splatting backend for topsy's render path.

    import topsy_amd
    vis = topsy_amd.test(1000, render_resolution=200)      # as reference topsy.test(...)
    vis.scale = 200.0
    img = vis.get_sph_presentation_image()                 # (200, 200, 4) uint8

The GPU work goes through libtopsy_splat.so (include/topsy_splat.h).  There is no CPU fallback.
"""
from . import config
from .drawreason import DrawReason

__version__ = "0.1.0"


def test(nparticle=config.TEST_DATA_NUM_PARTICLES_DEFAULT, **kwargs):
    """Visualizer over the seeded synthetic snapshot (mirror of reference topsy.test, __init__.py:180-187)."""
    from . import visualizer, loader
    kwargs.pop("canvas_class", None)
    return visualizer.Visualizer(data_loader_class=loader.TestDataLoader, data_loader_args=(nparticle,),
                                 data_loader_kwargs={"with_cells": kwargs.pop("with_cells", False),
                                                     "periodic": kwargs.get("periodic_tiling", False)},
                                 **kwargs)


def from_arrays(pos, smooth, mass, quantities=None, rgb=None, with_cells=False, **kwargs):
    """Visualizer over caller-supplied numpy arrays (e.g. taken from a pynbody snapshot)."""
    from . import visualizer, loader
    return visualizer.Visualizer(data_loader_class=loader.ArrayDataLoader,
                                 data_loader_kwargs={"pos": pos, "smooth": smooth, "mass": mass,
                                                     "quantities": quantities, "rgb": rgb, "with_cells": with_cells},
                                 **kwargs)


def synthetic_on_device(n_total, first=0, count=None, h_cap=0.0, **kwargs):
    """Visualizer over a device-generated shard of the synthetic snapshot (1e8-1e9 particles)."""
    from . import visualizer, loader
    return visualizer.Visualizer(data_loader_class=loader.DeviceSyntheticLoader,
                                 data_loader_kwargs={"n_total": n_total, "first": first, "count": count,
                                                     "h_cap": h_cap},
                                 **kwargs)

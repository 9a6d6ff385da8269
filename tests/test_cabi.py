"""The C-ABI library loads here (no GPU) and exports every symbol include/topsy_splat.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "topsy_splat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from topsy_amd import _native
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert sorted(_native.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    from topsy_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/topsy_splat.h but not exported"
    assert _native.load_library().tsp_version() >= 100


def test_no_gpu_fails_loudly():
    """Without a GPU the product must raise -- never fall back to a CPU path."""
    from topsy_amd import _native
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.BackendUnavailable):
        _native.Context(64, 2)
    import topsy_amd
    with pytest.raises(_native.BackendUnavailable):
        topsy_amd.test(100, render_resolution=32)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "topsy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src and "oracle/" not in src.replace("the CPU oracle under oracle/", ""), f


def test_missing_library_fails_loudly(tmp_path):
    """TOPSY_SPLAT_LIB points the binding at another build; a path with no library must raise, not fall back."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['TOPSY_SPLAT_LIB'] = %r\n"
            "from topsy_amd import _native\n"
            "try:\n    _native.load_library()\nexcept _native.BackendUnavailable as e:\n    print('UNAVAILABLE', e)\n"
            % (ROOT, str(tmp_path / "libmissing.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "UNAVAILABLE" in out.stdout and "libmissing.so" in out.stdout, out.stdout + out.stderr


def test_every_option_name_is_documented():
    """INTEGRATION.md section 6 lists every name tsp_set_option accepts (csrc/tsp_api.hip), and names no option that is gone."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "topsy_amd", "csrc", "tsp_api.hip")).read()
    body = api[api.index("int tsp_set_option("):]
    body = body[:body.index('set_error("unknown option')]
    accepted = set(re.findall(r'!strcmp\(name, "([a-z_0-9]+)"\)', body))
    assert {"count_fragments", "chunk_cull", "mid_item_records", "stream_batch_chunks", "huge_split"} <= accepted
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    table = doc[doc.index("## 6. Tuning knobs"):]
    documented = set(re.findall(r"`([a-z_0-9]+)`", "\n".join(l.split("|")[1] for l in table.splitlines() if l.startswith("| `"))))
    assert accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))

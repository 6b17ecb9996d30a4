"""CPU: the C-ABI library loads, exports every symbol include/pvface.h declares, its host entry points work, and the
product path fails loudly (no CPU fallback) when no GPU is present."""
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from pyannote_video_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pvface.h")).read()
    declared = sorted(set(re.findall(r"\b(pvf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name), name
    assert set(declared) == set(_lib.EXPORTS)
    assert l.pvf_version() >= 100


def test_host_entry_points_without_gpu():
    from pyannote_video_amd import _lib
    assert _lib.munkres(np.array([[4., 1.], [2., 3.]])) == [(0, 1), (1, 0)]
    ov = _lib.overlap_matrix([(0, 0, 10, 10)], [(5, 5, 15, 15), (20, 20, 30, 30)], 0.2)
    assert ov.tolist() == [[25.0, 0.0]]
    assert _lib.overlap_matrix([(0, 0, 10, 10)], [(5, 5, 15, 15)], 0.3)[0, 0] == 0.0   # gated: 25 < 0.3 * 100


def test_no_cpu_fallback():
    from pyannote_video_amd import _lib
    from pyannote_video_amd.runtime import Context
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.PvfError, match="no HIP device|no CPU fallback"):
        Context(device=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pyannote-video_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert not re.search(r"pvo_[a-z0-9_]+\s*\(", src) and "libpvo" not in src, f   # comments may cite oracle files

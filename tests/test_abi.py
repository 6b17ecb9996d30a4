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


def test_dist_library_exports_every_declared_symbol():
    """libpvface_dist.so (the RCCL all-gather behind a C ABI, include/pvface_dist.h) loads and exports what its header declares"""
    from pyannote_video_amd import dist
    hdr = open(os.path.join(ROOT, "include", "pvface_dist.h")).read()
    declared = sorted(set(re.findall(r"\b(pvfd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) == 7
    l = dist.dist_lib()
    for name in declared:
        assert hasattr(l, name), name
    assert set(declared) == set(dist.DIST_EXPORTS)


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


def test_associate_equals_overlap_plus_munkres_composition():
    """pvf_associate == the reference's _associate written out with the two host helpers (tracking.py:136-182)"""
    from pyannote_video_amd import _lib
    rng = np.random.default_rng(5)
    for trial in range(200):
        nt, nd = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        def boxes(n):
            c = rng.uniform(50, 400, (n, 2)); s = rng.uniform(20, 120, (n, 1))
            return [tuple(float(v) for v in (c[i, 0] - s[i, 0], c[i, 1] - s[i, 0], c[i, 0] + s[i, 0], c[i, 1] + s[i, 0])) for i in range(n)]
        T, D = boxes(nt), boxes(nd)
        ratio = float(rng.choice([0.2, 0.5]))
        n = max(nt, nd)
        area = np.zeros((n, n))
        area[:nt, :nd] = _lib.overlap_matrix(T, D, ratio)
        want = [(t, d) for t, d in _lib.munkres(np.max(area) - area) if t < nt and d < nd and area[t, d] > 0.]
        assert _lib.associate(T, D, ratio) == want

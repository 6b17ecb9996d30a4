"""CPU: the roofline objects bench.py prints (bench.detector_rooflines) and the counter summary it attaches (tools/pmc_kernels_json.py):
algorithmic work per frame from the level schedule, the family that took most of the run first, traffic attached only while the
detector's sources hash the same."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fam(**ms):
    fam = {name: {"ms": 0.0, "launches": 0} for name in bench.FAMILIES}
    for k, (m, n) in ms.items():
        fam[k] = {"ms": m, "launches": n}
    return fam


def test_algorithmic_work_of_the_detector_kernels_at_1080p():
    fam = _fam(pyramid=(400.0, 40), fhog=(450.0, 40), score_screened=(90.0, 40))
    rl = bench.detector_rooflines(fam, 1080, 1920, 5000, 128)
    assert [o["kernel"].split(" ")[0] for o in rl] == ["fhog_split_ml_k", "resize_rows_k", "score_screen_k"]      # most time first
    by = {o["kernel"].split(" ")[0]: o for o in rl}
    per_launch = 5000 / 40
    assert by["fhog_split_ml_k"]["algorithmic_bytes_per_launch"] / per_launch == pytest.approx(132.75e6, rel=2e-3)
    assert by["resize_rows_k"]["algorithmic_bytes_per_launch"] / per_launch == pytest.approx(168.9e6, rel=2e-3)      # K1
    assert by["score_screen_k"]["flop_per_launch"] / per_launch == pytest.approx(414725 * 3100 * 5 * 2.0, rel=1e-9)  # K3 / K3s
    o = by["fhog_split_ml_k"]
    assert o["bound"] == "hbm" and o["unit"] == "GB/s" and o["peak"] == 8000.0
    assert o["achieved"] == pytest.approx(o["algorithmic_bytes_per_launch"] / (o["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert o["frac"] == pytest.approx(o["achieved"] / 8000.0, abs=1e-4)
    s = by["score_screen_k"]
    assert s["bound"] == "mfma" and s["peak"] == bench.F16_PEAK_TFLOPS
    dense = bench.detector_rooflines(_fam(score=(680.0, 40)), 1080, 1920, 5000, 128)
    assert len(dense) == 1 and dense[0]["kernel"].startswith("score_roll_k") and dense[0]["peak"] == bench.FP32_PEAK_TFLOPS
    assert dense[0]["frac"] == pytest.approx(12.856e9 * 125 / 17e-3 / 1e12 / 157.3, rel=2e-3)


def test_traffic_is_attached_only_for_the_measured_sources(monkeypatch, tmp_path):
    fam = _fam(pyramid=(400.0, 40), fhog=(450.0, 40), score_screened=(90.0, 40))
    pm = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_KERNELS_FILE)))
    rl = bench.detector_rooflines(fam, 1080, 1920, 5000, 128)
    current = pm["detector_sha256_16"] == bench.detector_hash()
    for o in rl:
        key = {"fhog_split_ml_k": "fhog", "resize_rows_k": "pyramid", "score_screen_k": "score_screened"}[o["kernel"].split(" ")[0]]
        if current:
            assert o["traffic"] == pm["kernels"][key]["traffic_bytes_per_launch"]
            assert "traffic_attached_from_profiles_not_measured_in_this_run" in o
            # what the kernels fetch beyond their algorithmic bytes (the screening pass since round 6: its 16-byte pieces of the
            # plane-group layout sit 48 bytes apart, 1.34 x -- the price of the FHOG kernel's whole-line stores)
            assert 1.0 < o["traffic"] / (o["algorithmic_bytes_per_launch"]) < (1.45 if key == "score_screened" else 1.3)
        else:
            assert o["traffic"] is None
    # another configuration, or changed sources: the figure is dropped, not carried
    assert all(o["traffic"] is None for o in bench.detector_rooflines(fam, 720, 1280, 5000, 125))
    monkeypatch.setattr(bench, "detector_hash", lambda: "0" * 16)
    assert all(o["traffic"] is None for o in bench.detector_rooflines(fam, 1080, 1920, 5000, 128))


def test_detector_hash_ignores_comments():
    import hashlib
    import re
    h0 = bench.detector_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h0)
    src = "int a = 1; // note\n/* block\n comment */ int   b;\n"
    strip = lambda s: re.sub(r"\s+", " ", re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", s, flags=re.S)))
    assert strip(src) == strip("int a = 1;\nint b;\n")


def test_pmc_summary_to_json(tmp_path):
    fetch = tmp_path / "f.txt"; write = tmp_path / "w.txt"
    fetch.write_text("void resize_rows_k<16, 1>                 n=160   FETCH_SIZE=1000\n"
                     "fhog_split_ml_k                           n=8     FETCH_SIZE=5000\n"
                     "score_screen_k                            n=4     FETCH_SIZE=4000\n"
                     "score_list_k                              n=4     FETCH_SIZE=100\n"
                     "score_roll_k                              n=4     FETCH_SIZE=3900\n")
    write.write_text("void resize_rows_k<16, 1>                 n=160   WRITE_SIZE=500\n"
                     "fhog_split_ml_k                           n=8     WRITE_SIZE=6000\n"
                     "score_screen_k                            n=4     WRITE_SIZE=10\n")
    os.makedirs(os.path.join(ROOT, "gpurun_out", "zz_test"), exist_ok=True)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_kernels_json.py"), str(fetch), str(write), "zz_test"], cwd=ROOT,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        d = json.load(open(os.path.join(ROOT, "profiles", "zz_test_pmc_kernels.json")))
    finally:
        for p in (os.path.join(ROOT, "profiles", "zz_test_pmc_kernels.json"), os.path.join(ROOT, "gpurun_out", "zz_test", "pmc_kernels.json")):
            if os.path.exists(p):
                os.remove(p)
    k = d["kernels"]
    assert k["pyramid"]["batches_in_pass"] == 8 and k["pyramid"]["traffic_bytes_per_launch"] == (2 * 1000 + 500) * 160 / 8 * 1024
    assert k["fhog"]["traffic_bytes_per_launch"] == (2 * 5000 + 6000) * 1024
    assert k["score_screened"]["batches_in_pass"] == 4 and k["score_screened"]["traffic_bytes_per_launch"] == (2 * 4100 + 10) * 1024
    assert k["score"]["batches_in_pass"] == 4 and d["detector_sha256_16"] == bench.detector_hash()


def _avg_linkage_log(D, choose):
    """average-linkage agglomeration of a distance matrix; `choose(candidates)` picks among the pairs at the minimum distance"""
    import numpy as np
    D = np.array(D, np.float64)
    n = len(D)
    alive, size, log = list(range(n)), [1] * n, []
    while len(alive) > 1:
        best = min(D[a, b] for i, a in enumerate(alive) for b in alive[i + 1:])
        cand = [(a, b) for i, a in enumerate(alive) for b in alive[i + 1:] if abs(D[a, b] - best) <= 1e-12]
        a, b = choose(cand)
        for k in alive:
            if k not in (a, b):
                D[a, k] = D[k, a] = (size[a] * D[a, k] + size[b] * D[b, k]) / (size[a] + size[b])
        size[a] += size[b]
        alive.remove(b)
        log.append((a, b, best, size[a]))
    return np.array(log, np.float64)


def test_merge_order_verdict_up_to_ties():
    import numpy as np
    rng = np.random.default_rng(3)
    P = rng.normal(size=(6, 4))
    P = np.concatenate([P, P, P[:2]])                    # replayed points: exact ties, some three deep
    D = np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1))
    first = _avg_linkage_log(D, lambda c: c[0])
    last = _avg_linkage_log(D, lambda c: c[-1])
    assert bench.merge_order_verdict(first, first.copy()) == "exact"
    assert not np.array_equal(first[:, :2], last[:, :2])
    v = bench.merge_order_verdict(last, first)
    assert v.startswith("equal up to ties"), v
    wobble = last.copy(); wobble[:, 2] *= 1 + 3e-14     # the product's table agrees with the oracle's to a few 1e-14
    assert bench.merge_order_verdict(wobble, first).startswith("equal up to ties")
    # a different dendrogram is not excused: another distance, or another partition at an untied merge
    wrong = first.copy(); wrong[-1, 2] *= 1.001
    assert bench.merge_order_verdict(wrong, first).startswith("MISMATCH")
    k = int(np.nonzero(np.diff(first[:, 2]) > 1e-9)[0][0]) + 1             # the first untied merge: another cluster joins instead
    other = [c for c in first[k + 1:, 1].tolist() if c not in (first[k, 0], first[k, 1])][0]
    wrong = first.copy(); wrong[k, 1] = other
    assert bench.merge_order_verdict(wrong, first).startswith("MISMATCH")
    assert bench.merge_order_verdict(first[:-1], first).startswith("MISMATCH")

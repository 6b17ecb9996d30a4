"""GPU: the detector's screening pass (csrc/screen.hip) never changes a result.  With it the windows are first scored on the f16 matrix
cores and only those within a proven error bound of the threshold go through the exact fp32 chain; the candidates -- position, filter,
exact score -- must be the dense kernel's bit for bit: at the operating point, with the threshold swept through the bulk of the score
distribution (thousands of windows inside the bound), when the list of pairs overflows (the call is repeated on the dense kernel), and
against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _raw(ctx, frame, adj):
    return ctx.detect_raw(frame, 1, adj)


def test_screened_candidates_are_the_dense_kernels(ctx, small_video):
    frames = [ctx.upload(small_video.frame(i)) for i in (0, 7, 11)]
    adjs = (0.0, -0.05, -0.3, -0.80712890625)
    ctx.detector_screening(False)
    dense = [[_raw(ctx, f, a) for a in adjs] for f in frames]
    ctx.detector_screening(True)
    try:
        s0 = ctx.detector_screening_stats()
        got = [[_raw(ctx, f, a) for a in adjs] for f in frames]
        s1 = ctx.detector_screening_stats()
    finally:
        ctx.detector_screening(True)
    assert got == dense
    n_cands = sum(len(x) for per in dense for x in per)
    assert n_cands > 3000                                            # (the lowered thresholds reach into the bulk)
    assert s1["batches"] - s0["batches"] == len(frames) * len(adjs) and s1["retries"] == s0["retries"]
    assert s1["listed"] - s0["listed"] >= n_cands                    # every candidate went through the exact chain
    assert 0 <= s1["pipe_err"] <= 3200 * 2.0 ** -22                  # what the device's matrix pipe was measured to lose
    assert all(0.005 < b < 0.2 for b in s1["bounds"])
    for f in frames:
        f.release()


def test_screened_batches_equal_dense_and_oracle(ctx, oracle, small_video):
    from pyannote_video_amd import models
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    frames = [ctx.upload(small_video.frame(i % 12)) for i in range(24)]
    ctx.detector_screening(False)
    dense = ctx.detect_many(frames, 8, 1, arrays=True)
    ctx.detector_screening(True)
    try:
        got = ctx.detect_many(frames, 8, 1, arrays=True)
        raw5 = _raw(ctx, frames[5], 0.0)
    finally:
        ctx.detector_screening(True)
    assert all(np.array_equal(a, b) for a, b in zip(got, dense))
    assert raw5 == det.detect_raw(small_video.frame(5), 1, 0.0)          # position, filter and the exact chain's score, bit for bit
    for f in frames:
        f.release()


def test_list_overflow_repeats_the_call_on_the_dense_kernel(ctx, small_video):
    frame = ctx.upload(small_video.frame(3))
    adj = -0.80712890625
    ctx.detector_screening(False)
    dense = _raw(ctx, frame, adj)
    many = ctx.detect_many([frame] * 5, 2, 1, adj, arrays=True)
    ctx.detector_screening(True, 1000)                                  # far fewer pairs than this threshold lists
    try:
        s0 = ctx.detector_screening_stats()
        got = _raw(ctx, frame, adj)
        got_many = ctx.detect_many([frame] * 5, 2, 1, adj, arrays=True)
        s1 = ctx.detector_screening_stats()
        again = _raw(ctx, frame, 0.0)                                    # screening is back on for the next call
        s2 = ctx.detector_screening_stats()
    finally:
        ctx.detector_screening(True, 1 << 20)
    assert len(dense) > 1000 and got == dense
    assert all(np.array_equal(a, b) for a, b in zip(got_many, many))
    assert s1["retries"] - s0["retries"] >= 2                        # (both calls; a call that also outgrew the candidate slots counts twice)
    assert s2["batches"] > s1["batches"] and s2["retries"] == s1["retries"]
    ctx.detector_screening(False)
    try:
        assert again == _raw(ctx, frame, 0.0)
    finally:
        ctx.detector_screening(True)
    frame.release()


def test_feature_above_the_assumed_bound_repeats_the_call_on_the_dense_kernel(ctx, small_video, monkeypatch):
    """the error bound assumes FHOG features <= 0.4 / 0.849; the kernel checks every feature it reads and gives the call up otherwise.
    Real features never get there, so the limits are scaled down (PVF_SCREEN_LIMIT_SCALE, read at every launch) until they do."""
    frame = ctx.upload(small_video.frame(2))
    ctx.detector_screening(False)
    dense = _raw(ctx, frame, 0.0)
    ctx.detector_screening(True)
    s0 = ctx.detector_screening_stats()
    monkeypatch.setenv("PVF_SCREEN_LIMIT_SCALE", "0.25")
    got = _raw(ctx, frame, 0.0)
    s1 = ctx.detector_screening_stats()
    monkeypatch.delenv("PVF_SCREEN_LIMIT_SCALE")
    again = _raw(ctx, frame, 0.0)
    s2 = ctx.detector_screening_stats()
    assert got == dense and again == dense
    assert s1["retries"] == s0["retries"] + 1 and s2["retries"] == s1["retries"] and s2["batches"] == s1["batches"] + 1
    frame.release()


def test_library_bounds_equal_the_restated_derivation(ctx):
    """the per-filter bounds the library computes at model load == tests/screen_bound.py (whose analytic part test_screen_bound.py holds
    against the oracle on every window of a frame, on the CPU)"""
    import screen_bound as sb
    from pyannote_video_amd import models
    W = np.ascontiguousarray(models.load_container(models.DEFAULT_DETECTOR)["det.w"], np.float32).reshape(5, 10, 10, 32)
    got = np.array(ctx.detector_screening_stats()["bounds"])
    assert np.allclose(got, sb.bounds(W), rtol=1e-9, atol=0)

"""CPU: the analytic part of the screening pass's error bound (tests/screen_bound.py = csrc/screen.hip restated) against the oracle on real
feature maps -- every window of every level of a frame: the f16 products' real sum stays within e_w + e_f + e_chain of the exact chain
(the matrix pipe's own accumulation error, e_pipe, is what the device probe measures), no feature exceeds the maxima the bound assumes,
and a window the bound does NOT list has an exact score below the threshold."""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

import screen_bound as sb


def _detector(oracle):
    from pyannote_video_amd import models
    return oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))


def test_weight_scale_and_quantisation():
    rng = np.random.default_rng(5)
    W = (rng.normal(size=(5, 10, 10, 32)) * 0.03).astype(np.float32)
    q, s = sb.quantised(W)
    assert s == 2.0 ** round(np.log2(s)) and 64 <= np.abs(W).max() * s < 128           # a power of two, the largest weight below 2^7
    assert np.abs(q - W).max() <= np.abs(W).max() * 2.0 ** -11 + 2.0 ** -25 / s * 2        # round to nearest: half an f16 step
    big = (W * 1e4).astype(np.float32)
    assert np.isfinite(sb.quantised(big)[0]).all() and sb.scale_of(big) < 1.0            # no weight overflows f16
    assert sb.scale_of(np.zeros_like(W)) == 1.0
    b = sb.bounds(W)
    assert b.shape == (5,) and (b > 0).all()
    assert np.allclose(sb.bounds(big), b * 1e4, rtol=2e-3)                                # the bound scales with the weights


def test_f16_towards_zero():
    x = np.array([0.0, 0.1, 0.4, 0.40001, 0.8485, 1e-5, 6.1e-5, 0.3999], np.float32)
    t = sb.f16_towards_zero(x)
    assert (t <= x).all() and (x - t <= np.maximum(x * 2.0 ** -10, 2.0 ** -24)).all()
    assert sb.f16_towards_zero(np.array([sb.LIM_LO, sb.LIM_HI], np.float32)).tolist() == [sb.LIM_LO, sb.LIM_HI]      # the limits ARE f16 values
    # a feature above the assumed maximum converts to something above the limit the kernel compares with
    assert sb.f16_towards_zero(np.array([np.nextafter(np.float32(sb.FM_LO), np.float32(1)), np.nextafter(np.float32(sb.FM_HI), np.float32(1))]))[0] > sb.LIM_LO
    assert sb.f16_towards_zero(np.array([np.nextafter(np.float32(sb.FM_HI), np.float32(1))], np.float32))[0] > sb.LIM_HI


def test_bound_holds_on_every_window_of_a_frame(oracle, small_video):
    det = _detector(oracle)
    W = det.w
    F = W.shape[0]
    Wq, _ = sb.quantised(W)
    t = sb.terms(W)
    analytic = t["e_w"] + t["e_f"] + t["e_chain"]
    full = sb.bounds(W)
    assert (full > analytic).all() and (full < 0.2).all()
    rgb = small_video.frame(5)
    up = det.pyramid_level(rgb, 1, 0)
    L = det.levels(up.shape[0], up.shape[1])
    fr, fc = det.s.frows, det.s.fcols
    worst = np.zeros(F)
    n_windows = listed = cands = 0
    for l in range(L):
        img = det.pyramid_level(rgb, 1, l)
        feat = oracle.fhog(img, det.s.cell, fr, fc)
        if feat.shape[0] < fr or feat.shape[1] < fc:
            continue
        assert feat.min() >= 0 and feat[..., :27].max() <= 0.4 * (1 + 1e-6) and feat[..., 27:31].max() <= 0.84853 and np.all(feat[..., 31] == 0)
        assert sb.f16_towards_zero(feat[..., :27]).max() <= sb.LIM_LO and sb.f16_towards_zero(feat[..., 27:31]).max() <= sb.LIM_HI
        f16 = sb.f16_towards_zero(feat)
        win = np.moveaxis(sliding_window_view(f16, (fr, fc), axis=(0, 1)), 2, -1)      # [oh, ow, fr, fc, 32]
        win = win.reshape(win.shape[0], win.shape[1], -1)
        for f in range(F):
            S = det.score_level(feat, f)[fr // 2: fr // 2 + win.shape[0], fc // 2: fc // 2 + win.shape[1]].astype(np.float64)
            Sq = win @ Wq[f].reshape(-1)                                                     # the f16 products' sum, in double
            worst[f] = max(worst[f], np.abs(Sq - S).max())
            th = float(det.thresh[f])
            not_listed = Sq < th - full[f]
            assert (S[not_listed] < th).all()
            n_windows += S.size; listed += int((~not_listed).sum()); cands += int((S >= th).sum())
    assert (worst <= analytic).all(), (worst, analytic)
    assert n_windows > 50000 and cands >= 3 and cands <= listed <= cands + 200

"""The error bound of the detector's screening pass (csrc/screen.hip: screen_prepare_model), restated in numpy -- test infrastructure.
S = the exact chain (3100 fmaf), S' = what the f16 matrix cores return for the same window; the library lists a window for the exact chain
when S' >= threshold - bound[filter].  The terms (see the header of screen.hip):
  e_w     sum FM |w' - w|                       w' = f16(scale * w) / scale, round to nearest (numpy's float16 conversion), scale = 2^k
  e_f     sum |w'| max(2^-10 FM, 2^-14)         features converted with round-towards-zero; a subnormal may be flushed
  e_sub   sum over f16-SUBNORMAL w' of FM |w'|  a pipe that flushes subnormal inputs loses the whole product (at most 1.6e-4 for any model:
          only weights 2^21 below the largest are subnormal after the power-of-two scaling); carried whether or not the device flushes
  e_pipe  3200 * 2^-22 * sum FM |w'| (1 + 2^-10)  the <= 3220 rounding additions of a sum inside the matrix pipe (3100 non-zero products, 120
          hand-overs between MFMAs; adding a zero entry is exact) allowed just under 4 x an IEEE fp32 rounding error each
  e_chain g(3100) sum FM |w|                    the exact chain's own distance from the real sum, g(n) = n u / (1 - n u), u = 2^-24
FM = 0.4004 for the 27 orientation planes, 0.8492 for the 4 texture planes (the FHOG normalisation's maxima + the f16 step the kernel's
check of the data loses)."""
import math

import numpy as np

FM_LO, FM_HI = 0.4004, 0.8492
LIM_LO, LIM_HI = 0.400146484375, 0.8486328125


def scale_of(W):
    wmax = float(np.abs(W).max())
    if wmax == 0:
        return 1.0
    _, ex = math.frexp(wmax)
    return math.ldexp(1.0, max(-100, min(100, 7 - ex)))


def quantised(W):
    """(w' as float64, scale): what the B fragments of score_screen_k hold, divided by the scale"""
    s = scale_of(W)
    q = (W.astype(np.float64) * s).astype(np.float32).astype(np.float16).astype(np.float64) / s
    return q, s


def plane_max():
    fm = np.full(32, FM_LO)
    fm[27:31] = FM_HI
    fm[31] = 0.0
    return fm


def terms(W):
    """W: [filters, 10, 10, 32] float32 -> dict of the five terms per filter (float64 arrays)"""
    Wq, scale = quantised(W)
    W64 = W.astype(np.float64)
    fm = plane_max()
    keep = np.arange(32) < 31
    u = 2.0 ** -24
    e_w = (fm * np.abs(Wq - W64))[..., keep].sum(axis=(1, 2, 3))
    e_f = (np.abs(Wq) * np.maximum(fm * 2.0 ** -10, 2.0 ** -14))[..., keep].sum(axis=(1, 2, 3))
    a_h = (fm * np.abs(Wq))[..., keep].sum(axis=(1, 2, 3))
    a_w = (fm * np.abs(W64))[..., keep].sum(axis=(1, 2, 3))
    sub = (Wq != 0) & (np.abs(Wq) * scale < 2.0 ** -14)
    e_sub = (fm * np.abs(Wq) * sub)[..., keep].sum(axis=(1, 2, 3))
    return {"e_w": e_w, "e_f": e_f, "e_sub": e_sub, "e_pipe": 3200.0 * 2.0 ** -22 * a_h * (1.0 + 2.0 ** -10), "e_chain": 3100.0 * u / (1.0 - 3100.0 * u) * a_w}


def bounds(W):
    t = terms(W)
    return 1.02 * (t["e_w"] + t["e_f"] + t["e_sub"] + t["e_pipe"] + t["e_chain"]) + 1e-6


def f16_towards_zero(x):
    """float32 array -> float64 array of the f16 values v_cvt_pkrtz_f16_f32 produces (x >= 0)"""
    h = x.astype(np.float16)
    up = h.astype(np.float32) > x                    # rounded up: step one f16 down
    bits = h.view(np.uint16).copy()
    bits[up] -= 1
    return bits.view(np.float16).astype(np.float64)

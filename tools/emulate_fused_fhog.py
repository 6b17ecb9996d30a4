#!/usr/bin/env python
"""CPU emulation (numpy, float32, lane by lane) of the index logic of csrc/detect.hip:fhog_split_ml_k (the strip / band geometry the gradient and the vote waves share) -- strips of 64 lanes,
chunks of feature rows, bands, the right-neighbour hand-over, the two alternating accumulator sets, the three-row energy window
-- checked against the oracle's FHOG on random images.  Development aid: it validates everything about the kernel except the
hardware-specific pieces (DPP lane shifts, ds_add_f32), which the GPU parity tests cover.
    python tools/emulate_fused_fhog.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
from oracle import oracle as O  # noqa: E402

F = np.float32
DIRX = np.array([1.0000, 0.9397, 0.7660, 0.500, 0.1736, -0.1736, -0.5000, -0.7660, -0.9397], F)
DIRY = np.array([0.0000, 0.3420, 0.6428, 0.8660, 0.9848, 0.9848, 0.8660, 0.6428, 0.3420], F)
FUSED_OUT = 61


def grad_planes(img):
    """(mag, bin) of every pixel where a gradient exists at all (1 <= y < h-1, 1 <= x < w-1); validity is applied by the caller"""
    h, w, _ = img.shape
    I = img.astype(np.int32)
    gx = np.zeros((h, w, 3), np.int32); gy = np.zeros((h, w, 3), np.int32)
    gx[:, 1:-1] = I[:, 2:] - I[:, :-2]
    gy[1:-1] = I[2:] - I[:-2]
    v = gx * gx + gy * gy
    bx, by, bv = gx[..., 0].copy(), gy[..., 0].copy(), v[..., 0].copy()
    for k in (1, 2):
        m = v[..., k] > bv
        bx[m], by[m], bv[m] = gx[..., k][m], gy[..., k][m], v[..., k][m]
    mag = np.sqrt(bv.astype(F))
    fx, fy = bx.astype(F), by.astype(F)
    best = np.zeros((h, w), F); bo = np.zeros((h, w), np.int32)
    for o in range(9):
        dot = fx * DIRX[o] + fy * DIRY[o]
        m1 = dot > best
        best[m1] = dot[m1]; bo[m1] = o
        m2 = (~m1) & (-dot > best)
        best[m2] = -dot[m2]; bo[m2] = o + 9
    return mag, bo


def cell_features(h, n):
    eps = F(0.0001)
    z1 = [n[4], n[1], n[3], n[0]]; z2 = [n[5], n[2], n[4], n[1]]; z3 = [n[7], n[4], n[6], n[3]]; z4 = [n[8], n[5], n[7], n[4]]
    nn = [F(0.2) * np.sqrt(F(F(F(z1[k] + z2[k]) + z3[k]) + z4[k]) + eps) for k in range(4)]
    nv = [F(0.1) / nn[k] for k in range(4)]
    o = np.zeros(32, F); t = [F(0)] * 4
    for g in range(0, 18, 3):
        hh = [[F(min(h[g + j], nn[k]) * nv[k]) for k in range(4)] for j in range(3)]
        for j in range(3):
            o[g + j] = F(hh[j][0] + hh[j][1]) + F(hh[j][2] + hh[j][3])
        for k in range(4):
            t[k] = F(t[k] + F(F(hh[0][k] + hh[1][k]) + hh[2][k]))
    ts = F(2 * 0.2357)
    t = [F(t[k] * ts) for k in range(4)]
    for g in range(9):
        s = F(h[g] + h[g + 9])
        hh = [F(min(s, nn[k]) * nv[k]) for k in range(4)]
        o[18 + g] = F(hh[0] + hh[1]) + F(hh[2] + hh[3])
    o[27:31] = t
    return o


def fused(img, chunk_rows, pad=10):
    h, w, _ = img.shape
    cells_nr, cells_nc = int(h / 8.0 + 0.5), int(w / 8.0 + 0.5)
    visible_nr, visible_nc = min(cells_nr * 8, h) - 1, min(cells_nc * 8, w) - 1
    hog_nr, hog_nc = cells_nr - 2, cells_nc - 2
    fh, fw = hog_nr + pad - 1, hog_nc + pad - 1
    oy = ox = (pad - 1) // 2
    out = np.zeros((fh, fw, 32), F)
    mag, bo = grad_planes(img)
    strips = (hog_nc + FUSED_OUT - 1) // FUSED_OUT
    chunks = (hog_nr + chunk_rows - 1) // chunk_rows
    lanes = np.arange(64)
    for sx in range(strips):
        for cy in range(chunks):
            y0 = cy * chunk_rows
            R = min(chunk_rows, hog_nr - y0)
            hx = FUSED_OUT * sx + 1 + lanes
            x_first = 8 * hx - 12
            acc = np.zeros((2, 18, 64), F)
            hprev = np.zeros((18, 64), F)
            e0 = np.zeros(64, F); e1 = np.zeros(64, F); e2 = np.zeros(64, F)
            for gb in range(y0 + 1, y0 + R + 4):
                setU, setL = gb & 1, (gb - 1) & 1
                for i in range(8):
                    y = 8 * gb + i - 12
                    if not (1 <= y < visible_nr):
                        continue
                    m = np.zeros((8, 64), F); b = np.zeros((8, 64), np.int64)
                    for p in range(8):
                        x = x_first + p
                        ok = (x >= 1) & (x < visible_nc)
                        xs = np.clip(x, 0, w - 1)
                        m[p] = np.where(ok, mag[y, xs], F(0))
                        b[p] = np.where((x >= 0) & (x < w), bo[y, xs], 0)
                    fy = F((i + 0.5) / 8.0)
                    for j in range(16):
                        p = j & 7
                        if j < 8:
                            mv, bv = m[p], b[p]
                        else:                                          # from lane + 1 (lane 63: nothing -> 0)
                            mv = np.concatenate([m[p][1:], [F(0)]]); bv = np.concatenate([b[p][1:], [0]])
                        fx = F((p + 0.5) / 8.0)
                        wx = fx if j < 8 else F(1.0) - fx
                        acc[setL, bv, lanes] = acc[setL, bv, lanes] + F(F(F(1.0) - fy) * wx) * mv
                        acc[setU, bv, lanes] = acc[setU, bv, lanes] + F(fy * wx) * mv
                c = gb - 1
                hn = acc[setL].copy()
                acc[setL] = 0
                e = np.zeros(64, F)
                for o in range(9):
                    s2 = hn[o] + hn[o + 9]
                    e = e + s2 * s2
                e2, e1, e0 = e1, e0, e
                if c >= y0 + 3:
                    yh = c - 3
                    for L in range(1, FUSED_OUT + 1):
                        x = FUSED_OUT * sx + L - 1
                        if x >= hog_nc:
                            continue
                        n = [e2[L - 1], e2[L], e2[L + 1], e1[L - 1], e1[L], e1[L + 1], e0[L - 1], e0[L], e0[L + 1]]
                        out[yh + oy, x + ox] = cell_features(hprev[:, L], n)
                hprev = hn
    return out


def main():
    rng = np.random.default_rng(0)
    for (h, w, chunk) in [(96, 150, 32), (203, 131, 7), (64, 640, 3), (133, 517, 16)]:
        yy, xx = np.mgrid[0:h, 0:w]
        img = (rng.integers(0, 50, (h, w, 3)) + 100 + 80 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 7.0)[..., None]).clip(0, 255).astype(np.uint8)
        ref = O.fhog(img, 8, 10, 10)
        got = fused(img, chunk)
        same = np.array_equal(ref, got)
        print(h, w, chunk, "bit-exact" if same else "MISMATCH max |d| = %g at %s" % (np.abs(ref - got).max(), np.argwhere(ref != got)[:3].tolist()))
        assert same


if __name__ == "__main__":
    main()

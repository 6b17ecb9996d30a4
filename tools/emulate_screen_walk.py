"""Index algebra of score_screen_k (csrc/screen.hip) emulated in numpy, lane by lane: which cell and planes a lane loads per phase, what the
two DPP row moves of shift_bases hand on, how B fragments are packed (screen_prepare_model), how the ten accumulator slots rotate, and
which (row, column, filter) an accumulator element is at emission.  With float64 "MFMAs" the emitted sums must equal the direct window sums.
    python tools/emulate_screen_walk.py [NG=1..4]
Used by tests/test_host_logic.py-style CPU tests (tests/test_screen_walk_emulation.py): an edit of the kernel's index logic is first made here."""
import sys

import numpy as np


def pack_b(W):
    """[10 m][12 n'][64 lanes][8]: lane (column jc = 5 * shift + filter, plane octet kq) holds w[filter][m][n' - shift][8 kq .. 8 kq + 7]"""
    Bh = np.zeros((10, 12, 64, 8))
    for l in range(64):
        jc, kq = l & 15, l >> 4
        if jc >= 15:
            continue
        sft, f = jc // 5, jc % 5
        for j in range(12):
            n = j - sft
            if 0 <= n < 10:
                Bh[:, j, l, :] = W[f, :, n, 8 * kq: 8 * kq + 8]
    Bh[:, :, 48:, 7] = 0.0                      # plane 31 is padding
    return Bh


def mfma(Af, Bf):
    """16x16x32: A lane (row i = l & 15, octet l >> 4), B lane (column l & 15, octet l >> 4) -> D[row][col]"""
    Am = Af.reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32)
    Bm = Bf.reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32)
    return Am @ Bm.T


def walk(feat, W, NG, c_base=0):
    """one strip of NG groups over the whole height of `feat` [fh][fw][32]: {(r, c, f): sum} for the windows the strip owns"""
    fh, fw = feat.shape[:2]
    out_rows = fh - 9
    Bh = pack_b(W)
    lanes = np.arange(64)
    I, KQ = lanes & 15, lanes >> 4
    padded = np.zeros((fh + 1, fw + 48 * (NG + 1) + 16, 32))
    padded[:fh, :fw] = feat                                   # beyond the row / the map: zeros (the buffer range check)

    def load_cls(row, cls):
        out = np.zeros((NG + 1, 64, 8))
        for g in range(NG + 1):
            cell = c_base + 48 * g + 3 * I + cls
            for l in range(64):
                out[g, l] = padded[row, cell[l], 8 * KQ[l]: 8 * KQ[l] + 8]
        return out

    def shift_bases(a):
        for g in range(NG + 1):
            new = np.zeros_like(a[g])
            lane15 = (lanes & 15) == 15
            new[~lane15] = a[g][lanes[~lane15] + 1]           # row_shl:1
            if g < NG:
                new[lane15] = a[g + 1][lanes[lane15] - 15]    # row_shr:15 of the next group
            a[g] = new

    acc = np.zeros((10, NG, 16, 16))
    got = {}
    A = [load_cls(0, c) for c in range(3)]
    for s in range(fh):
        done = np.zeros((NG, 16, 16))
        for j in range(12):
            cls = j % 3
            if j > 0 and cls == 0:
                for k in range(3):
                    shift_bases(A[k])
            for q in range(9, -1, -1):
                for g in range(NG):
                    v = acc[q, g] + mfma(A[cls][g], Bh[q, j])
                    if j < 11:
                        acc[q, g] = v
                    elif q == 9:
                        done[g] = v
                    else:
                        acc[q + 1, g] = v
        acc[0] = 0
        r_out = s - 9
        if r_out >= 0:
            for g in range(NG):
                for pos in range(16):
                    for jc in range(15):
                        cc = c_base + 48 * g + 3 * pos + jc // 5
                        if cc < fw - 9:
                            got[(r_out, cc, jc % 5)] = done[g, pos, jc]
        A = [load_cls(s + 1, c) for c in range(3)]
    return got


def check(NG, fh=13, extra=7, seed=0):
    rng = np.random.default_rng(seed)
    fw = 48 * NG + 9 + extra
    feat = rng.random((fh, fw, 32)); feat[..., 31] = 0
    W = rng.normal(size=(5, 10, 10, 32)) * 0.05; W[..., 31] = 0
    got = walk(feat, W, NG)
    bad = 0
    for (r, c, f), v in got.items():
        ref = float((feat[r:r + 10, c:c + 10, :] * W[f]).sum())
        bad += abs(ref - v) > 1e-9
    return len(got), (fh - 9) * min(48 * NG, fw - 9) * 5, bad


if __name__ == "__main__":
    for ng in ([int(sys.argv[1])] if len(sys.argv) > 1 else [1, 2, 3, 4]):
        print("NG=%d: %d sums emitted, %d expected, %d wrong" % ((ng,) + check(ng)))

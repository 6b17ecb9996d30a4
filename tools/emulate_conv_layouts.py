#!/usr/bin/env python
"""CPU emulation (numpy, lane by lane) of the index algebra of the embedder's two dedicated convolution kernels (csrc/resnet.hip):
  stem_conv_k    the first layer from the uint8 chip: 13 chip rows of a block as bytes, 77 k-pairs along the 21 consecutive bytes under a tap
                 row (stem_frag_k's fragment order, the pad's zero weight, the per-lane byte base + the pair's immediate offset, the mean
                 of the colour a lane's k has), nine 32-row tiles of four output rows;
  conv3x3_c32_k  a band of seven output rows: the [9][37][33] tile with its zero border (which slots a thread parks, which of them are
                 border columns / rows outside the image: the three bit masks), conv_frag_k's fragment order, the 144 pairs' offsets.
Each is checked against a direct convolution of the same data (float64: what is validated is WHICH products are summed, not their
rounding -- the GPU tests hold the descriptors against the oracle).  Development aid and the body of tests/test_conv_layouts.py.
    python tools/emulate_conv_layouts.py
"""
import numpy as np

MEAN = (122.782, 117.001, 104.298)


# ---------------------------------------------------------------------------------------------------------------------------------
def generic_weights_rgb(w):
    """[cout][3][7][7] -> the generic layout [cout][224]: k = (r * 7 + q) * 4 + c, fourth channel and the tail zero (ctx.hip: add_conv)"""
    cout = w.shape[0]
    wt = np.zeros((cout, 224))
    for c in range(3):
        for r in range(7):
            for q in range(7):
                wt[:, (r * 7 + q) * 4 + c] = w[:, c, r, q]
    return wt


def stem_frag(wt):
    """stem_frag_k: frag[pair][lane]"""
    frag = np.zeros((77, 64))
    for p in range(77):
        for l in range(64):
            r, t = p // 11, 2 * (p - (p // 11) * 11) + (l >> 5)
            frag[p, l] = wt[l & 31, (r * 7 + t // 3) * 4 + t % 3] if t < 21 else 0.0
    return frag


def stem_block(chip, frag, oy0):
    """one block of stem_conv_k: output rows oy0 .. oy0 + 3 of one 150 x 150 x 3 chip -> [4][72][32] (before bias / affine / ReLU)"""
    S, OW = 150, 72
    flat = chip.reshape(-1).astype(np.float64)
    first = 2 * oy0 * S * 3
    s_in = np.zeros(1464 * 4)                                   # the block's dwords; past the chip's end: zero (buffer descriptor)
    n = min(len(s_in), len(flat) - first)
    s_in[:n] = flat[first:first + n]
    out = np.zeros((4 * OW, 32))
    for wave in range(3):
        for t in range(3):
            tile = wave * 3 + t
            acc = np.zeros((32, 32))                            # [pixel row of the tile][channel]
            for s in range(77):
                r, pp = s // 11, s - (s // 11) * 11
                off = r * S * 3 + 2 * pp
                for kh in range(2):
                    mean3 = (MEAN[1] if kh else MEAN[0], MEAN[0] if kh else MEAN[2], MEAN[2] if kh else MEAN[1])
                    for li in range(32):
                        m = tile * 32 + li
                        oy, ox = m // OW, m % OW
                        base = ((2 * oy) * S + 2 * ox) * 3 + kh
                        a = (s_in[base + off] - mean3[pp % 3]) / 256.0
                        acc[li] += a * frag[s, kh * 32:kh * 32 + 32]          # lane (n, kh) supplies b[n][k = 2 s + kh]
            out[tile * 32:tile * 32 + 32] = acc
    return out.reshape(4, OW, 32)


def stem_direct(chip, w, oy0):
    x = (chip.astype(np.float64) - np.array(MEAN)) / 256.0      # [150][150][3]
    out = np.zeros((4, 72, 32))
    for oy in range(4):
        for ox in range(72):
            patch = x[2 * (oy0 + oy):2 * (oy0 + oy) + 7, 2 * ox:2 * ox + 7, :]          # [r][q][c]
            out[oy, ox] = np.einsum("rqc,ncrq->n", patch, w)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
def generic_weights_c32(w):
    """[32][32][3][3] -> [cout][288]: k = (r * 3 + q) * 32 + c"""
    wt = np.zeros((32, 288))
    for c in range(32):
        for r in range(3):
            for q in range(3):
                wt[:, (r * 3 + q) * 32 + c] = w[:, c, r, q]
    return wt


def conv_frag(wt, kpad):
    frag = np.zeros((kpad // 2, 64))
    for s in range(kpad // 2):
        for l in range(64):
            frag[s, l] = wt[l & 31, 2 * s + (l >> 5)]
    return frag


def c32_band(x, frag, band):
    """one work item of conv3x3_c32_k: output rows 7 band .. 7 band + 6 of one face, x [35][35][32] -> [7][35][32] (before the epilogue)"""
    HW, TW, PITCH, ROWS = 35, 37, 33, 7
    NSLOT, PER = 9 * TW * 8, (9 * TW * 8 + 255) // 256
    oy0 = band * ROWS
    src = x.reshape(-1)
    T = np.full(9 * TW * PITCH, np.nan)                         # (NaN: a slot nobody parks would show)
    for tid in range(256):
        s_off, m_col, m_row0, m_row8 = [], 0, 0, 0
        for j in range(PER):
            i = tid + j * 256
            c4, px = i & 7, i >> 3
            ry, cx = px // TW, px % TW
            if i < NSLOT and 0 <= cx - 1 < HW:
                m_col |= 1 << j
            if ry == 0:
                m_row0 |= 1 << j
            if ry == 8:
                m_row8 |= 1 << j
            s_off.append((ry * HW + (cx - 1)) * 32 + 4 * c4)
        ok = m_col & ~(m_row0 if band == 0 else 0) & ~(m_row8 if band == 4 else 0)
        base = (oy0 - 1) * HW * 32                              # (relative to the face; may be negative for band 0: never dereferenced)
        for j in range(PER):
            i = tid + j * 256
            v = np.zeros(4)
            if (ok >> j) & 1:
                a = base + s_off[j]
                assert 0 <= a and a + 4 <= len(src), "a load outside the face"
                v = src[a:a + 4]
            if i < NSLOT:
                d = (i >> 3) * PITCH + 4 * (i & 7)
                T[d:d + 4] = v
    out = np.zeros((ROWS * HW, 32))
    for wave in range(4):
        for t in range(2):
            tile = wave * 2 + t
            acc = np.zeros((32, 32))
            for s in range(144):
                tap = s >> 4
                r, q = tap // 3, tap % 3
                off = (r * TW + q) * PITCH + 2 * (s & 15)
                for kh in range(2):
                    for li in range(32):
                        m = tile * 32 + li
                        if m >= ROWS * HW:
                            m = 0
                        oy, ox = m // HW, m % HW
                        a = T[(oy * TW + ox) * PITCH + kh + off]
                        acc[li] += a * frag[s, kh * 32:kh * 32 + 32]
            for row in range(32):
                m = tile * 32 + row
                if m < ROWS * HW:
                    out[m] = acc[row]
    assert not np.isnan(out).any(), "an MFMA read a slot that was never parked"
    return out.reshape(ROWS, HW, 32)


def c32_direct(x, w, band):
    xp = np.zeros((37, 37, 32))
    xp[1:36, 1:36] = x
    out = np.zeros((7, 35, 32))
    for oy in range(7):
        for ox in range(35):
            patch = xp[band * 7 + oy:band * 7 + oy + 3, ox:ox + 3, :]                    # [r][q][c]
            out[oy, ox] = np.einsum("rqc,ncrq->n", patch, w)
    return out


def main():
    rng = np.random.default_rng(1)
    chip = rng.integers(0, 256, (150, 150, 3), dtype=np.uint8)
    w = rng.normal(size=(32, 3, 7, 7))
    frag = stem_frag(generic_weights_rgb(w))
    for oy0 in (0, 36, 68):
        err = np.abs(stem_block(chip, frag, oy0) - stem_direct(chip, w, oy0)).max()
        print("stem rows %2d..%2d: max |emulated - direct| = %.2e" % (oy0, oy0 + 3, err))
    x = rng.normal(size=(35, 35, 32))
    w3 = rng.normal(size=(32, 32, 3, 3))
    f3 = conv_frag(generic_weights_c32(w3), 288)
    for band in range(5):
        err = np.abs(c32_band(x, f3, band) - c32_direct(x, w3, band)).max()
        print("c32 band %d: max |emulated - direct| = %.2e" % (band, err))


if __name__ == "__main__":
    main()

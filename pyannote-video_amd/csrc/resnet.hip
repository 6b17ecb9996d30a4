// resnet.hip -- K7: the 29-conv ResNet of dlib's face_recognition_model_v1 (reference pyannote/video/face/face.py:62,74-76)
// as NHWC implicit-GEMM convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak),
// with bias + affine + residual add (incl. dlib's zero-extending add_prev and the 2x2 avg-pool skip) + ReLU fused
// into the epilogue.  Batch = all faces of many frames.
#include "pvf_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// chip alignment geometry (host; a few doubles per face).  [EXT get_face_chip_details + chip_details]
void face_chip_details(const EmbedModel& m, const int32_t* pts, ChipDetails* out)
{
    const double size = m.chip_size, padding = m.chip_padding;
    double fx[51], fy[51], tx[51], ty[51];
    int n = 0;
    for (int i = 17; i < 68; ++i) {
        if ((55 <= i && i <= 59) || (65 <= i && i <= 67)) continue;
        if (17 <= i && i <= 26) continue;
        fx[n] = ((padding + (double)m.mean_shape[2 * (i - 17)]) / (2 * padding + 1)) * size;
        fy[n] = ((padding + (double)m.mean_shape[2 * (i - 17) + 1]) / (2 * padding + 1)) * size;
        tx[n] = pts[2 * i]; ty[n] = pts[2 * i + 1];
        ++n;
    }
    double mfx = 0, mfy = 0, mtx = 0, mty = 0;
    for (int i = 0; i < n; ++i) { mfx += fx[i]; mfy += fy[i]; mtx += tx[i]; mty += ty[i]; }
    mfx /= n; mfy /= n; mtx /= n; mty /= n;
    double a = 0, b = 0, s = 0;
    for (int i = 0; i < n; ++i) {
        const double ax = fx[i] - mfx, ay = fy[i] - mfy, bx = tx[i] - mtx, by = ty[i] - mty;
        a += ax * bx + ay * by;
        b += ax * by - ay * bx;
        s += ax * ax + ay * ay;
    }
    const double ca = a / s, cb = b / s;
    const double scale = std::sqrt(ca * ca + cb * cb);
    const double hx = size / 2.0, hy = size / 2.0;
    const double cx = (ca * (hx - mfx) - cb * (hy - mfy)) + mtx;
    const double cy = (cb * (hx - mfx) + ca * (hy - mfy)) + mty;
    const double wv = size * scale;
    out->l = cx - wv / 2; out->t = cy - wv / 2; out->r = cx + wv / 2; out->b = cy + wv / 2;
    out->cs = ca / scale; out->sn = cb / scale;
    out->rows = m.chip_size; out->cols = m.chip_size;
}

struct ConvArgs {
    const float* in; int B, H, W, Cin;
    const float* w; int K;             // K = ksz*ksz*Cin (k = (r*ksz + s)*Cin + c), weights transposed: [Cout][K padded to 32]
    const float* frag;                 // the same weights in MFMA fragment order (conv_frag_k), for the layers that have a kernel of their own; or null
    const float* bias; const float* gamma; const float* beta;
    float* out; int OH, OW, Cout;      // output tensor dims
    int AH, AW;                        // conv-valid dims (<= OH, OW)
    int ksz, stride, pad;
    int relu;
    int skip_mode;                     // 0 none, 1 identity [B][OH][OW][Cout], 2 avg-pool 2x2 s2 of x [B][XH][XW][XC]
    const float* skip; int XH, XW, XC, SH, SW;
};

// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32.  Block tile BM x BN = (64 WM) x (32 WN) output pixels x channels, 4 waves; a wave owns
// a 64 x 32 tile = two 32 x 32 accumulators that share every B fragment.  K (= taps x input channels, the order of the oracle's chain)
// advances in chunks of 32 through LDS; the next chunk's global loads are in flight during the MFMAs of the current one.
// LDS layout: one row of 32 k-values per output pixel (A) / output channel (B), padded to 36 floats, with the even k first and the odd k
// behind them (position (k >> 1) + 16 (k & 1)): lane (i, h) of a 32 x 32 x 2 MFMA needs k = 2 s + h for s = 0..15, i.e. 16 CONTIGUOUS floats
// = 4 ds_read_b128 per 16 MFMAs (the 36-float pitch makes them conflict-free), instead of one ds_read_b32 per operand and MFMA.
// Weights are stored transposed ([cout][K padded to 32], ctx.hip) so that both tiles are staged with the same 16-byte loads along k.
// (Keeping the activations in that even / odd order in HBM as well -- so that a chunk could be staged with plain 16-byte copies -- was
// measured 10 % SLOWER: the epilogue's stores and skip loads then scatter within each 128-byte line, and four scattered dwords per lane
// quad cost the memory pipeline more than the 36 register moves per chunk that the split on the way into LDS costs.)

// exact x / d for 0 <= x < 2^22, 1 <= d < 2^22 (rd = 1.0f / d): float estimate, one correction step.  Output pixel -> (face, row, column)
// needs two divisions per tile row; 64-bit integer division is emulated with ~100 instructions, and 16 of them per thread took about as
// long as the K loop of a 3 x 3 x 32 layer.
__device__ __forceinline__ int small_div(int x, int d, float rd)
{
    int q = (int)((float)x * rd);
    const int r = x - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// bias + affine + residual add + ReLU and the stores of one block's results (shared by the two K loops above / below); `smem` is free
// after the K loop's last barrier and takes each wave's 64 x 32 tile for the transposing store path
template <int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, float* smem, const f32x16& acc0, const f32x16& acc1, long M, long m0, int n0,
                                              int rel0, int b_first, int hw, float r_hw, float r_ow, int wave, int lane)
{
    constexpr int PITCH = 36;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, kh = lane >> 5;
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int col = n0 + wn * 32 + li;
    const float bias = a.bias[col], g = a.gamma[col], bt = a.beta[col];
    if (a.AH == a.OH && a.AW == a.OW && a.skip_mode != 2) {
        // every output pixel is a convolution result and the skip tensor (if any) has the output's shape: 24 of the 29 layers.
        // The accumulators hold one channel per lane (32 four-byte stores and skip loads per lane, two 128-byte rows per instruction);
        // the wave's tile is turned through LDS so that a lane owns four neighbouring channels of a pixel: 8 sixteen-byte stores (and
        // skip loads), each instruction covering eight full 128-byte rows.  Per-element arithmetic and its order are unchanged.
        float* T = smem + wave * 64 * PITCH;            // (all waves are past the K loop's last barrier)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                T[(t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh) * PITCH + li] = (t == 0 ? acc0[reg] : acc1[reg]);
        __builtin_amdgcn_wave_barrier();                // LDS serves a wave's requests in order; this only pins the compiler's order
        const int c4 = lane & 7, cb = n0 + wn * 32 + 4 * c4;
        const float4 bias4 = *reinterpret_cast<const float4*>(a.bias + cb), g4 = *reinterpret_cast<const float4*>(a.gamma + cb),
                     bt4 = *reinterpret_cast<const float4*>(a.beta + cb);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const long m = m0 + wm * 64 + row;
            if (m >= M) continue;
            const float4 x = *reinterpret_cast<const float4*>(&T[row * PITCH + 4 * c4]);
            float4 v = make_float4((x.x + bias4.x) * g4.x + bt4.x, (x.y + bias4.y) * g4.y + bt4.y, (x.z + bias4.z) * g4.z + bt4.z,
                                   (x.w + bias4.w) * g4.w + bt4.w);
            if (a.skip_mode == 1) {
                const float4 sk = *reinterpret_cast<const float4*>(a.skip + (size_t)m * a.Cout + cb);
                v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
            }
            if (a.relu) { v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y; v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w; }
            *reinterpret_cast<float4*>(a.out + (size_t)m * a.Cout + cb) = v;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            const long m = m0 + wm * 64 + t * 32 + row;
            if (m >= M) continue;
            const int rel = rel0 + wm * 64 + t * 32 + row;
            const int bb = small_div(rel, hw, r_hw);
            const int rem = rel - bb * hw;
            const int oy = small_div(rem, a.OW, r_ow);
            const int ox = rem - oy * a.OW;
            const int b = b_first + bb;
            float v = 0.0f;
            if (oy < a.AH && ox < a.AW) v = ((t == 0 ? acc0[reg] : acc1[reg]) + bias) * g + bt;
            if (a.skip_mode == 1) v += a.skip[(size_t)m * a.Cout + col];
            else if (a.skip_mode == 2) {
                if (col < a.XC && oy < a.SH && ox < a.SW) {
                    const float* q = a.skip + (((size_t)b * a.XH + 2 * oy) * a.XW + 2 * ox) * a.XC + col;
                    v += (((q[0] + q[a.XC]) + q[(size_t)a.XW * a.XC]) + q[(size_t)a.XW * a.XC + a.XC]) * 0.25f;
                }
            }
            if (a.relu && v < 0.0f) v = 0.0f;
            a.out[(size_t)m * a.Cout + col] = v;
        }
}

template <int WM, int WN>
__global__ void __launch_bounds__(256) conv_mfma_k(ConvArgs a)
{
    constexpr int BM = 64 * WM, BN = 32 * WN, KC = 32, PITCH = 36;
    static_assert(WM * WN == 4, "four waves per block");
    // A tile, B tile; after the K loop the same space takes each wave's 64 x 32 results for the transposing epilogue
    constexpr int SM_ROWS = (BM + BN > 256) ? BM + BN : 256;
    __shared__ __attribute__((aligned(16))) float smem[SM_ROWS * PITCH];
    float* As = smem;
    float* Bs = smem + BM * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long M = (long)a.B * a.OH * a.OW;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }

    // staging roles: A tile = BM rows x 8 float4 (4 consecutive k = 4 input channels of one tap), B tile = BN rows x 8 float4.
    // Input pixels are fetched with raw buffer loads: a tap that falls outside the image (or a row past M, or k past K) gets an offset
    // beyond the buffer and comes back as zeros -- no branch around any load, so the address arithmetic of the next chunk is one
    // straight-line block the scheduler spreads between the MFMAs of the current one.
    constexpr int A_F4 = BM * 8 / 256, B_F4 = BN * 8 / 256;
    constexpr int RSRC_FLAGS = 0x00020000;
    constexpr int OOB = (int)0x80000000;
    const int hw = a.OH * a.OW;
    const int b_first = (int)(m0 / hw);                                    // the tile's first face: offsets below stay small (scalar)
    const int rel0 = (int)(m0 - (long)b_first * hw);                       // tile rows are rel0 + i within [first face ...): < hw + BM
    const float r_hw = 1.0f / (float)hw, r_ow = 1.0f / (float)a.OW;
    const size_t face = (size_t)a.H * a.W * a.Cin;
    const size_t in_bytes = (size_t)(a.B - b_first) * face * sizeof(float);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b_first * face), 0,
                                                                           in_bytes > 0x7ffffff0u ? 0x7ffffff0 : (int)in_bytes, RSRC_FLAGS);
    int pa_off[A_F4], pa_y[A_F4], pa_x[A_F4];       // float index of the input pixel under tap (0, 0); its coordinates (y = -2^20: no pixel)
#pragma unroll
    for (int q = 0; q < A_F4; ++q) {
        const int i = (tid + q * 256) >> 3;
        const long m = m0 + i;
        pa_off[q] = 0; pa_y[q] = -(1 << 20); pa_x[q] = 0;
        if (m < M) {
            const int rel = rel0 + i;
            const int bb = small_div(rel, hw, r_hw);
            const int rem = rel - bb * hw;
            const int oy = small_div(rem, a.OW, r_ow);
            const int ox = rem - oy * a.OW;
            if (oy < a.AH && ox < a.AW) {
                pa_y[q] = oy * a.stride - a.pad; pa_x[q] = ox * a.stride - a.pad;
                pa_off[q] = ((bb * a.H + pa_y[q]) * a.W + pa_x[q]) * a.Cin;
            }
        }
    }
    const int Kpad = (a.K + KC - 1) / KC * KC;
    const int j4 = tid & 7;                             // which float4 of a row this thread moves
    const float* wrow[B_F4];
#pragma unroll
    for (int q = 0; q < B_F4; ++q) wrow[q] = a.w + (size_t)(n0 + ((tid + q * 256) >> 3)) * Kpad + 4 * j4;
    u32x4 va[A_F4];
    float4 vb[B_F4];
    // Three stages per chunk, one chunk apart: its byte offsets are worked out (plain VALU work, spread between the MFMAs of an earlier
    // chunk), its loads are issued right after the barrier of the chunk before it, and it is parked in LDS one barrier later.
    // Chunk walk: (tap row, tap column, first channel) advance with scalar adds, no division in the loop (input channels: a multiple of 32).
    int f_r = 0, f_s = 0, f_c = 0, f_k = 0;             // the chunk whose offsets are computed next
    int voff[A_F4], woff = 0;
    auto offsets = [&]() {
        const int kq = f_k + 4 * j4;
        const int r = f_r, sft = f_s;                   // wave-uniform
        const int delta = (r * a.W + sft) * a.Cin + f_c + 4 * j4;
        const int k_ok = (kq < a.K) ? -1 : 0;
#pragma unroll
        for (int q = 0; q < A_F4; ++q) {
            const int iy = pa_y[q] + r, ix = pa_x[q] + sft;
            // all-ones when the tap lies inside the image: plain integer logic (no short-circuit branches around the loads)
            const int ok = k_ok & -(int)((unsigned)iy < (unsigned)a.H) & -(int)((unsigned)ix < (unsigned)a.W);
            voff[q] = (((pa_off[q] + delta) * 4) & ok) | (OOB & ~ok);
        }
        woff = f_k;
        if (f_k + KC < Kpad) {                          // (past the last chunk: stay on it; those loads are issued but never parked)
            f_k += KC;
            f_c += KC;
            if (f_c >= a.Cin) { f_c = 0; if (++f_s == a.ksz) { f_s = 0; ++f_r; } }
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int q = 0; q < A_F4; ++q) va[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[q], 0, 0);
#pragma unroll
        for (int q = 0; q < B_F4; ++q) vb[q] = *reinterpret_cast<const float4*>(wrow[q] + woff);     // zero-padded beyond K
    };
    auto park = [&]() {
#pragma unroll
        for (int q = 0; q < A_F4; ++q) {
            uint32_t* row = reinterpret_cast<uint32_t*>(&As[((tid + q * 256) >> 3) * PITCH + 2 * j4]);
            *reinterpret_cast<uint2*>(row) = make_uint2(va[q].x, va[q].z);          // k = 4 j4, 4 j4 + 2   (even half)
            *reinterpret_cast<uint2*>(row + 16) = make_uint2(va[q].y, va[q].w);     // k = 4 j4 + 1, + 3    (odd half)
        }
#pragma unroll
        for (int q = 0; q < B_F4; ++q) {
            float* row = &Bs[((tid + q * 256) >> 3) * PITCH + 2 * j4];
            *reinterpret_cast<float2*>(row) = make_float2(vb[q].x, vb[q].z);
            *reinterpret_cast<float2*>(row + 16) = make_float2(vb[q].y, vb[q].w);
        }
    };
    const int li = lane & 31, kh = lane >> 5;
    const float* pa0 = &As[(wm * 64 + li) * PITCH + 16 * kh];
    const float* pa1 = pa0 + 32 * PITCH;
    const float* pb = &Bs[(wn * 32 + li) * PITCH + 16 * kh];
    offsets();
    issue();
    offsets();
    for (int k0 = 0; k0 < Kpad; k0 += KC) {
        park();
        __syncthreads();
        issue();                                      // chunk k0 + KC: in flight during all of this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        offsets();                                    // chunk k0 + 2 KC
        f32x4 fa0[4], fa1[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa0[q] = *reinterpret_cast<const f32x4*>(pa0 + 4 * q);
            fa1[q] = *reinterpret_cast<const f32x4*>(pa1 + 4 * q);
            fb[q] = *reinterpret_cast<const f32x4*>(pb + 4 * q);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s >> 2][s & 3], fb[s >> 2][s & 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s >> 2][s & 3], fb[s >> 2][s & 3], acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    conv_epilogue<WM, WN>(a, smem, acc0, acc1, M, m0, n0, rel0, b_first, hw, r_hw, r_ow, wave, lane);
}

// (Two other forms of the K loop were built and measured no faster, and are gone: tiles straight into LDS by LDS-DMA, round 4 --
// profiles/r04_conv_lds_vs_regs.txt -- and a spatial input tile reused across the 9 taps, one fetch per pixel and one barrier per block,
// round 5 -- profiles/r05_conv_band_experiment.txt: a layer's MFMAs and its unavoidable HBM traffic are of the same size and run one
// after the other inside a block; the staging of the A operand is not what the matrix pipe waits for.)

// ---------------------------------------------------------------------------------------------------
// The first layer (7 x 7, stride 2, 3 -> 32 channels on the 150 x 150 chip) straight from the uint8 chips.  Through the generic kernel it
// cost three passes over HBM (an input pass: bytes -> 4-channel floats, 270 KB per face written and read back; the result, 663 KB per
// face) and a K of 7 * 7 * 4 = 196 padded to 224 for 147 real products; its A tile was gathered tap by tap from global memory although
// a block's whole input is 6 KB.  Here a block owns FOUR output rows of one face (288 pixels = nine 32-row MFMA tiles, three per wave,
// three waves): the 13 chip rows under them are parked in LDS as bytes (one contiguous, 4-byte aligned range of the chip), the layer's
// weights next to them in fragment order (stem_frag_k, once per model), and the K loop is unrolled over 77 k-pairs -- the 21 values
// under a tap ROW (7 taps x 3 colours) are 21 consecutive bytes of a chip row, so k runs along them in pairs, eleven per tap row, the
// last pair padded with a zero weight.  Per MFMA a lane reads ONE byte at a constant offset from its pixel's base (ds_read_u8 with an
// immediate), converts it ((q - mean) / 256, as the input pass did) and feeds it with the pair's weight fragment to its three tiles.
// The chain runs over the same products in the same order as the generic kernel's did (taps row-major, colour fastest; that one's
// fourth channel and its k = 196 .. 223 added exact zeros); the pairing inside an MFMA differs ((c0, c1), (c2, next tap's c0) ...
// instead of (c0, c1), (c2, 0)) -- and the descriptors are the same bit for bit (tools/bench_embed.py prints their checksum): the
// 32 x 32 x 2 MFMA adds its two products to the accumulator one after the other.  Measured on 4096 faces: 24.1 -> 22.9 ms for the
// whole network with 98 pairs (the generic pairing), 22.4 -> 21.8 with 77.
#define STEM_ROWS 4
#define STEM_IN_DW 1464                                   // 13 rows x 450 bytes = 5850 bytes
#define STEM_NP 77                                        // k-pairs: 7 tap rows x 11 pairs (21 values of a row + one zero-weight pad)
__global__ void __launch_bounds__(64) stem_frag_k(const float* __restrict__ w, float* __restrict__ frag)
{
    // frag[pair][lane] = weight of channel lane & 31 for the pair's k of this half-wave: tap row r, position t = 2 pp + (lane >> 5) in the
    // row's 21 values (t = 3 q + c: tap column q, colour c; t = 21: the pad); w is the generic layout [cout][224], k = (r * 7 + q) * 4 + c
    const int p = blockIdx.x, l = threadIdx.x;
    const int r = p / 11, t = 2 * (p - r * 11) + (l >> 5);
    frag[p * 64 + l] = (t < 21) ? w[(size_t)(l & 31) * 224 + (r * 7 + t / 3) * 4 + t % 3] : 0.0f;
}

__global__ void __launch_bounds__(192) stem_conv_k(const uint8_t* __restrict__ chips, int B, const float* __restrict__ frag, const float* __restrict__ bias,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out)
{
    constexpr int S = 150, OW = 72;
    __shared__ uint32_t s_in[STEM_IN_DW];
    __shared__ __attribute__((aligned(16))) float s_b[STEM_NP * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int face = blockIdx.y, oy0 = blockIdx.x * STEM_ROWS;
    {
        // the chips behind a descriptor (a dword past the last chip's end reads as zero); a block's range starts on a multiple of 4:
        // a chip is 67500 bytes, two chip rows 900
        const size_t total = (size_t)B * S * S * 3;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(chips + (size_t)face * S * S * 3), 0,
                                                                             (int)std::min<size_t>(total - (size_t)face * S * S * 3, 0x7ffffff0u), 0x00020000);
        const int first = 2 * oy0 * S * 3;
        for (int i = tid; i < STEM_IN_DW; i += 192) s_in[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * i, first, 0);
        const float4* f4 = reinterpret_cast<const float4*>(frag);
        float4* b4 = reinterpret_cast<float4*>(s_b);
        for (int i = tid; i < STEM_NP * 16; i += 192) b4[i] = f4[i];
    }
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    const uint8_t* px = reinterpret_cast<const uint8_t*>(s_in);
    const uint8_t* base[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int m = (wave * 3 + t) * 32 + li;           // pixel of the block: row m / 72, column m % 72
        const int oy = m / OW, ox = m - oy * OW;
        base[t] = px + ((2 * oy) * S + 2 * ox) * 3 + kh;
    }
    // colour of this lane's k in pair pp of a row: (2 pp + kh) % 3 -- three cases by pp % 3
    const float M0 = 122.782f, M1 = 117.001f, M2 = 104.298f;
    const float mean3[3] = {kh ? M1 : M0, kh ? M0 : M2, kh ? M2 : M1};
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
    for (int s = 0; s < STEM_NP; ++s) {
        const int r = s / 11, pp = s - r * 11;
        const int off = r * S * 3 + 2 * pp;               // + kh (in the base): the row's 21 bytes under the tap row are consecutive
        const float b = s_b[s * 64 + lane];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float a = ((float)base[t][off] - mean3[pp % 3]) / 256.0f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
    }
    // C layout of the 32 x 32 MFMA: column (channel) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const float bs = bias[li], g = gamma[li], bt = beta[li];
    float* o = out + ((size_t)face * OW + oy0) * OW * 32 + li;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (wave * 3 + t) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            float v = (acc[t][reg] + bs) * g + bt;
            v = v < 0.0f ? 0.0f : v;
            o[(size_t)m * 32] = v;
        }
}

// ---------------------------------------------------------------------------------------------------
// The six 3 x 3 layers of the 35 x 35 x 32 stage (a quarter of the network's arithmetic, and through the generic kernel its least efficient
// part: 32 output channels are ONE MFMA column tile, so every A element staged through LDS feeds a single MFMA) in the first layer's
// form.  A work item is a band of SEVEN output rows of one face (245 pixels: eight 32-row tiles, two per wave); the nine input rows
// under it sit in LDS with their zero border, [row][column][33 floats] (the odd pitch spreads a tile's pixels over the banks), the
// layer's weights next to them in fragment order (81 KB together: two blocks per CU); the K loop is unrolled over the 144 k-pairs, and
// per MFMA a lane reads ONE float at a constant offset from its pixel's base.  Products, pairing and chain order are the generic
// kernel's (k = (tap, channel), channel pairs (2 j, 2 j + 1)), so is the epilogue's arithmetic: bit-identical results.
// How it got here (4096 faces, per layer; the generic kernel: 1.03-1.17 ms; the layer's MFMAs alone at the clock they run at: 0.8 ms --
// measured with the loads and the stores compiled out): one block per band with the tile loaded slot by slot 1.58 ms (eleven round
// trips to HBM in a row), with all of a thread's slots requested before the first is parked 1.1-1.3 ms, resident blocks that request the
// NEXT band's rows before the current band's MFMAs 0.95-1.12 ms.  What is left beside the MFMAs is parking the rows (44 LDS writes per
// thread), two barriers and the stores, at two waves per SIMD.
#define C32_ROWS 7
#define C32_TILE_FLOATS (9 * 37 * 33)
__global__ void __launch_bounds__(64) conv_frag_k(const float* __restrict__ w, int Kpad, float* __restrict__ frag)
{
    const int s = blockIdx.x, l = threadIdx.x;            // frag[s][lane] = w[channel lane & 31][k = 2 s + (lane >> 5)]
    frag[s * 64 + l] = w[(size_t)(l & 31) * Kpad + 2 * s + (l >> 5)];
}

__global__ void __launch_bounds__(256) conv3x3_c32_k(const float* __restrict__ in, const float* __restrict__ frag, const float* __restrict__ bias,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ skip,
                                                     float* __restrict__ out, int n_items)
{
    constexpr int HW = 35, TW = 37, PITCH = 33, NP = 144;
    constexpr int NSLOT = 9 * TW * 8, PER = (NSLOT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem_c32[];
    float* T = smem_c32;                                  // [9][37][33]
    float* Bf = smem_c32 + C32_TILE_FLOATS;               // [144][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    // A block is resident for the whole layer and takes every gridDim.x-th band (item = face * 5 + band): the weight fragments are parked
    // once, and the NEXT band's input rows (and this band's skip values) are requested before the current band's MFMAs start, so that
    // what a band costs beside its MFMAs is parking 11 float4 per thread and issuing its stores.
    {
        const float4* f4 = reinterpret_cast<const float4*>(frag);
        float4* b4 = reinterpret_cast<float4*>(Bf);
        for (int i = tid; i < NP * 16; i += 256) b4[i] = f4[i];
    }
    const float bs = bias[li], g = gamma[li], bt = beta[li];
    const float* base[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int m = (wave * 2 + t) * 32 + li;                 // pixel of the band: row m / 35, column m % 35 (rows past the band: computed on pixel 0, never stored)
        if (m >= C32_ROWS * HW) m = 0;
        const int oy = m / HW, ox = m - oy * HW;
        base[t] = T + (oy * TW + ox) * PITCH + kh;
    }
    const float* bl = Bf + lane;
    // a thread's slots of the tile (four channels of one tile pixel each): where they come from relative to the band's first input row;
    // which of them lie in a border column (always zero), in the tile's first / last row (zero for a face's first / last band)
    int s_off[PER];
    unsigned m_col = 0, m_row0 = 0, m_row8 = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * 256;
        const int c4 = i & 7, px = i >> 3;
        const int ry = px / TW, cx = px - ry * TW;
        if ((i < NSLOT) && ((unsigned)(cx - 1) < (unsigned)HW)) m_col |= 1u << j;
        if (ry == 0) m_row0 |= 1u << j;
        if (ry == 8) m_row8 |= 1u << j;
        s_off[j] = (ry * HW + (cx - 1)) * 32 + 4 * c4;
    }
    float4 v[PER];
    float sk[2][16];
    auto request = [&](int item) {                        // the band's nine input rows, zero outside the image
        const int face = item / 5, band = item - face * 5;
        const float* src = in + ((size_t)face * HW + (band * C32_ROWS - 1)) * HW * 32;
        const unsigned ok = m_col & ~(band == 0 ? m_row0 : 0u) & ~(band == 4 ? m_row8 : 0u);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            v[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if ((ok >> j) & 1u) v[j] = *reinterpret_cast<const float4*>(src + s_off[j]);
        }
    };
    // the band's output (and skip) rows behind buffer descriptors: 245 pixels x 32 channels; the tile rows past the band (the last tile's
    // rows 21 .. 31) fall outside and are dropped / read as zero -- no lane-dependent branch around any access
    constexpr int BAND_BYTES = C32_ROWS * HW * 32 * 4;
    int e_off[2][16];                                     // byte offset of (this lane's channel, accumulator row) inside the band
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) e_off[t][reg] = (((wave * 2 + t) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh) * 32 + li) * 4;
    auto request_skip = [&](int item) {                   // the band's skip values: in flight during its own MFMAs
        const int face = item / 5, oy0 = (item - face * 5) * C32_ROWS;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(skip + ((size_t)face * HW + oy0) * HW * 32), 0, BAND_BYTES, 0x00020000);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) sk[t][reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, e_off[t][reg], 0, 0));
    };
    int item = blockIdx.x;
    if (item < n_items) request(item);
    while (item < n_items) {
        // park the band that was requested during the previous one's MFMAs
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * 256;
            if (i < NSLOT) {
                float* d = T + (i >> 3) * PITCH + 4 * (i & 7);
                d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
            }
        }
        __syncthreads();
        const int next = item + gridDim.x;
        if (next < n_items) request(next);
        if (skip) request_skip(item);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int tap = s >> 4, r = tap / 3, q = tap - r * 3;
            const int off = (r * TW + q) * PITCH + 2 * (s & 15);
            const float b = bl[s * 64];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(base[t][off], b, acc[t], 0, 0, 0);
        }
        // C layout of the 32 x 32 MFMA: column (channel) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); a band's pixels are
        // contiguous in the output tensor.  All 32 results are finished first (ONE wait for the skip values), then the 32 stores go out
        // back to back: written value by value the compiler waits before every store for everything in flight -- the next band's rows
        // included -- because the skip value it is about to add might still be on its way.
        const int face = item / 5, oy0 = (item - face * 5) * C32_ROWS;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + ((size_t)face * HW + oy0) * HW * 32), 0, BAND_BYTES, 0x00020000);
        float x[2][16];
        if (skip) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float y = ((acc[t][reg] + bs) * g + bt) + sk[t][reg];
                    x[t][reg] = y < 0.0f ? 0.0f : y;
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float y = (acc[t][reg] + bs) * g + bt;
                    x[t][reg] = y < 0.0f ? 0.0f : y;
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x[t][reg]), ro, e_off[t][reg], 0, 0);
        __syncthreads();                                  // every wave is done with the tile before the next band is parked
        item = next;
    }
}

__global__ void __launch_bounds__(256) maxpool3s2_k(const float* __restrict__ in, int B, int H, int W, int C, float* __restrict__ out, int OH, int OW)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * OH * OW * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t t = i / C;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    float mx = -INFINITY;
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) {
            const float v = in[(((size_t)b * H + oy * 2 + r) * W + ox * 2 + s) * C + c];
            mx = fmaxf(mx, v);
        }
    out[i] = mx;
}

// avg_pool_everything + fc_no_bias<128>: one block per face
__global__ void __launch_bounds__(256) head_k(const float* __restrict__ x, int HW, const float* __restrict__ fc, float* __restrict__ out)
{
    __shared__ float feat[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float s = 0.0f;
    for (int i = 0; i < HW; ++i) s += x[((size_t)b * HW + i) * 256 + t];
    feat[t] = s / (float)HW;
    __syncthreads();
    if (t < 128) {
        float acc = 0.0f;
        for (int o = 0; o < 256; ++o) acc += feat[o] * fc[(size_t)o * 128 + t];
        out[(size_t)b * 128 + t] = acc;
    }
}

static void launch_conv(Ctx* c, const ConvArgs& a)
{
    const long M = (long)a.B * a.OH * a.OW;
    PVF_REQUIRE(a.Cin % 32 == 0, "conv: input channels must be a multiple of 32 (the 3-channel first layer has a kernel of its own: stem_conv_k)");
    PVF_REQUIRE(a.OH * a.OW < (1 << 21) && a.Cout % 32 == 0, "conv: output map too large for the kernel's index arithmetic / Cout not a multiple of 32");
    if (a.frag && a.Cin == 32 && a.Cout == 32 && a.H == 35 && a.W == 35 && a.OH == 35 && a.OW == 35 && a.AH == 35 && a.AW == 35 && a.ksz == 3 &&
        a.stride == 1 && a.pad == 1 && a.relu && a.skip_mode != 2) {
        static std::atomic<uint64_t> attr_set{0};             // per device (a function attribute belongs to the device it was set on)
        const size_t lds = (size_t)(C32_TILE_FLOATS + 144 * 64) * sizeof(float);
        const uint64_t bit = 1ull << (c->device & 63);
        if (!(attr_set.load() & bit)) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set.fetch_or(bit);
        }
        const int items = a.B * (35 / C32_ROWS);
        hipLaunchKernelGGL(conv3x3_c32_k, dim3((unsigned)std::min(items, 2 * c->n_cu)), dim3(256), lds, c->stream, a.in, a.frag, a.bias, a.gamma, a.beta,
                           a.skip_mode == 1 ? a.skip : nullptr, a.out, items);
    } else if (a.Cout == 32) {
        const dim3 grid((unsigned)((M + 255) / 256), 1);
        hipLaunchKernelGGL((conv_mfma_k<4, 1>), grid, dim3(256), 0, c->stream, a);
    } else {
        PVF_REQUIRE(a.Cout % 64 == 0, "conv: Cout must be 32 or a multiple of 64");
        const dim3 grid((unsigned)((M + 127) / 128), a.Cout / 64);
        hipLaunchKernelGGL((conv_mfma_k<2, 2>), grid, dim3(256), 0, c->stream, a);
    }
}

// d_chips: [n][150][150][3] u8 on device; h_out [n][128]
void resnet_forward(Ctx* c, const uint8_t* d_chips, int n, float* h_out)
{
    const EmbedModel& e = c->emb;
    PVF_REQUIRE(e.loaded, "embedder not loaded");
    const int S = e.chip_size;
    const int MAXB = 4096;   // faces per forward: the deep layers (9x9 ... 2x2 maps) have few output rows per face, and a grid of one to two rounds of
                             // blocks leaves half the chip idle in its tail (measured at 1024: 1.2 rounds, matrix pipe 50 % busy); 6 GB of activations
    const int h1 = 1 + (S - 7) / 2;       // 72
    const int hp = 1 + (h1 - 3) / 2;      // 35
    const int cap = std::min(n, MAXB);                   // the arenas grow with the largest batch seen, not to the limit
    const size_t big = (size_t)cap * h1 * h1 * 32;
    c->s_act0.ensure(std::max(big, (size_t)cap * S * S * 4) * sizeof(float));
    c->s_act1.ensure(big * sizeof(float));
    c->s_act2.ensure((size_t)cap * hp * hp * 32 * sizeof(float) + (size_t)cap * 128 * sizeof(float));
    for (int b0 = 0; b0 < n; b0 += MAXB) {
        const int B = std::min(MAXB, n - b0);
        ProfScope ps(c, "conv");
        float* x = c->s_act0.as<float>();
        float* y = c->s_act1.as<float>();
        float* z = c->s_act2.as<float>();
        // conv1 (straight from the uint8 chips) -> y ; maxpool -> z
        const ConvLayer& L0 = e.convs[0];
        PVF_REQUIRE(S == 150 && L0.cout == 32 && L0.k == 7 && h1 == 72, "stem kernel: 150 x 150 chips, 7 x 7 stride 2, 32 channels");
        if (!c->emb.d_stem) {
            HIP_CHECK(hipMalloc(&c->emb.d_stem, STEM_NP * 64 * sizeof(float)));
            hipLaunchKernelGGL(stem_frag_k, dim3(STEM_NP), dim3(64), 0, c->stream, L0.d_w, c->emb.d_stem);
        }
        hipLaunchKernelGGL(stem_conv_k, dim3(h1 / STEM_ROWS, B), dim3(192), 0, c->stream, d_chips + (size_t)b0 * S * S * 3, B, c->emb.d_stem,
                           L0.d_bias, L0.d_gamma, L0.d_beta, y);
        ConvArgs a;
        const size_t npool = (size_t)B * hp * hp * 32;
        hipLaunchKernelGGL(maxpool3s2_k, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, c->stream, y, B, h1, h1, 32, z, hp, hp);
        // rotate buffers: cur = z (unit input), t1/t2 scratch
        float* cur = z; float* t1 = x; float* t2 = y;
        int H = hp, W = hp;
        static const int UN[14][3] = {{32, 32, 0}, {32, 32, 0}, {32, 32, 0}, {32, 64, 1}, {64, 64, 0}, {64, 64, 0}, {64, 64, 0},
                                      {64, 128, 1}, {128, 128, 0}, {128, 128, 0}, {128, 256, 1}, {256, 256, 0}, {256, 256, 0}, {256, 256, 1}};
        // fragment-ordered weights of the 32 -> 32 layers, made on first use
        auto frag_of = [&](int layer) -> const float* {
            ConvLayer& L = c->emb.convs[layer];
            if (L.cin != 32 || L.cout != 32 || L.k != 3) return nullptr;
            if (!L.d_frag) {
                HIP_CHECK(hipMalloc(&L.d_frag, 144 * 64 * sizeof(float)));
                hipLaunchKernelGGL(conv_frag_k, dim3(144), dim3(64), 0, c->stream, L.d_w, 288, L.d_frag);
            }
            return L.d_frag;
        };
        for (int u = 0; u < 14; ++u) {
            const int cin = UN[u][0], nn = UN[u][1], down = UN[u][2];
            const ConvLayer& La = e.convs[1 + 2 * u];
            const ConvLayer& Lb = e.convs[2 + 2 * u];
            const int stride = down ? 2 : 1, pad = down ? 0 : 1;
            const int ah = 1 + (H + 2 * pad - 3) / stride, aw = 1 + (W + 2 * pad - 3) / stride;
            memset(&a, 0, sizeof a);
            a.in = cur; a.B = B; a.H = H; a.W = W; a.Cin = cin; a.w = La.d_w; a.K = 9 * cin; a.bias = La.d_bias; a.gamma = La.d_gamma; a.beta = La.d_beta;
            a.out = t1; a.OH = ah; a.OW = aw; a.Cout = nn; a.AH = ah; a.AW = aw; a.ksz = 3; a.stride = stride; a.pad = pad; a.relu = 1; a.skip_mode = 0;
            a.frag = frag_of(1 + 2 * u);
            launch_conv(c, a);
            int sh = H, sw = W;
            if (down) { sh = 1 + (H - 2) / 2; sw = 1 + (W - 2) / 2; }
            const int oh = std::max(ah, sh), ow = std::max(aw, sw);
            memset(&a, 0, sizeof a);
            a.in = t1; a.B = B; a.H = ah; a.W = aw; a.Cin = nn; a.w = Lb.d_w; a.K = 9 * nn; a.bias = Lb.d_bias; a.gamma = Lb.d_gamma; a.beta = Lb.d_beta;
            a.out = t2; a.OH = oh; a.OW = ow; a.Cout = nn; a.AH = ah; a.AW = aw; a.ksz = 3; a.stride = 1; a.pad = 1; a.relu = 1;
            a.skip_mode = down ? 2 : 1; a.skip = cur; a.XH = H; a.XW = W; a.XC = cin; a.SH = sh; a.SW = sw;
            a.frag = frag_of(2 + 2 * u);
            launch_conv(c, a);
            float* old = cur; cur = t2; t2 = old;
            H = oh; W = ow;
        }
        float* d_out = reinterpret_cast<float*>(c->s_act2.as<uint8_t>() + (size_t)cap * hp * hp * 32 * sizeof(float));
        // cur may alias s_act2's front part; the embedding slot sits behind it
        hipLaunchKernelGGL(head_k, dim3(B), dim3(256), 0, c->stream, cur, H * W, e.d_fc, d_out);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(h_out + (size_t)b0 * 128, d_out, (size_t)B * 128 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

// dsst.hip -- K8: dlib.correlation_tracker (DSST) start_track / update, batched over trackers
// (reference pyannote/video/tracking.py:250-251 start_track, :203 update -> PSR, :165,231 get_position).
// Double precision like dlib; 64x64 2-D FFTs live in LDS (65-element row pitch), every sum keeps the order stated in
// oracle/pvo_dsst.c, exp() is the shared deterministic polynomial -- so PSR and positions are bit-identical.
#include "fhog_dev.h"
#include <cmath>

#define FS 64
#define NPL 32
#define NSC 32
#define SWIN 23
#define SDIM 512
#define LP 65 // LDS row pitch (complex elements)

static constexpr double REG_SPACE = 0.001, NU_SPACE = 0.025, REG_SCALE = 0.001, NU_SCALE = 0.025, ALPHA = 1.020;

struct TrkJob {
    double* state;   // A, B, As, Bs (may be shared by clones: only the full update / start write it)
    double* pos;     // this call's working position (4 doubles of scratch; starts as `box`)
    const uint8_t* img; int h, w;
    double map[4];   // chip (x,y) -> image (map0 + x*map2, map1 + y*map3)
    double cx, cy;   // start: object centre in chip coordinates
    double box[4];   // position the call starts from
};

__device__ __forceinline__ double det_exp(double x)
{
    const double inv_ln2 = 1.4426950408889634074, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    if (x < -700.0) return 0.0;
    const double kf = floor(x * inv_ln2 + 0.5);
    const double r = (x - kf * ln2_hi) - kf * ln2_lo;
    // Horner form of sum_{i<=13} r^i / i!  (same coefficients and order as the oracle's loop)
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600;
    p = p * r + 1.0 / 39916800;
    p = p * r + 1.0 / 3628800;
    p = p * r + 1.0 / 362880;
    p = p * r + 1.0 / 40320;
    p = p * r + 1.0 / 5040;
    p = p * r + 1.0 / 720;
    p = p * r + 1.0 / 120;
    p = p * r + 1.0 / 24;
    p = p * r + 1.0 / 6;
    p = p * r + 1.0 / 2;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const long long k = (long long)kf;
    const unsigned long long bits = (unsigned long long)(k + 1023) << 52;
    return p * __longlong_as_double((long long)bits);
}

__device__ __forceinline__ int brev6(int v) { return (int)(__brev((unsigned)v) >> 26); }

// One radix-2 DIT butterfly exactly as the oracle writes it: t = w * b ; (a, b) <- (a + t, a - t)
__device__ __forceinline__ void bfly(double2& a, double2& b, double wr, double wi)
{
    const double tr = wr * b.x - wi * b.y;
    const double ti = wr * b.y + wi * b.x;
    const double2 u = a;
    a = make_double2(u.x + tr, u.y + ti);
    b = make_double2(u.x - tr, u.y - ti);
}

// In-place 2-D FFT of s[64][LP] (double2 = re,im); rows then columns; blockDim.x = 256.
// The 64-point radix-2 DIT of each line is evaluated as two register-resident groups of three stages:
//   phase 1: lane owns x[8q .. 8q+7] of the bit-reversed line        -> stages 1-3 (pairs at distance 1, 2, 4)
//   phase 2: lane owns x[8a + b], a = 0..7                            -> stages 4-6 (pairs at distance 8, 16, 32)
// Butterflies, operand order and twiddles are those of the stage-by-stage form (oracle/pvo_dsst.c fft1d), so values are
// bit-identical.  Lane <-> task: phase 1 gives consecutive lanes consecutive q (their 16-byte elements are neighbours along a row,
// 65 elements = 4 banks apart along a column); phase 2 gives them consecutive LINES of one b -- with consecutive b the elements of
// the lanes that LDS serves together lie 8 (rows) or 8 * 65 (columns) elements = a multiple of 32 banks apart, an 8-way conflict on
// every ds_read/write_b128 that made the transform LDS-bound (measured: 2.9 -> ... ms per 2000 trackers).
// Everything is done IN PLACE on the positions a lane reads (no intra-phase barrier, 8 live values per lane):
// element x_br[k] never moves from position brev6(k), hence the OUTPUT IS LEFT IN BIT-REVERSED POSITIONS along both axes:
//   X[kr][kc]  is found at  s[brev6(kr) * LP + brev6(kc)]            (use FFT_AT below).
#define FFT_AT(s, r, c) (s)[brev6(r) * LP + brev6(c)]
template <int NT = 256>     // threads of the block: 256 (two line tasks per thread and phase) or 512 (one)
__device__ __forceinline__ void fft2d_lds(double2* s, const double* __restrict__ tw, bool inverse)
{
    const int tid = threadIdx.x;
    const double sg = inverse ? 1.0 : -1.0;      // wi = +sin (inverse) / -sin (forward)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // ---- phase 1: task (line, q): positions brev6(8q + i) = 8 * brev3(i) + brev3(q)
#pragma unroll
        for (int u = 0; u < 512 / NT; ++u) {
            const int tt = tid + NT * u, line = tt >> 3, q = tt & 7;
            const int rq = (int)(__brev((unsigned)q) >> 29);
            double2 e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ri = ((i & 1) << 2) | (i & 2) | ((i >> 2) & 1);
                const int pos = ri * 8 + rq;
                e[i] = s[pass == 0 ? line * LP + pos : pos * LP + line];
            }
            {
                const double wr = tw[0], wi = sg * tw[1];                         // stage 1 (m = 2)
                bfly(e[0], e[1], wr, wi); bfly(e[2], e[3], wr, wi); bfly(e[4], e[5], wr, wi); bfly(e[6], e[7], wr, wi);
                const double w1r = tw[2 * 16], w1i = sg * tw[2 * 16 + 1];        // stage 2 (m = 4, tstep 16)
                bfly(e[0], e[2], wr, wi); bfly(e[1], e[3], w1r, w1i);
                bfly(e[4], e[6], wr, wi); bfly(e[5], e[7], w1r, w1i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bfly(e[j], e[j + 4], tw[2 * 8 * j], sg * tw[2 * 8 * j + 1]);   // stage 3 (m = 8, tstep 8)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ri = ((i & 1) << 2) | (i & 2) | ((i >> 2) & 1);
                const int pos = ri * 8 + rq;
                s[pass == 0 ? line * LP + pos : pos * LP + line] = e[i];
            }
        }
        __syncthreads();
        // ---- phase 2: task (line, b): x[8a + b] lives at brev6(8a + b) = 8 * brev3(b) + brev3(a)
#pragma unroll
        for (int u = 0; u < 512 / NT; ++u) {
            const int tt = tid + NT * u, line = tt & 63, b = tt >> 6;     // a wave = the 64 lines of one b (see the bank note above)
            const int rb = (int)(__brev((unsigned)b) >> 29);
            double2 f[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int ra = ((a & 1) << 2) | (a & 2) | ((a >> 2) & 1);
                const int pos = rb * 8 + ra;
                f[a] = s[pass == 0 ? line * LP + pos : pos * LP + line];
            }
            {
                const double wr = tw[2 * 4 * b], wi = sg * tw[2 * 4 * b + 1];     // stage 4 (m = 16, tstep 4): j = b
                bfly(f[0], f[1], wr, wi); bfly(f[2], f[3], wr, wi); bfly(f[4], f[5], wr, wi); bfly(f[6], f[7], wr, wi);
                const double w0r = tw[2 * 2 * b], w0i = sg * tw[2 * 2 * b + 1];   // stage 5 (m = 32, tstep 2): j = 8 (a & 1) + b
                const double w1r = tw[2 * 2 * (8 + b)], w1i = sg * tw[2 * 2 * (8 + b) + 1];
                bfly(f[0], f[2], w0r, w0i); bfly(f[1], f[3], w1r, w1i);
                bfly(f[4], f[6], w0r, w0i); bfly(f[5], f[7], w1r, w1i);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) bfly(f[a], f[a + 4], tw[2 * (8 * a + b)], sg * tw[2 * (8 * a + b) + 1]);   // stage 6: j = 8a + b
            if (inverse && pass == 1) {
                const double sc = 1.0 / 64.0;
#pragma unroll
                for (int a = 0; a < 8; ++a) { f[a].x = (f[a].x * sc) * sc; f[a].y = (f[a].y * sc) * sc; }
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int ra = ((a & 1) << 2) | (a & 2) | ((a >> 2) & 1);
                const int pos = rb * 8 + ra;
                s[pass == 0 ? line * LP + pos : pos * LP + line] = f[a];
            }
        }
        __syncthreads();
    }
}

// sequential 32-point FFT on x[32] (double2), one thread
__device__ __forceinline__ void fft32_seq(double2* x, const double* __restrict__ tw, bool inverse)
{
    for (int i = 0; i < NSC; ++i) {
        const int j = (int)(__brev((unsigned)i) >> 27);
        if (j > i) { const double2 t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (int st = 1; st <= 5; ++st) {
        const int m = 1 << st, half = m >> 1, tstep = NSC / m;
        for (int k = 0; k < NSC; k += m)
            for (int j = 0; j < half; ++j) {
                const double wr = tw[2 * j * tstep], wi = inverse ? tw[2 * j * tstep + 1] : -tw[2 * j * tstep + 1];
                const double2 a = x[k + j], b = x[k + j + half];
                const double tr = wr * b.x - wi * b.y;
                const double ti = wr * b.y + wi * b.x;
                x[k + j] = make_double2(a.x + tr, a.y + ti);
                x[k + j + half] = make_double2(a.x - tr, a.y - ti);
            }
    }
    if (inverse) {
        const double sc = 1.0 / NSC;
        for (int i = 0; i < NSC; ++i) { x[i].x *= sc; x[i].y *= sc; }
    }
}

// ---- translation features: plane i of tracker b -> mask * value -> s (natural positions).  The features arrive as the compact
// record of fhog1_compact_k (fhog.hip: float S[4096], float T[4][4096], uint8 bin[4096]; 84 KB per tracker, re-read per plane out of
// L2): orientation plane i < 18 is S where the pixel's bin is i, plane 18 + j is S where bin % 9 is j, planes 27..30 are the textures,
// plane 31 the grey chip -- the values the 31-plane form of round 4 held, which never exist in memory now.
__device__ __forceinline__ float plane_value(const uint8_t* __restrict__ rec, const uint8_t* __restrict__ chip, int i, int q)
{
    const float* S = reinterpret_cast<const float*>(rec);
    const int sl = TRKF_SLOT(q);
    if (i < 27) {
        const int a = rec[TRKF_A + sl];
        const int key = i < 18 ? a : (a >= 9 ? a - 9 : a);
        const float v = S[sl];
        return key == (i < 18 ? i : i - 18) ? v : 0.0f;
    }
    if (i < 31) return S[(size_t)(i - 26) * TRKF_PLANE + sl];
    const uint8_t* p = chip + (size_t)q * 3;
    return (float)(((unsigned)p[0] + p[1] + p[2]) / 3) / 255.0f;
}

template <int NT = 256>
__device__ __forceinline__ void load_plane_lds(double2* s, const uint8_t* __restrict__ chip, const uint8_t* __restrict__ rec, int i,
                                               const double* __restrict__ mask64)
{
#pragma unroll
    for (int k = 0; k < FS * FS / NT; ++k) {
        const int q = threadIdx.x + NT * k;
        const float v = plane_value(rec, chip, i, q);
        s[(q >> 6) * LP + (q & 63)] = make_double2((double)v * mask64[q], 0.0);
    }
}

// plane spectra to HBM: only the full (filter-updating) update needs them twice, see dsst_update_many
__global__ void __launch_bounds__(256) trans_planes_fft_k(const uint8_t* __restrict__ chips, const uint8_t* __restrict__ feat,
                                                          const double* __restrict__ mask64, const double* __restrict__ tw64,
                                                          double2* __restrict__ F)
{
    extern __shared__ __attribute__((aligned(16))) double2 s[];
    const int i = blockIdx.x, b = blockIdx.y;
    load_plane_lds(s, chips + (size_t)b * FS * FS * 3, feat + (size_t)b * TRKF_BYTES, i, mask64);
    __syncthreads();
    fft2d_lds(s, tw64, false);
    double2* out = F + ((size_t)b * NPL + i) * FS * FS;
    for (int q = threadIdx.x; q < FS * FS; q += 256) out[q] = FFT_AT(s, q >> 6, q & 63);
}

// target image exp(-dist/3) in a 21x21 window around (px,py), FFT, conj; spectrum left in s in bit-reversed positions (FFT_AT)
template <int NT = 256>
__device__ __forceinline__ void make_target_lds(double2* s, double px, double py, const double* __restrict__ tw64)
{
    for (int q = threadIdx.x; q < FS * FS; q += NT) s[(q >> 6) * LP + (q & 63)] = make_double2(0.0, 0.0);
    __syncthreads();
    const long cx = (long)floor(px + 0.5), cy = (long)floor(py + 0.5);
    for (int q = threadIdx.x; q < 21 * 21; q += NT) {
        const long r = cy - 10 + q / 21, c = cx - 10 + q % 21;
        if (r < 0 || c < 0 || r > FS - 1 || c > FS - 1) continue;
        const double dx = (double)c - px, dy = (double)r - py;
        const double dist = sqrt(dx * dx + dy * dy);
        s[r * LP + c] = make_double2(det_exp(-dist / 3.0), 0.0);
    }
    __syncthreads();
    fft2d_lds<NT>(s, tw64, false);
    for (int q = threadIdx.x; q < FS * FS; q += NT) { const int p = (q >> 6) * LP + (q & 63); s[p].y = -s[p].y; }
    __syncthreads();
}

__global__ void __launch_bounds__(256) corr_k(const TrkJob* __restrict__ jobs, const double2* __restrict__ F, double2* __restrict__ Gfreq)
{
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const double* st = jobs[b].state;
    const double2* A = reinterpret_cast<const double2*>(st + TRK_A);
    double gr = 0, gi = 0;
    for (int i = 0; i < NPL; ++i) {
        const double2 f = F[((size_t)b * NPL + i) * FS * FS + q];
        const double2 a = A[(size_t)i * FS * FS + q];
        gr = gr + (f.x * a.x + f.y * a.y);
        gi = gi + (f.y * a.x - f.x * a.y);
    }
    const double rec = 1.0 / (st[TRK_B + q] + REG_SPACE);
    Gfreq[(size_t)b * FS * FS + q] = make_double2(gr * rec, gi * rec);
}

// response spectrum in s (natural positions) -> response -> peak, PSR, new position; then (Ghat != nullptr) the new target's spectrum
template <int NT = 256>
__device__ __forceinline__ void peak_body(double2* s, const TrkJob& j, int b, const double* __restrict__ tw64, double* __restrict__ results /* [n][8] */,
                          double2* __restrict__ Ghat)
{
    __shared__ double red_v[NT];
    __shared__ int red_i[NT];
    __shared__ double row_s[FS], row_q[FS], row_c[FS];
    __shared__ double pk[4];
    const int tid = threadIdx.x;
    fft2d_lds<NT>(s, tw64, true);
    // arg-max of the real part, first occurrence in row-major order
    double bv = -INFINITY; int bi = 0x7fffffff;
    for (int q = tid; q < FS * FS; q += NT) {
        const double v = FFT_AT(s, q >> 6, q & 63).x;
        if (v > bv) { bv = v; bi = q; }
    }
    red_v[tid] = bv; red_i[tid] = bi;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const double ov = red_v[tid + off]; const int oi = red_i[tid + off];
            if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int py = red_i[0] >> 6, px = red_i[0] & 63;
        double ox = px, oy = py;
        if (!(px < 1 || py < 1 || px > FS - 2 || py > FS - 2)) {
            double z[3][3];
            for (int r = -1; r <= 1; ++r) for (int c = -1; c <= 1; ++c) z[r + 1][c + 1] = FFT_AT(s, py + r, px + c).x;
            const double sx = ((z[0][2] + z[1][2]) + z[2][2]) - ((z[0][0] + z[1][0]) + z[2][0]);
            const double sy = ((z[2][0] + z[2][1]) + z[2][2]) - ((z[0][0] + z[0][1]) + z[0][2]);
            const double sxy = (z[0][0] + z[2][2]) - (z[0][2] + z[2][0]);
            const double sxx = ((z[0][0] + z[1][0]) + z[2][0]) + ((z[0][2] + z[1][2]) + z[2][2]);
            const double syy = ((z[0][0] + z[0][1]) + z[0][2]) + ((z[2][0] + z[2][1]) + z[2][2]);
            const double sall = sxx + ((z[0][1] + z[1][1]) + z[2][1]);
            const double k2 = sx / 6.0, k3 = sy / 6.0, k5 = sxy / 4.0;
            const double k4 = sxx / 2.0 - sall / 3.0, k6 = syy / 2.0 - sall / 3.0;
            const double h00 = 2 * k4, h01 = k5, h11 = 2 * k6;
            const double det = h00 * h11 - h01 * h01;
            if (det != 0) {
                double dx = -((h11 * k2 - h01 * k3) / det);
                double dy = -((h00 * k3 - h01 * k2) / det);
                if (!(dx * k2 + dy * k3 < 0)) {
                    if (dx < -1) dx = -1;
                    if (dx > 1) dx = 1;
                    if (dy < -1) dy = -1;
                    if (dy > 1) dy = 1;
                    ox = px + dx; oy = py + dy;
                }
            }
        }
        pk[0] = ox; pk[1] = oy;
    }
    __syncthreads();
    const double ppx = pk[0], ppy = pk[1];
    const long rx = (long)floor(ppx + 0.5), ry = (long)floor(ppy + 0.5);
    if (tid < FS) {
        const int r = tid;
        double rs = 0, rq = 0, cnt = 0;
        for (int c = 0; c < FS; ++c) {
            if (c >= rx - 4 && c <= rx + 3 && r >= ry - 4 && r <= ry + 3) continue;
            const double v = FFT_AT(s, r, c).x;
            rs = rs + v;
            rq = rq + v * v;
            cnt += 1;
        }
        row_s[r] = rs; row_q[r] = rq; row_c[r] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        double sum = 0, sumsq = 0, cnt = 0;
        for (int r = 0; r < FS; ++r) { sum = sum + row_s[r]; sumsq = sumsq + row_q[r]; cnt += row_c[r]; }
        const double mean = sum / cnt;
        double var = (1.0 / (cnt - 1)) * (sumsq - sum * sum / cnt);
        if (!(var >= 0)) var = 0;
        long qx = rx, qy = ry;
        if (qx < 0) qx = 0;
        if (qy < 0) qy = 0;
        if (qx > FS - 1) qx = FS - 1;
        if (qy > FS - 1) qy = FS - 1;
        const double psr = (FFT_AT(s, (int)qy, (int)qx).x - mean) / sqrt(var);
        double* ps = j.pos;
        const double g0 = ps[0], g1 = ps[1], g2 = ps[2], g3 = ps[3];
        const double ix = j.map[0] + ppx * j.map[2], iy = j.map[1] + ppy * j.map[3];
        const double vx = ix - (g0 + g2) / 2, vy = iy - (g1 + g3) / 2;
        ps[0] = g0 + vx; ps[1] = g1 + vy; ps[2] = g2 + vx; ps[3] = g3 + vy;
        results[(size_t)b * 8] = psr;
        results[(size_t)b * 8 + 5] = ppx; results[(size_t)b * 8 + 6] = ppy;
    }
    if (!Ghat) return;                          // block-uniform
    __syncthreads();
    make_target_lds<NT>(s, ppx, ppy, tw64);
    double2* out = Ghat + (size_t)b * FS * FS;
    for (int q = tid; q < FS * FS; q += NT) out[q] = FFT_AT(s, q >> 6, q & 63);
}

__global__ void __launch_bounds__(256) peak_k(const TrkJob* __restrict__ jobs, const double2* __restrict__ Gfreq, const double* __restrict__ tw64,
                                              double* __restrict__ results, double2* __restrict__ Ghat)
{
    extern __shared__ __attribute__((aligned(16))) double2 s[];
    const int b = blockIdx.x;
    const TrkJob j = jobs[b];
    for (int q = threadIdx.x; q < FS * FS; q += 256) s[(q >> 6) * LP + (q & 63)] = Gfreq[(size_t)b * FS * FS + q];
    __syncthreads();
    peak_body(s, j, b, tw64, results, Ghat);
}

// ---- start_track in one pass per tracker: the block walks the 32 planes; a plane's spectrum goes straight from LDS into
// A_i = G * F_i and into the running |F|^2 sum (plane order) and is never written out.
// HBM per tracker: 84 KB of compact features in (round 4: 0.5 MB of planes), 2.06 MB of filters out (three-kernel form: + 2 MB F
// written and read, + G).
// 512 threads: one line task per thread and FFT phase, 8 spectrum points per thread -- four waves per SIMD hide the LDS and
// fp64 latencies of the butterfly chains (a 256-thread form needed > 256 registers and ran one wave per SIMD).
#define FUSED_NT 512
#define FUSED_PT (FS * FS / FUSED_NT)
// A thread owns the spectrum points at the LDS positions ph = tid + 512 k, k = 0..7: row (tid >> 6) + 8 k, column tid & 63 (consecutive
// lanes = neighbouring elements, no bank conflict).  Position (pr, pc) holds X[brev6(pr)][brev6(pc)]; with w = tid >> 6 that is
// logical row 8 brev3(w) + brev3(k), column brev6(tid & 63): the 8 points of a thread are q0 + 64 brev3(k) -- constant offsets from
// one address, and a wave's 64 points of one k are one 1 KB row of A (permuted).
struct FusedOwn { int lds0; int q0; };
__device__ __forceinline__ FusedOwn fused_own()
{
    const int tid = threadIdx.x, w = tid >> 6;
    const int b3 = ((w & 1) << 2) | (w & 2) | ((w >> 2) & 1);
    return FusedOwn{w * LP + (tid & 63), b3 * 8 * FS + brev6(tid & 63)};
}
#define FUSED_LDS(k) (8 * LP * (k))                                                        // k-th owned LDS position, from lds0
#define FUSED_Q(k) (FS * ((((k) & 1) << 2) | ((k) & 2) | (((k) >> 2) & 1)))                // k-th owned logical index, from q0

// plane i of the tracker's features -> mask * value -> s.  All eight loads of a thread are issued before the first is used
// (the plane test is block-uniform: no branch inside the element loop).
__device__ __forceinline__ void fused_load_plane(double2* s, const uint8_t* __restrict__ chip, const uint8_t* __restrict__ rec, int i,
                                                 const double* __restrict__ mask64)
{
    const int tid = threadIdx.x;
    float v[FUSED_PT];
    double m[FUSED_PT];
    static_assert(FUSED_NT == 512 && FUSED_PT == 8, "the record's slot order is made for 512 threads x 8 pixels");
    if (i < 31) {
        // the thread's eight pixels tid + 512 k: 32 consecutive bytes of a float array, 8 of the bins (TRKF_SLOT)
        const float4* pl = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rec) + (i < 27 ? 0 : (size_t)(i - 26) * TRKF_PLANE) + tid * 8);
        const float4 lo = pl[0], hi = pl[1];
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        if (i < 27) {
            const uint2 ab = *reinterpret_cast<const uint2*>(rec + TRKF_A + tid * 8);
            const int want = i < 18 ? i : i - 18;
#pragma unroll
            for (int k = 0; k < FUSED_PT; ++k) {
                const int a = (int)(((k < 4 ? ab.x : ab.y) >> (8 * (k & 3))) & 0xffu);
                const int key = i < 18 ? a : (a >= 9 ? a - 9 : a);
                v[k] = key == want ? v[k] : 0.0f;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < FUSED_PT; ++k) {
            const uint8_t* p = chip + (size_t)(tid + FUSED_NT * k) * 3;
            v[k] = (float)(((unsigned)p[0] + p[1] + p[2]) / 3) / 255.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) m[k] = mask64[tid + FUSED_NT * k];
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) s[(tid >> 6) * LP + (tid & 63) + FUSED_LDS(k)] = make_double2((double)v[k] * m[k], 0.0);
}

__global__ void __launch_bounds__(FUSED_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) start_fused_k(const TrkJob* __restrict__ jobs, const uint8_t* __restrict__ chips, const uint8_t* __restrict__ feat,
                                                          const double* __restrict__ mask64, const double* __restrict__ tw64)
{
    extern __shared__ __attribute__((aligned(16))) double2 s[];
    __shared__ double tw_lds[2 * FS];           // twiddles next to the data: re-read after every barrier instead of pinned in registers
    const int b = blockIdx.x, tid = threadIdx.x;
    const TrkJob j = jobs[b];
    if (tid < 2 * FS) tw_lds[tid] = tw64[tid];
    __syncthreads();
    tw64 = tw_lds;
    make_target_lds<FUSED_NT>(s, j.cx, j.cy, tw64);
    const FusedOwn own = fused_own();
    double2 g[FUSED_PT];
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) g[k] = s[own.lds0 + FUSED_LDS(k)];
    __syncthreads();
    double bsum[FUSED_PT];
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) bsum[k] = 0;
    double2* A = reinterpret_cast<double2*>(j.state + TRK_A) + own.q0;
    const uint8_t* chip = chips + (size_t)b * FS * FS * 3;
    const uint8_t* fb = feat + (size_t)b * TRKF_BYTES;
    for (int i = 0; i < NPL; ++i) {
        const double* mk = mask64;
        asm volatile("" : "+s"(mk));               // the window is re-read per plane (L1/L2 hits), not kept in 16 registers across the loop
        fused_load_plane(s, chip, fb, i, mk);
        __syncthreads();
        fft2d_lds<FUSED_NT>(s, tw64, false);
        double2* Ai = A + (size_t)i * FS * FS;
#pragma unroll
        for (int k = 0; k < FUSED_PT; ++k) {
            const double2 f = s[own.lds0 + FUSED_LDS(k)];
            Ai[FUSED_Q(k)] = make_double2(g[k].x * f.x - g[k].y * f.y, g[k].x * f.y + g[k].y * f.x);
            bsum[k] = bsum[k] + (f.x * f.x + f.y * f.y);
        }
        __syncthreads();
    }
    double* Bq = j.state + TRK_B + own.q0;
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) Bq[FUSED_Q(k)] = bsum[k];
}

// ---- deferred update in one pass per tracker: plane spectra are multiplied into the response sum as they appear (plane order,
// like corr_k), then normalised, inverted and searched in the same block.  Filters are only read (2.06 MB per tracker).
__global__ void __launch_bounds__(FUSED_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) update_fused_k(const TrkJob* __restrict__ jobs, const uint8_t* __restrict__ chips, const uint8_t* __restrict__ feat,
                                                           const double* __restrict__ mask64, const double* __restrict__ tw64, double* __restrict__ results)
{
    extern __shared__ __attribute__((aligned(16))) double2 s[];
    __shared__ double tw_lds[2 * FS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const TrkJob j = jobs[b];
    if (tid < 2 * FS) tw_lds[tid] = tw64[tid];
    __syncthreads();
    tw64 = tw_lds;
    const FusedOwn own = fused_own();
    const double2* A = reinterpret_cast<const double2*>(j.state + TRK_A) + own.q0;
    const uint8_t* chip = chips + (size_t)b * FS * FS * 3;
    const uint8_t* fb = feat + (size_t)b * TRKF_BYTES;
    double gr[FUSED_PT], gi[FUSED_PT];
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) { gr[k] = 0; gi[k] = 0; }
    for (int i = 0; i < NPL; ++i) {
        const double* mk = mask64;
        asm volatile("" : "+s"(mk));
        fused_load_plane(s, chip, fb, i, mk);
        __syncthreads();
        fft2d_lds<FUSED_NT>(s, tw64, false);
        const double2* Ai = A + (size_t)i * FS * FS;
        double2 a[FUSED_PT];
#pragma unroll
        for (int k = 0; k < FUSED_PT; ++k) a[k] = Ai[FUSED_Q(k)];         // all eight in flight together
#pragma unroll
        for (int k = 0; k < FUSED_PT; ++k) {
            const double2 f = s[own.lds0 + FUSED_LDS(k)];
            gr[k] = gr[k] + (f.x * a[k].x + f.y * a[k].y);
            gi[k] = gi[k] + (f.y * a[k].x - f.x * a[k].y);
        }
        __syncthreads();
    }
    const double* Bq = j.state + TRK_B + own.q0;
#pragma unroll
    for (int k = 0; k < FUSED_PT; ++k) {
        const int q = own.q0 + FUSED_Q(k);
        const double rec = 1.0 / (Bq[FUSED_Q(k)] + REG_SPACE);
        s[(q >> 6) * LP + (q & 63)] = make_double2(gr[k] * rec, gi[k] * rec);       // natural positions for the inverse transform
    }
    __syncthreads();
    peak_body<FUSED_NT>(s, j, b, tw64, results, nullptr);
}

__global__ void __launch_bounds__(256) filter_update_k(const TrkJob* __restrict__ jobs, const double2* __restrict__ F, const double2* __restrict__ Ghat)
{
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    double* st = jobs[b].state;
    double2* A = reinterpret_cast<double2*>(st + TRK_A);
    const double2 g = Ghat[(size_t)b * FS * FS + q];
    double bq = st[TRK_B + q] * (1 - NU_SPACE);
    for (int i = 0; i < NPL; ++i) {
        const double2 f = F[((size_t)b * NPL + i) * FS * FS + q];
        const double2 a = A[(size_t)i * FS * FS + q];
        const double nr = g.x * f.x - g.y * f.y;
        const double ni = g.x * f.y + g.y * f.x;
        A[(size_t)i * FS * FS + q] = make_double2(NU_SPACE * nr + (1 - NU_SPACE) * a.x, NU_SPACE * ni + (1 - NU_SPACE) * a.y);
        bq = bq + NU_SPACE * (f.x * f.x + f.y * f.y);
    }
    st[TRK_B + q] = bq;
}

__device__ __forceinline__ void scale_rect_d(double r[4], double s)
{
    const double cx = (r[0] + r[2]) / 2, cy = (r[1] + r[3]) / 2;
    const double w = (r[2] - r[0]) * s, h = (r[3] - r[1]) * s;
    r[0] = cx - w / 2; r[1] = cy - h / 2; r[2] = cx + w / 2; r[3] = cy + h / 2;
}

// ---- scale samples: chip, gradients, histograms and features of one 23 x 23 sample in ONE wave, nothing in HBM but the 16 x 32 floats
// the scale transform reads (round 4: scale_chips_k -> fhog_grad4_k<4> -> fhog_hist_k<4> -> fhog_feat_k, the chip, its (magnitude, bin)
// planes and its cell histograms written to HBM and read back: 0.3 MB per tracker and call, 2.7 ms per 2000 trackers).
//   1. (SAMPLE) the wave samples its chip from the frame: 32 rectangles around the tracker's position, the arithmetic of dlib's
//      pyramid-free extract_image_chip path exactly as the oracle states it (oracle/pvo_dsst.c scale_sample) -> bytes in LDS;
//      (!SAMPLE: stage access of the parity tests, the chip is given)
//   2. gradient magnitude + orientation bin of the pixels that vote (1 .. 21 of 23: visible = min(6 * 4, 23) - 1), into a 36 x 36 plane
//      shifted by 3 * 4 / 2 = 6 so that histogram cell (hy, hx) owns rows 4 hy .. 4 hy + 7, columns 4 hx .. 4 hx + 7
//   3. lane = histogram cell (8 x 8 cells = 64 lanes): its 64 votes in row-major order into acc[bin][lane] -- the order dlib's scatter
//      loop adds in (oracle/pvo_fhog.c) -- then the cell's energy
//   4. lanes 0..15: cell_features() of the 4 x 4 output cells; slot 31 of a cell (the pad of the 32-float record) carries the grey
//      value of chip pixel (cell row, cell column), the 32nd "plane" of the scale filter (SAMPLE only; 0 for stage access)
// LDS per wave: 9.9 KB (16 waves per CU): the plane is 32 x 32 -- only rows / columns 7 .. 27 ever hold a vote, a cell window's
// coordinates 32 .. 35 read row / column 31 (zero) instead; the chip's bytes share their room with the bins' sums (dead by then).
#define SPL 32
template <bool SAMPLE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
scale_fhog_k(const TrkJob* __restrict__ jobs, double alpha_pow_m16, const uint8_t* __restrict__ chips_in, const uint8_t* __restrict__ lut, float* __restrict__ feat)
{
    __shared__ __attribute__((aligned(16))) float acc_s[4][18][64];          // first: the chip's 23 x 23 x 3 bytes
    __shared__ float mag_s[4][SPL * SPL];
    __shared__ uint8_t bin_s[4][SPL * SPL];
    __shared__ float nrm_s[4][36];
    static_assert(sizeof(float) * 18 * 64 >= SWIN * SWIN * 3, "the chip fits where the sums will be");
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + w, b = blockIdx.y;
    uint8_t* chip = reinterpret_cast<uint8_t*>(&acc_s[w][0][0]);
    float* mag = mag_s[w];
    uint8_t* bin = bin_s[w];
    for (int q = lane; q < SPL * SPL; q += 64) { mag[q] = 0.0f; bin[q] = 0; }
    if (SAMPLE) {
        const TrkJob& j = jobs[b];
        const double* jp = j.pos;
        const uint8_t* img = j.img;
        const int jw = j.w, jh = j.h;
        double ppp[4] = {jp[0], jp[1], jp[2], jp[3]};
        scale_rect_d(ppp, alpha_pow_m16);
        for (int i = 0; i < k; ++i) scale_rect_d(ppp, ALPHA);
        const double m0 = (ppp[2] - ppp[0]) / (double)(SWIN - 1), m3 = (ppp[3] - ppp[1]) / (double)(SWIN - 1);
        for (int q = lane; q < SWIN * SWIN; q += 64) {
            const int r = q / SWIN, c = q % SWIN;
            const double px = m0 * c + 0.0 * r + ppp[0];
            const double py = 0.0 * c + m3 * r + ppp[1];
            const double fx = floor(px), fy = floor(py);
            uint8_t* o = chip + q * 3;
            if (!(fx >= 0 && fy >= 0 && fx + 1 < jw && fy + 1 < jh)) { o[0] = 0; o[1] = 0; o[2] = 0; continue; }
            const int left = (int)fx, top = (int)fy;
            const double lr = px - left, tb = py - top;
            // the two pixels of a source row are 6 consecutive bytes: two (unaligned) dwords at +0 and +2 cover exactly them
            const uint8_t* ptl = img + ((size_t)top * jw + left) * 3;
            const uint8_t* pbl = ptl + (size_t)jw * 3;
            const uint32_t t0 = *reinterpret_cast<const uint32_t*>(ptl), t1 = *reinterpret_cast<const uint32_t*>(ptl + 2);
            const uint32_t b0 = *reinterpret_cast<const uint32_t*>(pbl), b1 = *reinterpret_cast<const uint32_t*>(pbl + 2);
            const uint32_t tl3[3] = {t0 & 0xffu, (t0 >> 8) & 0xffu, (t0 >> 16) & 0xffu}, tr3[3] = {t0 >> 24, (t1 >> 16) & 0xffu, t1 >> 24};
            const uint32_t bl3[3] = {b0 & 0xffu, (b0 >> 8) & 0xffu, (b0 >> 16) & 0xffu}, br3[3] = {b0 >> 24, (b1 >> 16) & 0xffu, b1 >> 24};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const double tl = tl3[ch], tr = tr3[ch], bl = bl3[ch], br = br3[ch];
                const double v = (1 - tb) * ((1 - lr) * tl + lr * tr) + tb * ((1 - lr) * bl + lr * br);
                o[ch] = (uint8_t)v;
            }
        }
    } else {
        const uint8_t* src = chips_in + ((size_t)b * NSC + k) * SWIN * SWIN * 3;
        for (int q = lane; q < SWIN * SWIN * 3; q += 64) chip[q] = src[q];
    }
    __syncthreads();
    float grey = 0.0f;
    if (SAMPLE && lane < 16) {
        const uint8_t* p = chip + ((lane >> 2) * SWIN + (lane & 3)) * 3;
        grey = (float)(((unsigned)p[0] + p[1] + p[2]) / 3) / 255.0f;
    }
    for (int q = lane; q < 21 * 21; q += 64) {
        const int y = 1 + q / 21, x = 1 + q % 21;
        float v2; int o;
        pixel_grad(chip + (y - 1) * SWIN * 3, chip + y * SWIN * 3, chip + (y + 1) * SWIN * 3, 3 * x, lut, &v2, &o);
        mag[(y + 6) * SPL + x + 6] = sqrtf(v2);
        bin[(y + 6) * SPL + x + 6] = (uint8_t)o;
    }
    __syncthreads();                                   // the chip is dead: its room becomes the sums
#pragma unroll
    for (int o = 0; o < 18; ++o) acc_s[w][o][lane] = 0.0f;
    {
        const int hy = lane >> 3, hx = lane & 7;
        int cx[8];
#pragma unroll
        for (int wx = 0; wx < 8; ++wx) cx[wx] = min(4 * hx + wx, SPL - 1);
#pragma unroll
        for (int wy = 0; wy < 8; ++wy) {
            const float fy = ((float)(wy % 4) + 0.5f) / 4.0f;
            const float wyv = (wy < 4) ? fy : 1.0f - fy;
            const int row = min(4 * hy + wy, SPL - 1) * SPL;
#pragma unroll
            for (int wx = 0; wx < 8; ++wx) {
                const float fx = ((float)(wx % 4) + 0.5f) / 4.0f;
                const float wxv = (wx < 4) ? fx : 1.0f - fx;
                const int o = bin[row + cx[wx]];
                acc_s[w][o][lane] = acc_s[w][o][lane] + (wyv * wxv) * mag[row + cx[wx]];
            }
        }
        float e = 0.0f;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const float s2 = acc_s[w][o][lane] + acc_s[w][o + 9][lane];
            e = e + s2 * s2;
        }
        if (hy >= 1 && hy <= 6 && hx >= 1 && hx <= 6) nrm_s[w][(hy - 1) * 6 + (hx - 1)] = e;
    }
    __syncthreads();
    if (lane < 16) {
        const int y = lane >> 2, x = lane & 3;
        float n[9], h[18], o[32];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) n[i * 3 + jj] = nrm_s[w][(y + i) * 6 + (x + jj)];
#pragma unroll
        for (int q = 0; q < 18; ++q) h[q] = acc_s[w][q][(y + 2) * 8 + (x + 2)];
        cell_features(h, n, o);
        if (SAMPLE) o[31] = grey;
        float4* dst = reinterpret_cast<float4*>(feat + (((size_t)b * NSC + k) * 16 + lane) * PVF_FHOG_STRIDE);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
}

// stage access (pvf_debug_fhog at cell 4, 23 x 23, padding 1): n images, 32 per grid row like the tracker's calls
void fhog_scale_chips(Ctx* c, const uint8_t* d_chips, int n, float* d_feat)
{
    // the kernel addresses image (b, k) as b * 32 + k: n images = rows of 32, the tail of the last row reads and writes within scratch
    const int rows = (n + NSC - 1) / NSC;
    c->s_trk0.ensure((size_t)rows * NSC * SWIN * SWIN * 3);
    c->s_trk2.ensure((size_t)rows * NSC * 16 * PVF_FHOG_STRIDE * sizeof(float));
    HIP_CHECK(hipMemcpyAsync(c->s_trk0.p, d_chips, (size_t)n * SWIN * SWIN * 3, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL((scale_fhog_k<false>), dim3(NSC / 4, rows), dim3(256), 0, c->stream, (const TrkJob*)nullptr, 0.0, c->s_trk0.as<uint8_t>(),
                       orientation_lut(c), c->s_trk2.as<float>());
    HIP_CHECK(hipMemcpyAsync(d_feat, c->s_trk2.p, (size_t)n * 16 * PVF_FHOG_STRIDE * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
}

// the 32-point transform of fft32_seq on a register-resident line: every index is a compile-time constant after unrolling, so the 32
// complex values never touch memory (the LDS-resident loop form spent its time on dependent LDS round trips: 0.58 ms per 2000 trackers).
// Same permutation, stages, butterfly order and twiddles => the same bits.
__device__ __forceinline__ void fft32_regs(double2 (&x)[NSC], const double* __restrict__ tw)
{
#pragma unroll
    for (int i = 0; i < NSC; ++i) {
        const int j = ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
        if (j > i) { const double2 t = x[i]; x[i] = x[j]; x[j] = t; }
    }
#pragma unroll
    for (int st = 1; st <= 5; ++st) {
        const int m = 1 << st, half = m >> 1, tstep = NSC / m;
#pragma unroll
        for (int k = 0; k < NSC; k += m)
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const double wr = tw[2 * j * tstep], wi = -tw[2 * j * tstep + 1];
                const double2 a = x[k + j], b = x[k + j + half];
                const double tr = wr * b.x - wi * b.y;
                const double ti = wr * b.y + wi * b.x;
                x[k + j] = make_double2(a.x + tr, a.y + ti);
                x[k + j + half] = make_double2(a.x - tr, a.y - ti);
            }
    }
}

// Fs[idx][k] = feature(idx) of scale chip k * mask_scale[k]; FFT over k; one thread per idx (idx = cell * 32 + plane; plane 31 = the grey
// value scale_fhog_k left in the record's pad slot)
__global__ void __launch_bounds__(128) scale_fft_k(const float* __restrict__ feat, const double* __restrict__ mask_scale, const double* __restrict__ tw32,
                                                   double2* __restrict__ Fs)
{
    const int b = blockIdx.y, idx = blockIdx.x * 128 + threadIdx.x;
    double2 x[NSC];
#pragma unroll
    for (int k = 0; k < NSC; ++k) {
        const float v = feat[((size_t)b * NSC + k) * SDIM + idx];
        x[k] = make_double2((double)v * mask_scale[k], 0.0);
    }
    fft32_regs(x, tw32);
    double2* out = Fs + ((size_t)b * SDIM + idx) * NSC;
#pragma unroll
    for (int k = 0; k < NSC; ++k) out[k] = x[k];
}

// The 32-point transform of fft32_seq by the lanes of ONE wave on a line in LDS: lane i < 32 moves element brev5(i) to i, then lane t < 16
// does butterfly t of each stage -- the same butterflies, operands and twiddles as the sequential form, each evaluated exactly once, so the
// values are bit-identical; only who computes them changed.  (Rounds 1-4: one thread ran the 80 butterflies and the 32 exponentials of
// the scale target one after the other while its block waited: most of scale_start_k / scale_update_k's time.)  LDS serves a wave's
// accesses in order; the wave barriers pin the compiler's order.  Call with all lanes of the wave.
__device__ __forceinline__ void fft32_wave(double2* x, const double* __restrict__ tw, bool inverse, int lane)
{
    double2 v = make_double2(0.0, 0.0);
    if (lane < NSC) v = x[(int)(__brev((unsigned)lane) >> 27)];
    __builtin_amdgcn_wave_barrier();
    if (lane < NSC) x[lane] = v;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 1; st <= 5; ++st) {
        const int m = 1 << st, half = m >> 1, tstep = NSC / m;
        if (lane < NSC / 2) {
            const int g = lane / half, j = lane - g * half, k = g * m;
            const double wr = tw[2 * j * tstep], wi = inverse ? tw[2 * j * tstep + 1] : -tw[2 * j * tstep + 1];
            const double2 a = x[k + j], b = x[k + j + half];
            const double tr = wr * b.x - wi * b.y;
            const double ti = wr * b.y + wi * b.x;
            x[k + j] = make_double2(a.x + tr, a.y + ti);
            x[k + j + half] = make_double2(a.x - tr, a.y - ti);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (inverse && lane < NSC) {
        const double sc = 1.0 / NSC;
        double2 w = x[lane];
        w.x *= sc; w.y *= sc;
        x[lane] = w;
    }
    __builtin_amdgcn_wave_barrier();
}

// the scale filter's target exp(-|i - pos|), transformed and conjugated (oracle/pvo_dsst.c scale_target), by one wave
__device__ __forceinline__ void scale_target_wave(double2* g, double pos, const double* __restrict__ tw32, int lane)
{
    if (lane < NSC) {
        const double dist = fabs((double)lane - pos);
        g[lane] = make_double2(det_exp(-dist / 1.000), 0.0);
    }
    __builtin_amdgcn_wave_barrier();
    fft32_wave(g, tw32, false, lane);
    if (lane < NSC) g[lane].y = -g[lane].y;
    __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(256) scale_start_k(const TrkJob* __restrict__ jobs, const double2* __restrict__ Fs, const double* __restrict__ tw32)
{
    __shared__ double2 Gs[NSC];
    const int b = blockIdx.x, tid = threadIdx.x;
    double* st = jobs[b].state;
    double2* As = reinterpret_cast<double2*>(st + TRK_AS);
    if (tid < 64) scale_target_wave(Gs, NSC / 2, tw32, tid);
    __syncthreads();
    for (int e = tid; e < SDIM * NSC; e += 256) {
        const int k = e & 31;
        const double2 f = Fs[(size_t)b * SDIM * NSC + e];
        As[e] = make_double2(Gs[k].x * f.x - Gs[k].y * f.y, Gs[k].x * f.y + Gs[k].y * f.x);
    }
    if (tid < NSC) {
        // a chain of 512 additions in index order (the oracle's), fed by independent loads: sixteen in flight before the first is added
        double bsum = 0;
        const double2* fp = Fs + (size_t)b * SDIM * NSC + tid;
        for (int i = 0; i < SDIM; i += 16) {
            double2 f[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = fp[(size_t)(i + j) * NSC];
#pragma unroll
            for (int j = 0; j < 16; ++j) bsum = bsum + (f[j].x * f[j].x + f[j].y * f[j].y);
        }
        st[TRK_BS + tid] = bsum;
    }
}

__global__ void __launch_bounds__(256) scale_update_k(const TrkJob* __restrict__ jobs, const double2* __restrict__ Fs, const double* __restrict__ tw32,
                                                      double ln_alpha, double* __restrict__ results, int update_model)
{
    __shared__ double2 Gs[NSC];
    const int b = blockIdx.x, tid = threadIdx.x;
    double* st = jobs[b].state;
    double2* As = reinterpret_cast<double2*>(st + TRK_AS);
    if (tid < NSC) {
        double gr = 0, gi = 0;
        const double2* fp = Fs + (size_t)b * SDIM * NSC + tid;
        const double2* ap = As + tid;
        for (int i = 0; i < SDIM; i += 8) {                 // (the sums are chains in index order; their operands are loaded eight steps ahead)
            double2 f[8], a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { f[j] = fp[(size_t)(i + j) * NSC]; a[j] = ap[(size_t)(i + j) * NSC]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gr = gr + (f[j].x * a[j].x + f[j].y * a[j].y);
                gi = gi + (f[j].y * a[j].x - f[j].x * a[j].y);
            }
        }
        const double rec = 1.0 / (st[TRK_BS + tid] + REG_SCALE);
        Gs[tid] = make_double2(gr * rec, gi * rec);
    }
    __syncthreads();
    __shared__ double s_pos;
    if (tid < 64) fft32_wave(Gs, tw32, true, tid);
    if (tid == 0) {
        int bk = 0;
        for (int k = 1; k < NSC; ++k) if (Gs[k].x > Gs[bk].x) bk = k;
        double pos = bk;
        if (bk > 0 && bk + 1 < NSC) {
            const double p1 = bk - 1, p2 = bk, p3 = bk + 1, f1 = -Gs[bk - 1].x, f2 = -Gs[bk].x, f3 = -Gs[bk + 1].x;
            const double d1 = p2 * p2 - p3 * p3, d2 = p3 * p3 - p1 * p1, d3 = p1 * p1 - p2 * p2;
            const double t1 = (d1 * f1 + d2 * f2) + d3 * f3;
            const double d4 = p2 - p3, d5 = p3 - p1, d6 = p1 - p2;
            const double t2 = 2 * ((d4 * f1 + d5 * f2) + d6 * f3);
            if (t1 != 0 && t2 != 0) {
                pos = t1 / t2;
                if (pos < p1) pos = p1;
                if (pos > p3) pos = p3;
            }
        }
        double* ps = jobs[b].pos;
        double r[4] = {ps[0], ps[1], ps[2], ps[3]};
        scale_rect_d(r, det_exp((pos - (double)NSC / 2) * ln_alpha));
        ps[0] = r[0]; ps[1] = r[1]; ps[2] = r[2]; ps[3] = r[3];
        results[(size_t)b * 8 + 1] = r[0]; results[(size_t)b * 8 + 2] = r[1]; results[(size_t)b * 8 + 3] = r[2]; results[(size_t)b * 8 + 4] = r[3];
        results[(size_t)b * 8 + 7] = pos;
        s_pos = pos;
    }
    if (!update_model) return;                  // deferred update: position and confidence only, filters untouched
    if (tid < 64) {
        __builtin_amdgcn_wave_barrier();
        scale_target_wave(Gs, s_pos, tw32, tid);        // (lane 0 of this wave wrote s_pos: LDS serves the wave's accesses in order)
    }
    __syncthreads();
    for (int e = tid; e < SDIM * NSC; e += 256) {
        const int k = e & 31;
        const double2 f = Fs[(size_t)b * SDIM * NSC + e];
        const double2 a = As[e];
        const double nr = Gs[k].x * f.x - Gs[k].y * f.y;
        const double ni = Gs[k].x * f.y + Gs[k].y * f.x;
        As[e] = make_double2(NU_SCALE * nr + (1 - NU_SCALE) * a.x, NU_SCALE * ni + (1 - NU_SCALE) * a.y);
    }
    if (tid < NSC) {
        double bq = st[TRK_BS + tid] * (1 - NU_SCALE);
        const double2* fp = Fs + (size_t)b * SDIM * NSC + tid;
        for (int i = 0; i < SDIM; i += 16) {
            double2 f[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = fp[(size_t)(i + j) * NSC];
#pragma unroll
            for (int j = 0; j < 16; ++j) bq = bq + NU_SCALE * (f[j].x * f[j].x + f[j].y * f[j].y);
        }
        st[TRK_BS + tid] = bq;
    }
}

// Clones share the filters of their source until one of them is about to write them (start_track or a full update): both passes of a
// shot start one tracker per detection from the same frame and box, the first (deferred) updates only read the filters, and nearly
// every such tracker is dropped right after -- so the usual clone costs no device work at all.
__global__ void __launch_bounds__(256) copy_state_k(const double* __restrict__ src, double* __restrict__ dst)
{
    const double2* s = reinterpret_cast<const double2*>(src);
    double2* d = reinterpret_cast<double2*>(dst);
    constexpr int N2 = (int)(TRK_DOUBLES / 2);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N2; i += gridDim.x * 256) d[i] = s[i];
}

double* tracker_state_alloc(Ctx* c)
{
    double* d = nullptr;
    if (!c->tracker_pool.empty()) { d = c->tracker_pool.back(); c->tracker_pool.pop_back(); }
    else HIP_CHECK(hipMalloc((void**)&d, TRK_DOUBLES * sizeof(double)));
    return d;
}

// make t the only owner of its filters (copy = keep their contents)
static void own_state(Ctx* c, Tracker* t, bool copy)
{
    if (!t->share) return;
    if (*t->share > 1) {
        static_assert(TRK_DOUBLES % 2 == 0, "tracker state is copied as double2");
        double* fresh = tracker_state_alloc(c);
        if (copy) {
            ProfScope ps(c, "dsst");
            hipLaunchKernelGGL(copy_state_k, dim3(64), dim3(256), 0, c->stream, t->d_state, fresh);
        }
        --*t->share;
        t->d_state = fresh;
    } else {
        delete t->share;
    }
    t->share = nullptr;
}

void dsst_clone_many(Ctx* c, const std::vector<Tracker*>& src, const std::vector<Tracker*>& dst)
{
    const int n = (int)src.size();
    for (int i = 0; i < n; ++i) {
        PVF_REQUIRE(src[i]->started && !src[i]->pending, "clone: source tracker must be started and have no deferred update");
        PVF_REQUIRE(dst[i]->d_state == nullptr && dst[i]->share == nullptr, "clone: destination already owns a state");
        if (!src[i]->share) src[i]->share = new int(1);
        ++*src[i]->share;
        dst[i]->share = src[i]->share;
        dst[i]->d_state = src[i]->d_state;
        memcpy(dst[i]->pos, src[i]->pos, sizeof src[i]->pos);
        dst[i]->started = true; dst[i]->pending = false;
    }
}

// every call works on its own copy of the position (the tracker's position lives on the host, Tracker::pos)
__global__ void init_pos_k(const TrkJob* __restrict__ jobs, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * 4) jobs[i >> 2].pos[i & 3] = jobs[i >> 2].box[i & 3];
}

// ---------------------------------------------------------------------------------------------------
static void scale_rect_h(double r[4], double s)
{
    const double cx = (r[0] + r[2]) / 2, cy = (r[1] + r[3]) / 2;
    const double w = (r[2] - r[0]) * s, h = (r[3] - r[1]) * s;
    r[0] = cx - w / 2; r[1] = cy - h / 2; r[2] = cx + w / 2; r[3] = cy + h / 2;
}

struct DsstBuffers { uint8_t* chips64; uint8_t* feat; double2* F; double2* G0; double2* G1; double2* Fs; double* results; double* pos; TrkJob* jobs; };

static DsstBuffers prepare(Ctx* c, const std::vector<Tracker*>& t, const std::vector<Frame>& f, const double* boxes,
                           std::vector<TrkJob>& jobs, std::vector<ChipJob>& cj, bool need_F)
{
    const int n = (int)t.size();
    PVF_REQUIRE(c->ttab.set, "tracker tables not set (pvf_set_tracker_tables)");
    jobs.resize(n); cj.resize(n);
    for (int i = 0; i < n; ++i) {
        const double* p = boxes ? boxes + 4 * i : t[i]->pos;
        double r[4] = {p[0], p[1], p[2], p[3]};
        scale_rect_h(r, 1.4);
        ChipDetails d{r[0], r[1], r[2], r[3], 1.0, 0.0, FS, FS};
        cj[i] = chip_plan(f[i], d);
        TrkJob& j = jobs[i];
        j.state = t[i]->d_state; j.img = f[i].d; j.h = f[i].h; j.w = f[i].w;
        j.map[0] = r[0]; j.map[1] = r[1];
        j.map[2] = (r[2] - r[0]) / (double)(FS - 1);
        j.map[3] = (r[3] - r[1]) / (double)(FS - 1);
        j.cx = ((p[0] + p[2]) / 2 - j.map[0]) / j.map[2];
        j.cy = ((p[1] + p[3]) / 2 - j.map[1]) / j.map[3];
        for (int k = 0; k < 4; ++k) j.box[k] = p[k];
    }
    DsstBuffers b;
    const size_t chips64 = (size_t)n * FS * FS * 3;
    c->s_trk0.ensure(chips64 + 256);
    b.chips64 = c->s_trk0.as<uint8_t>();
    // a buffer of the tracker's own (round 4).  Up to round 3 this was the detector's s_feat: the chips' features overwrote the ZERO BORDER
    // of the detector's level-0 feature maps without telling it (feat_ring_owner), so the next detector batch of the same plan scored
    // the windows that reach into the border -- the top rows of the first frames -- on tracker features.  Invisible at the shipped
    // threshold on faces away from the frame's edge; found by tests/test_gpu_parity.py::test_detector_with_hundreds_of_candidates_at_
    // the_threshold after a tracker test.  84 KB per tracker of a call: the compact translation record
    // (TRKF_BYTES), then -- the translation pass done -- the 32 x 16 x 32 floats of the scale samples (64 KB) in the same place.
    static_assert(TRKF_BYTES >= (size_t)NSC * 16 * PVF_FHOG_STRIDE * sizeof(float), "the scale features reuse the translation record's room");
    c->s_trkfeat.ensure((size_t)n * TRKF_BYTES);
    b.feat = c->s_trkfeat.as<uint8_t>();
    b.F = nullptr;
    if (need_F) {                                  // plane spectra in HBM: the full update only (2 MB per tracker)
        c->s_trk1.ensure((size_t)n * NPL * FS * FS * sizeof(double2));
        b.F = c->s_trk1.as<double2>();
    }
    const size_t g = need_F ? (size_t)n * FS * FS * sizeof(double2) : 0, fs = (size_t)n * SDIM * NSC * sizeof(double2);
    c->s_trk2.ensure(2 * g + fs + (size_t)n * 12 * sizeof(double) + (size_t)n * sizeof(TrkJob) + 256);
    uint8_t* q = c->s_trk2.as<uint8_t>();
    b.G0 = reinterpret_cast<double2*>(q); q += g;
    b.G1 = reinterpret_cast<double2*>(q); q += g;
    b.Fs = reinterpret_cast<double2*>(q); q += fs;
    b.results = reinterpret_cast<double*>(q); q += (size_t)n * 8 * sizeof(double);
    b.pos = reinterpret_cast<double*>(q); q += (size_t)n * 4 * sizeof(double);
    b.jobs = reinterpret_cast<TrkJob*>(q);
    for (int i = 0; i < n; ++i) jobs[i].pos = b.pos + 4 * (size_t)i;
    void* hj = c->stage.take((size_t)n * sizeof(TrkJob));
    memcpy(hj, jobs.data(), (size_t)n * sizeof(TrkJob));
    HIP_CHECK(hipMemcpyAsync(b.jobs, hj, (size_t)n * sizeof(TrkJob), hipMemcpyHostToDevice, c->stream));
    c->stage.sent(c->stream);
    hipLaunchKernelGGL(init_pos_k, dim3((4 * n + 255) / 256), dim3(256), 0, c->stream, b.jobs, n);
    return b;
}

static const size_t LDS_FFT = (size_t)FS * LP * sizeof(double2);

static void translation_features(Ctx* c, const DsstBuffers& b, const std::vector<ChipJob>& cj, int n)
{
    chip_extract_batch(c, cj, b.chips64);
    ProfScope ps(c, "dsst");
    fhog1_compact(c, b.chips64, n, b.feat);
    if (b.F) hipLaunchKernelGGL(trans_planes_fft_k, dim3(NPL, n), dim3(256), LDS_FFT, c->stream, b.chips64, b.feat, c->ttab.d_mask64, c->ttab.d_tw64, b.F);
}

static void scale_features(Ctx* c, const DsstBuffers& b, int n)
{
    ProfScope ps(c, "dsst");
    float* feat = reinterpret_cast<float*>(b.feat);     // reuse: n x 32 samples of 4 x 4 x 32 floats (the translation pass is done with it)
    hipLaunchKernelGGL((scale_fhog_k<true>), dim3(NSC / 4, n), dim3(256), 0, c->stream, b.jobs, c->ttab.alpha_pow_m16, (const uint8_t*)nullptr,
                       orientation_lut(c), feat);
    hipLaunchKernelGGL(scale_fft_k, dim3(SDIM / 128, n), dim3(128), 0, c->stream, feat, c->ttab.d_mask_scale, c->ttab.d_tw32, b.Fs);
}

static void ensure_fft_lds(int device)
{
    static std::atomic<uint64_t> done_on{0};               // per device (a function attribute belongs to the device it was set on)
    const uint64_t bit = 1ull << (device & 63);
    if (done_on.load() & bit) return;
    HIP_CHECK(hipFuncSetAttribute((const void*)trans_planes_fft_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FFT));
    HIP_CHECK(hipFuncSetAttribute((const void*)peak_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FFT));
    HIP_CHECK(hipFuncSetAttribute((const void*)start_fused_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FFT));
    HIP_CHECK(hipFuncSetAttribute((const void*)update_fused_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FFT));
    done_on.fetch_or(bit);
}

void dsst_start_many(Ctx* c, const std::vector<Tracker*>& t, const std::vector<Frame>& f, const double* boxes)
{
    const int n = (int)t.size();
    if (n == 0) return;
    ensure_fft_lds(c->device);
    for (int i = 0; i < n; ++i) {
        own_state(c, t[i], false);
        if (!t[i]->d_state) t[i]->d_state = tracker_state_alloc(c);
        memcpy(t[i]->pos, boxes + 4 * i, 4 * sizeof(double));
        t[i]->started = true;
        t[i]->pending = false;
    }
    std::vector<TrkJob> jobs; std::vector<ChipJob> cj;
    DsstBuffers b = prepare(c, t, f, boxes, jobs, cj, false);
    translation_features(c, b, cj, n);
    {
        ProfScope ps(c, "dsst");
        hipLaunchKernelGGL(start_fused_k, dim3(n), dim3(FUSED_NT), LDS_FFT, c->stream, b.jobs, b.chips64, b.feat, c->ttab.d_mask64, c->ttab.d_tw64);
    }
    scale_features(c, b, n);
    {
        ProfScope ps(c, "dsst");
        hipLaunchKernelGGL(scale_start_k, dim3(n), dim3(256), 0, c->stream, b.jobs, b.Fs, c->ttab.d_tw32);
    }
    HIP_CHECK(hipGetLastError());
    // start_track returns nothing: the call ends with its kernels queued, and the host work of the next call (the first updates of
    // these trackers, usually) runs beside them
}

// mode 0: dlib's update().  mode 1 (deferred): the same response, peak, position and confidence, but the translation and
// scale filters are left as they were and the tracker remembers where it started from -- most trackers of the batched host path
// are killed right after their first update (their face was detected again), so the model update would be thrown away.
// mode 2 (commit): for a tracker that survived a deferred update, redo that update in full on the same frame; the state is
// then bit-identical to what mode 0 would have left (same inputs, same arithmetic).
void dsst_update_many(Ctx* c, const std::vector<Tracker*>& t, const std::vector<Frame>& f, double* psr, double* boxes_out, int mode)
{
    const int n = (int)t.size();
    if (n == 0) return;
    ensure_fft_lds(c->device);
    for (int i = 0; i < n; ++i) {
        PVF_REQUIRE(t[i]->started, "tracker.update before start_track");
        if (mode == 2) {
            PVF_REQUIRE(t[i]->pending, "commit of a tracker that has no deferred update");
            memcpy(t[i]->pos, t[i]->prev_pos, sizeof t[i]->pos);
            t[i]->pending = false;
        } else {
            PVF_REQUIRE(!t[i]->pending, "tracker has a deferred update: commit it (with the frame it was computed on) first");
        }
        if (mode == 1) memcpy(t[i]->prev_pos, t[i]->pos, sizeof t[i]->pos);
        else own_state(c, t[i], true);             // the filters are about to be written
    }
    std::vector<TrkJob> jobs; std::vector<ChipJob> cj;
    DsstBuffers b = prepare(c, t, f, nullptr, jobs, cj, mode != 1);
    translation_features(c, b, cj, n);
    if (mode == 1) {
        ProfScope ps(c, "dsst");
        hipLaunchKernelGGL(update_fused_k, dim3(n), dim3(FUSED_NT), LDS_FFT, c->stream, b.jobs, b.chips64, b.feat, c->ttab.d_mask64, c->ttab.d_tw64, b.results);
    } else {
        // the full update reads every plane spectrum twice (response, then filter update with the NEW target): spectra go through HBM
        ProfScope ps(c, "dsst");
        hipLaunchKernelGGL(corr_k, dim3(FS * FS / 256, n), dim3(256), 0, c->stream, b.jobs, b.F, b.G0);
        hipLaunchKernelGGL(peak_k, dim3(n), dim3(256), LDS_FFT, c->stream, b.jobs, b.G0, c->ttab.d_tw64, b.results, b.G1);
        hipLaunchKernelGGL(filter_update_k, dim3(FS * FS / 256, n), dim3(256), 0, c->stream, b.jobs, b.F, b.G1);
    }
    scale_features(c, b, n);
    {
        ProfScope ps(c, "dsst");
        hipLaunchKernelGGL(scale_update_k, dim3(n), dim3(256), 0, c->stream, b.jobs, b.Fs, c->ttab.d_tw32, c->ttab.ln_alpha, b.results,
                           mode != 1 ? 1 : 0);
    }
    HIP_CHECK(hipGetLastError());
    c->h_misc.ensure((size_t)n * 8 * sizeof(double));
    double* hr = c->h_misc.as<double>();
    HIP_CHECK(hipMemcpyAsync(hr, b.results, (size_t)n * 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) {
        if (psr) psr[i] = hr[8 * i];
        for (int k = 0; k < 4; ++k) t[i]->pos[k] = hr[8 * i + 1 + k];
        if (boxes_out) memcpy(boxes_out + 4 * i, t[i]->pos, 4 * sizeof(double));
        if (mode == 1) t[i]->pending = true;
    }
}

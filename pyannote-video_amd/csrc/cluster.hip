// cluster.hip -- K10/K11: track-pair mean Euclidean distances and average-linkage agglomeration
// (reference pyannote/video/face/clustering.py:100-112 pdist + block means, :116-119 merged similarity, :138-141 stop at 0.6).
// float64 like scipy/numpy in the reference; neither the N x N matrix nor an N x T intermediate is materialised.
//   128-D rows (the embeddings): 16 x 16 tiles of x_a . x_b on the f64 matrix cores (v_mfma_f64_16x16x4_f64), distance and the
//   per-track-pair reduction inside the tile pass (pair_tiles_k below);
//   other dimensions: the plain per-row kernels (row_track_sums_k / track_pair_mean_k) with an N x T scratch.
// HAC keeps D in HBM with cached row minima; a merge costs O(T) plus re-scans of the rows whose minimum died.  Up to 10 240 tracks the
// whole agglomeration is ONE persistent workgroup (row minima in LDS, no launches inside the loop).
#include "pvf_internal.h"
#include <cmath>

__global__ void __launch_bounds__(256) transpose_k(const double* __restrict__ X, int N, int dim, double* __restrict__ Xt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * dim) return;
    const int a = (int)(i / dim), k = (int)(i % dim);
    Xt[(size_t)k * N + a] = X[i];
}

// one block per row a: lanes sweep all rows b (dimension-major X => coalesced), distances of a 2048-row chunk sit in LDS,
// then one lane per track continues that track's running sum over the chunk in row order (a single sequential chain).
#define PD_CHUNK 2048
__global__ void __launch_bounds__(256) row_track_sums_k(const double* __restrict__ X, const double* __restrict__ Xt, int N, int dim,
                                                        const int32_t* __restrict__ row_start, int T, double* __restrict__ S, int a0)
{
    extern __shared__ __attribute__((aligned(16))) double sm[]; // xa[dim] + dist[PD_CHUNK]
    double* xa = sm;
    double* dist = sm + dim;
    const int a = a0 + blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < dim; k += 256) xa[k] = X[(size_t)a * dim + k];
    for (int j = tid; j < T; j += 256) S[(size_t)a * T + j] = 0.0;
    __syncthreads();
    for (int c0 = 0; c0 < N; c0 += PD_CHUNK) {
        const int c1 = min(c0 + PD_CHUNK, N);
        for (int b = c0 + tid; b < c1; b += 256) {
            double s = 0;
            for (int k = 0; k < dim; ++k) { const double d = xa[k] - Xt[(size_t)k * N + b]; s += d * d; }
            dist[b - c0] = sqrt(s);
        }
        __syncthreads();
        for (int j = tid; j < T; j += 256) {
            const int b0 = max(row_start[j], c0), b1 = min(row_start[j + 1], c1);
            if (b0 < b1) {
                double run = S[(size_t)a * T + j];
                for (int b = b0; b < b1; ++b) run += dist[b - c0];
                S[(size_t)a * T + j] = run;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) track_pair_mean_k(const double* __restrict__ S, const int32_t* __restrict__ row_start, int T,
                                                         double* __restrict__ D, int t0)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t0 + blockIdx.y;
    if (j >= T) return;
    if (j <= i) { D[(size_t)i * T + j] = 0.0; return; }      // upper triangle only (mirror_upper_k fills the rest)
    double sum = 0;
    for (int a = row_start[i]; a < row_start[i + 1]; ++a) sum += S[(size_t)a * T + j];
    const double cnt = (double)(row_start[i + 1] - row_start[i]) * (double)(row_start[j + 1] - row_start[j]);
    D[(size_t)i * T + j] = sum / cnt;
}

// ---------------------------------------------------------------------------------------------------
// K10 on the f64 matrix cores.
//
// Rows (sorted by track) are cut into ROW BLOCKS of 16 consecutive rows wherever the tracks begin and end: a block holds up to 16
// SEGMENTS (runs of rows of one track), a track longer than what is left of its first block continues in the next ones.  The blocking
// depends on the track sizes only, never on which rank computes which rows, so every D entry is formed by the same additions wherever
// it is computed.  (Round 2 / early round 3 started every track >= 16 rows on a block of its own and packed shorter tracks whole:
// 16-row blocks 39 % full at 10 rows per track.)
// A wave owns one row block (its 16 x 128 values stay in registers as 32 A fragments) and sweeps a range of column blocks, which
// its workgroup stages through LDS (one copy serves 4 row blocks).  Per 16 x 16 tile:
//   32 MFMAs give x_a . x_b;  d = sqrt(|a|^2 + |b|^2 - 2 a.b)  (recomputed as sum (a_k - b_k)^2 where the Gram form would cancel;
//   cosine: 1 - a.b / (|a| |b|)) -- the tile of distances sits in the C layout, 4 values per lane;
//   the reduction to track pairs runs on the matrix cores as well: with the 0/1 matrices Rind[row segment][row] (which rows of this
//   row block form which segment) and Cind[column][column segment],
//       R' = d^T . Rind^T   (4 MFMAs: the C layout of d IS the A layout of d^T)      sums over the rows of every row segment, per column
//       T2 += R'^T . Cind   (4 MFMAs: the C layout of R' IS the A layout of R'^T)    ... and over the columns of every column segment
//   (products with 1.0 and sums with 0.0 are exact; the order of the additions is the matrix core's, the same for every launch
//   shape).  A column segment whose track goes on in the next block hands its column of T2 on to segment 0 of the next tile (a lane
//   shuffle); complete ones are written: row segments that are a whole track -> D[i][j] = sum / (n_i n_j); row segments of a track
//   that spans several blocks -> P[part][j], summed in part order by pair_chunks_k.
// ONLY THE UPPER TRIANGLE IS COMPUTED (round 4), like the reference: clustering.py:104-112 walks itertools.combinations(range(n), 2) and
// stores matrix[i, j] = matrix[j, i] = similarity.  Rows are sorted by track, so every (row, column) pair of a track pair i < j lies in
// a tile whose column block is not before its row block: a wave visits the column blocks cb >= its row block only, entries with
// track(row) >= track(column) are never written, and mirror_upper_k copies D[i][j] to D[j][i] afterwards (round 3 swept every column
// block for every row block: twice the algorithm's FLOPs).  The additions that form an entry i < j are the same as before, in the same order.
// D[i][i] = 0 like scipy's squareform diagonal.  Round 2 reduced the tile with LDS round trips and lane-serial running sums (16
// dependent steps per tile on 16 lanes): 10-16 TFLOP/s; the matrix-core reduction removes every serial step from the tile.
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define PT_PITCH 130                      // doubles per staged row: (2 j + k) mod 32 distinct bank pairs for the B fragment reads

struct PtArgs {
    const double* X; const double* nrm; int N, T;
    const int* row_track; const int* row_segidx;                         // per row: track, index of its segment inside its block
    const int* blk_cont;                                                   // per block: the segment that goes on in the next block, or -1
    const int* seg_track; const int* seg_part;                             // [block][16]: track of each segment (-1: none); its row in P (-1: the segment is its whole track)
    const int* range_b0;                                                   // column ranges: block index bounds [n_ranges + 1]
    const int* row_start; double* D; double* P;
    int n_blocks, n_ranges, t0, t1, metric;
};

__global__ void __launch_bounds__(256) row_norms_k(const double* __restrict__ X, int N, int dim, double* __restrict__ nrm)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N) return;
    double s = 0;
    for (int k = 0; k < dim; ++k) { const double v = X[(size_t)a * dim + k]; s += v * v; }
    nrm[a] = s;
}

// sqrt of a squared distance: 0 or a number far from the ends of the exponent range.  The steps are the device library's own (v_rsq_f64,
// one coupled Newton step on g ~ sqrt x and h ~ 1 / (2 sqrt x), two corrections by the exact residual x - g g), i.e. the same bits, without
// the rescaling and class checks it wraps around them for subnormal / huge / infinite arguments (7 of its 17 instructions).
__device__ __forceinline__ double sqrt_nonneg(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return x == 0.0 ? 0.0 : g;                       // (rsq(0) is infinite)
}

__device__ __forceinline__ void pair_tiles_body(const PtArgs& a)
{
    constexpr int DIM = 128, KS = DIM / 4;
    __shared__ __attribute__((aligned(16))) double Bs[2][16 * PT_PITCH];
    __shared__ int colSeg[2][16];                     // per staged column: segment index inside its block (-1: padding)
    __shared__ int colSegTrack[2][16];                // per segment of the staged block: its track (-1: none)
    __shared__ double colNrm[2][16];                  // |b|^2 per staged column
    __shared__ int colCont[2];                        // the staged block's segment that goes on in the next block (-1: none)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ab = blockIdx.x * 4 + wave;             // row block of this wave
    const bool have = ab < a.n_blocks;
    // upper triangle: this workgroup's four row blocks need the column blocks from its first row block on; a range that ends before it
    // has nothing for them.  (Starting inside a range leaves the sums of a column track that began earlier incomplete: that track is
    // the one the first row of the row block belongs to, i.e. not AFTER any row track of this workgroup -- never written.)
    const int cb1 = a.range_b0[blockIdx.y + 1];
    const int cb0 = max(a.range_b0[blockIdx.y], (int)blockIdx.x * 4);
    if (cb0 >= cb1) return;
    const int ar0 = have ? ab * 16 : 0, anr = have ? min(16, a.N - ab * 16) : 0;
    // a wave whose rows all lie outside [t0, t1) has nothing to write (row tracks are contiguous in a block)
    bool wanted = false;
    if (have) { const int ta = a.row_track[ar0], tb = a.row_track[ar0 + anr - 1]; wanted = (tb >= a.t0 && ta < a.t1); }
    const int i16 = lane & 15, k4 = lane >> 4;
    double af[KS];
    {
        const bool ok = have && i16 < anr;
        const double* xa = a.X + (size_t)(ar0 + (ok ? i16 : 0)) * DIM + k4;
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = ok ? xa[4 * s] : 0.0;
    }
    double na4[4], rb[4];
    int rt[4], rpart[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = k4 + 4 * r;                    // C-layout row of register r == K index of step r in the row reduction
        const bool ok = have && row < anr;
        na4[r] = ok ? a.nrm[ar0 + row] : 0.0;
        rb[r] = (ok && a.row_segidx[ar0 + row] == i16) ? 1.0 : 0.0;          // Rind^T[row][segment i16]
        // T2's C layout: register r of this lane holds row segment k4 + 4 r, column segment i16
        const int seg = k4 + 4 * r;
        int t = have ? a.seg_track[ab * 16 + seg] : -1;
        if (t >= 0 && !(t >= a.t0 && t < a.t1)) t = -1;                       // rows of another rank's share
        rt[r] = t;
        rpart[r] = t >= 0 ? a.seg_part[ab * 16 + seg] : -1;
    }
    // staging: a wave moves rows wave, wave + 4, ... of the column block straight from HBM into LDS (buffer_load_dwordx4 ... lds: 64 lanes x
    // 16 bytes = one row of 128 doubles, no registers in between), asynchronously: the block for step cb + 1 is requested before the
    // MFMAs of step cb and only waited for at the end of the step.  (Round 2 / early round 3 loaded into registers and parked them in LDS
    // BEFORE computing: every step began with a full memory latency.)  Rows past the block's last one are requested beyond the buffer's
    // end, which returns zeros.
    // (the descriptor covers this workgroup's column range only: 32-bit offsets, whatever N is)
    const int range_rows = min(a.N - cb0 * 16, (cb1 - cb0) * 16);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (size_t)cb0 * 16 * DIM), 0, range_rows * DIM * 8, 0x00020000);
    auto stage = [&](int cb, int buf) {
        const int r0 = cb * 16, nr = min(16, a.N - r0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave + 4 * q;
            const int voff = (r < nr) ? lane * 16 : 0x7ffffff0;                  // (wave-uniform choice; out of range -> zeros)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)&Bs[buf][r * PT_PITCH], 16, voff,
                                                     (r0 - cb0 * 16 + (r < nr ? r : 0)) * (DIM * 8), 0, 0);
        }
        if (tid < 16) {
            colSeg[buf][tid] = tid < nr ? a.row_segidx[r0 + tid] : -1;
            colSegTrack[buf][tid] = a.seg_track[cb * 16 + tid];
            colNrm[buf][tid] = tid < nr ? a.nrm[r0 + tid] : 0.0;
            if (tid == 0) colCont[buf] = a.blk_cont[cb];
        }
    };
    f64x4 t2 = (f64x4){0.0, 0.0, 0.0, 0.0};          // sums per (row segment, column segment), carried across the chunks of a long column track
    if (cb0 < cb1) stage(cb0, 0);
    __builtin_amdgcn_s_waitcnt(0);                     // the rows requested above are in LDS
    __syncthreads();
    for (int cb = cb0; cb < cb1; ++cb) {
        const int buf = (cb - cb0) & 1;
        if (cb + 1 < cb1) stage(cb + 1, buf ^ 1);
        if (wanted && cb >= ab) {
            const int br0 = cb * 16;
            f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0};
            const double* bp = &Bs[buf][i16 * PT_PITCH + k4];
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bp[4 * s], acc, 0, 0, 0);
            // C/D of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * reg.  Rows past the block's last one and padding columns need
            // no masking here: their distances are finite (their fragments are zeros) and the 0/1 matrices of the two reductions leave them out.
            const double nb = colNrm[buf][i16];
            double d[4];
            if (a.metric == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double den = sqrt(na4[r] * nb);
                    d[r] = den > 0.0 ? 1.0 - acc[r] / den : 0.0;
                }
            } else {
                double d2[4];
                bool fix = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double sum = na4[r] + nb;
                    d2[r] = sum - 2.0 * acc[r];
                    // the Gram form cancels for close rows: its absolute error in d is ~1e-16 (|a|^2 + |b|^2) / d, i.e. below 1e-13 down
                    // to d ~ 3e-3 |x|; closer pairs (identical or near-identical rows) take the differences instead
                    fix = fix || (d2[r] < 1e-5 * sum && k4 + 4 * r < anr && colSeg[buf][i16] >= 0);
                }
                if (__builtin_amdgcn_ballot_w64(fix) != 0) {           // rare: some lane of the wave holds such a pair
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = k4 + 4 * r;
                        if (d2[r] < 1e-5 * (na4[r] + nb) && row < anr && colSeg[buf][i16] >= 0) {
                            const double* xa = a.X + (size_t)(ar0 + row) * DIM;
                            const double* xb = a.X + (size_t)(br0 + i16) * DIM;
                            double e = 0.0;
                            for (int k = 0; k < DIM; ++k) { const double t = xa[k] - xb[k]; e += t * t; }
                            d2[r] = e;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = sqrt_nonneg(d2[r] > 0.0 ? d2[r] : 0.0);
            }
            // rows of every row track: R'[column][row segment] = sum_row d[row][column] Rind[segment][row]; this lane's d[s] is
            // element (column i16, row 4 s + k4) of d^T, i.e. the A operand of step s
            f64x4 rp = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) rp = __builtin_amdgcn_mfma_f64_16x16x4f64(d[s], rb[s], rp, 0, 0, 0);
            // columns of every column track: T2[row segment][column segment] += sum_col R'[col][row segment] Cind[col][column segment];
            // rp[q] is element (row segment i16, column 4 q + k4) of R'^T: the A operand of step q
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double cind = (colSeg[buf][4 * q + k4] == i16) ? 1.0 : 0.0;
                t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(rp[q], cind, t2, 0, 0, 0);
            }
            // complete column segments are written; the one that goes on in the next block (at most one, the block's last) hands its sums to
            // segment 0 of the next tile
            const int cont = colCont[buf];                 // (uniform)
            const int j = colSegTrack[buf][i16];
            if (j >= 0 && i16 != cont) {
                // sums, not means: the division by the two track lengths (35 instructions per entry, in a branch few lanes take) is left
                // to pair_norm_k / pair_chunks_k, one pass over the finished rows
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = rt[q];
                    if (i < 0 || i >= j) continue;             // track pairs i < j only
                    if (rpart[q] >= 0) a.P[(size_t)rpart[q] * a.T + j] = t2[q];
                    else a.D[(size_t)i * a.T + j] = t2[q];
                }
            }
            f64x4 carry = (f64x4){0.0, 0.0, 0.0, 0.0};
            if (cont >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double v = __shfl(t2[q], (lane & 48) | cont, 64);
                    carry[q] = (i16 == 0) ? v : 0.0;
                }
            }
            t2 = carry;
        }
        __builtin_amdgcn_s_waitcnt(0);                 // this wave's share of the next block has arrived ...
        __syncthreads();                               // ... and so has everybody else's
    }
}

// 168 registers = three waves per SIMD (the compiler's own choice, 232 registers and two waves, was measured 11 % slower at N = 180 000:
// 33.0 vs 37.2 TFLOP/s; a wave's epilogue -- four IEEE square roots per lane -- needs other waves' MFMAs to hide behind)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) pair_tiles_k(PtArgs a) { pair_tiles_body(a); }

// long tracks: D[i][j] = (P[c0][j] + P[c0+1][j] + ...) / (n_i n_j), chunks in order
__global__ void __launch_bounds__(256) pair_chunks_k(const double* __restrict__ P, const int* __restrict__ big_track, const int* __restrict__ big_c0,
                                                     const int* __restrict__ big_nc, const int32_t* __restrict__ row_start, int T, double* __restrict__ D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = big_track[blockIdx.y];
    if (j >= T) return;
    if (j <= i) { D[(size_t)i * T + j] = 0.0; return; }      // (P holds nothing for j <= i)
    double s = 0.0;
    for (int c = 0; c < big_nc[blockIdx.y]; ++c) s += P[(size_t)(big_c0[blockIdx.y] + c) * T + j];
    const double cnt = (double)(row_start[i + 1] - row_start[i]) * (double)(row_start[j + 1] - row_start[j]);
    D[(size_t)i * T + j] = s / cnt;
}

// rows of D that pair_tiles_k wrote as sums (tracks that lie inside one 16-row block): sum / (rows of i x rows of j), the diagonal zero
__global__ void __launch_bounds__(256) pair_norm_k(double* __restrict__ D, const int32_t* __restrict__ row_start, const int* __restrict__ is_big, int T, int t0)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t0 + blockIdx.y;
    if (j >= T || is_big[i]) return;
    const int ni = row_start[i + 1] - row_start[i], nj = row_start[j + 1] - row_start[j];
    if (j <= i || ni <= 0 || nj <= 0) { D[(size_t)i * T + j] = 0.0; return; }
    const double v = D[(size_t)i * T + j];
    D[(size_t)i * T + j] = v / ((double)ni * (double)nj);
}

// D[j][i] = D[i][j] for i < j (clustering.py:111-112): 32 x 32 tiles turned through LDS so that reads and writes are both row-wise
__global__ void __launch_bounds__(256) mirror_upper_k(double* __restrict__ D, int T)
{
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int i = bi * 32 + r, j = bj * 32 + tx;
        tile[r][tx] = (i < T && j < T) ? D[(size_t)i * T + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int j = bj * 32 + r, i = bi * 32 + tx;          // writes D[j][i], i runs along the row
        if (i < T && j < T && i < j) D[(size_t)j * T + i] = tile[tx][r];
    }
}

void mirror_upper_dev(Ctx* c, double* dD, int T)
{
    const int nb = (T + 31) / 32;
    hipLaunchKernelGGL(mirror_upper_k, dim3(nb, nb), dim3(256), 0, c->stream, dD, T);
}

// the table the clustering works on, made on the device: row k = np.round(float64(emb[order[k]]), decimals) (pipeline's X: the float64
// value of the 5-decimal text the reference writes and reads back, pyannote-face.py:307-311 + clustering.py:70-75; numpy evaluates the
// rounding as rint(x * 10^d) / 10^d -- three correctly rounded operations, the same three here and in pvf_round_rows)
__global__ void __launch_bounds__(256) gather_round_rows_k(const uint8_t* __restrict__ emb, long long stride_bytes, const int32_t* __restrict__ order,
                                                           int N, int dim, double scale, int do_round, double* __restrict__ X)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * dim) return;
    const int k = (int)(i / dim), d = (int)(i % dim);
    const int src = order ? order[k] : k;
    double v = (double)reinterpret_cast<const float*>(emb + (size_t)src * stride_bytes)[d];
    if (do_round) v = rint(v * scale) / scale;
    X[i] = v;
}

// the input table on the device: dX[N][dim] float64, from the float64 host table or from float32 rows (host or device) gathered + rounded
static void stage_table(Ctx* c, const PairInput& in, int N, int dim, double* dX, uint8_t* d_tmp, int32_t* d_order)
{
    if (in.X) { HIP_CHECK(hipMemcpyAsync(dX, in.X, (size_t)N * dim * 8, hipMemcpyHostToDevice, c->stream)); return; }
    const uint8_t* src = reinterpret_cast<const uint8_t*>(in.emb);
    long long stride = in.emb_stride;
    if (!in.emb_on_device) {
        // host rows: one strided copy into scratch (4 bytes per value: half of what the float64 table would move)
        HIP_CHECK(hipMemcpy2DAsync(d_tmp, (size_t)dim * 4, in.emb, (size_t)in.emb_stride, (size_t)dim * 4, (size_t)in.n_src, hipMemcpyHostToDevice, c->stream));
        src = d_tmp; stride = (long long)dim * 4;
    }
    if (in.order) HIP_CHECK(hipMemcpyAsync(d_order, in.order, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    double scale = 1.0;
    for (int d = 0; d < in.decimals; ++d) scale *= 10.0;
    hipLaunchKernelGGL(gather_round_rows_k, dim3((unsigned)(((size_t)N * dim + 255) / 256)), dim3(256), 0, c->stream, src, stride,
                       in.order ? d_order : nullptr, N, dim, scale, in.decimals >= 0 ? 1 : 0, dX);
}

static inline size_t stage_tmp_bytes(const PairInput& in, int dim) { return (in.X || in.emb_on_device) ? 0 : (size_t)in.n_src * dim * 4; }

// rows [t0, t1) of D leave the device: into the full matrix `out` addresses (rows outside the range untouched) or, compact, into
// (t1 - t0) x T doubles; host or device memory
static void copy_rows_out(Ctx* c, const double* dD, int T, int t0, int t1, const PairOutput& o)
{
    if (!o.out || t1 <= t0) return;
    double* dst = o.compact ? o.out : o.out + (size_t)t0 * T;
    HIP_CHECK(hipMemcpyAsync(dst, dD + (size_t)t0 * T, (size_t)(t1 - t0) * T * sizeof(double), o.on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
}

static void pair_mean_dist_mfma(Ctx* c, const PairInput& in, int N, const int32_t* row_start, int T, const PairOutput& out, double** d_D_keep, int t0, int t1,
                                int metric, bool mirror)
{
    constexpr int DIM = 128;
    // ---- blocking (host, O(N)): blocks of 16 consecutive rows; segments = runs of one track inside a block
    const int nb = (N + 15) / 16;
    std::vector<int> row_track(N), row_segidx(N, 0), blk_cont(nb, -1), blk_clean(nb, 1), seg_track((size_t)nb * 16, -1), seg_part((size_t)nb * 16, -1);
    std::vector<int> big_track, big_c0, big_nc, is_big(T, 0);
    for (int t = 0; t < T; ++t) for (int r = row_start[t]; r < row_start[t + 1]; ++r) row_track[r] = t;
    for (int b = 0; b < nb; ++b) {
        const int r0 = b * 16, nr = std::min(16, N - r0);
        int seg = 0;
        for (int r = r0; r < r0 + nr; ++r) {
            if (r > r0 && row_track[r] != row_track[r - 1]) ++seg;
            row_segidx[r] = seg;
            seg_track[(size_t)b * 16 + seg] = row_track[r];
        }
        if (r0 + nr < N && row_track[r0 + nr] == row_track[r0 + nr - 1]) blk_cont[b] = seg;      // the last segment's track goes on
        if (b > 0 && row_track[r0] == row_track[r0 - 1]) blk_clean[b] = 0;                         // ... and this block takes it over
    }
    // tracks that span several blocks: one row of P per (track, block), summed in block order afterwards
    int n_chunks = 0;
    for (int t = 0; t < T; ++t) {
        if (row_start[t + 1] <= row_start[t]) continue;
        const int b_first = row_start[t] / 16, b_last = (row_start[t + 1] - 1) / 16;
        if (b_last == b_first) continue;
        big_track.push_back(t); big_c0.push_back(n_chunks); big_nc.push_back(b_last - b_first + 1);
        is_big[t] = 1;
        for (int b = b_first; b <= b_last; ++b)
            seg_part[(size_t)b * 16 + (b == b_first ? row_segidx[row_start[t]] : 0)] = n_chunks++;
    }
    // column ranges, cut only at blocks that do not take a segment over from their predecessor (the sums of a track that spans blocks
    // are carried from tile to tile inside a range).  With the upper triangle about half of the (row group, range) workgroups have work
    // (the ones on the diagonal part of a range); ~24 pieces per workgroup slot (3 resident per CU) keep the tail of the launch -- the
    // last pieces running alone -- near 2 % of it; a piece is at least 16 column blocks (its row blocks are loaded once per piece)
    const int row_groups = (nb + 3) / 4;
    int want_ranges = (2 * 24 * 3 * c->n_cu + row_groups - 1) / row_groups;
    want_ranges = std::max(1, std::min(want_ranges, std::max(1, nb / 16)));
    want_ranges = std::max(want_ranges, (nb >> 16) + 1);            // a range is addressed with 32-bit byte offsets: at most 2^20 rows
    std::vector<int> range_b0{0};
    for (int k = 1; k < want_ranges; ++k) {
        int b = (int)((long long)nb * k / want_ranges);
        while (b < nb && !blk_clean[b]) ++b;
        if (b > range_b0.back() && b < nb) range_b0.push_back(b);
    }
    range_b0.push_back(nb);
    const int n_ranges = (int)range_b0.size() - 1;
    for (int k = 0; k < n_ranges; ++k) PVF_REQUIRE(range_b0[k + 1] - range_b0[k] <= (1 << 17), "pair_mean_dist: a track of more than a million rows");
    // ---- device buffers
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t xb = al((size_t)N * DIM * 8), nbz = al((size_t)N * 8), ib = al((size_t)N * 4), bb = al((size_t)nb * 4), pb = al((size_t)std::max(n_chunks, 1) * T * 8);
    const size_t rsb = al((size_t)(T + 1) * 4), rgb = al((size_t)(n_ranges + 1) * 4), bigb = al((size_t)std::max<size_t>(big_track.size(), 1) * 4);
    const size_t tmpb = al(stage_tmp_bytes(in, DIM)), ordb = al(in.order ? (size_t)N * 4 : 0);
    c->s_clu0.ensure(xb + nbz + 2 * ib + bb + 2 * al((size_t)nb * 16 * 4) + pb + 2 * rsb + rgb + 3 * bigb + tmpb + ordb + 4096);
    c->s_clu1.ensure((size_t)T * T * sizeof(double) + (size_t)T * 64 + 4096);
    uint8_t* p = c->s_clu0.as<uint8_t>();
    auto take = [&](size_t bytes) { uint8_t* q = p; p += bytes; return q; };
    double* dX = (double*)take(xb); double* dN = (double*)take(nbz);
    int* dRT = (int*)take(ib); int* dRS = (int*)take(ib);
    int* dBC = (int*)take(bb);
    int* dST = (int*)take(al((size_t)nb * 16 * 4)); int* dSP = (int*)take(al((size_t)nb * 16 * 4));
    double* dP = (double*)take(pb);
    int* dRow = (int*)take(rsb); int* dRange = (int*)take(rgb);
    int* dBigT = (int*)take(bigb); int* dBigC0 = (int*)take(bigb); int* dBigNc = (int*)take(bigb);
    int* dIsBig = (int*)take(rsb);
    uint8_t* dTmp = take(tmpb); int32_t* dOrder = (int32_t*)take(ordb);
    double* dD = c->s_clu1.as<double>();
    auto up = [&](void* d, const void* h, size_t bytes) { if (bytes) HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream)); };
    stage_table(c, in, N, DIM, dX, dTmp, dOrder);
    up(dRT, row_track.data(), (size_t)N * 4); up(dRS, row_segidx.data(), (size_t)N * 4);
    up(dBC, blk_cont.data(), (size_t)nb * 4);
    up(dST, seg_track.data(), (size_t)nb * 16 * 4); up(dSP, seg_part.data(), (size_t)nb * 16 * 4);
    up(dRow, row_start, (size_t)(T + 1) * 4); up(dRange, range_b0.data(), (size_t)(n_ranges + 1) * 4);
    up(dIsBig, is_big.data(), (size_t)T * 4);
    up(dBigT, big_track.data(), big_track.size() * 4); up(dBigC0, big_c0.data(), big_c0.size() * 4); up(dBigNc, big_nc.data(), big_nc.size() * 4);
    // the staging buffers above are std::vectors: the copies must have run before they go out of scope
    HIP_CHECK(hipStreamSynchronize(c->stream));
    {
        ProfScope ps(c, "pdist");
        hipLaunchKernelGGL(row_norms_k, dim3((N + 255) / 256), dim3(256), 0, c->stream, dX, N, DIM, dN);
        PtArgs a;
        a.X = dX; a.nrm = dN; a.N = N; a.T = T; a.row_track = dRT; a.row_segidx = dRS;
        a.blk_cont = dBC; a.seg_track = dST; a.seg_part = dSP; a.range_b0 = dRange; a.row_start = dRow; a.D = dD; a.P = dP;
        a.n_blocks = nb; a.n_ranges = n_ranges; a.t0 = t0; a.t1 = t1; a.metric = metric;
        hipLaunchKernelGGL(pair_tiles_k, dim3(row_groups, n_ranges), dim3(256), 0, c->stream, a);
        if (t1 > t0) hipLaunchKernelGGL(pair_norm_k, dim3((T + 255) / 256, t1 - t0), dim3(256), 0, c->stream, dD, dRow, dIsBig, T, t0);
        // tracks of the requested range that span several blocks
        std::vector<int> sel;
        for (size_t k = 0; k < big_track.size(); ++k) if (big_track[k] >= t0 && big_track[k] < t1) sel.push_back((int)k);
        if (!sel.empty()) {
            const int k0 = sel.front(), nsel = (int)sel.size();     // (big_track is sorted: those inside [t0, t1) are consecutive entries)
            hipLaunchKernelGGL(pair_chunks_k, dim3((T + 255) / 256, nsel), dim3(256), 0, c->stream, dP, dBigT + k0, dBigC0 + k0, dBigNc + k0, dRow, T, dD);
        }
        if (mirror) mirror_upper_dev(c, dD, T);
    }
    HIP_CHECK(hipGetLastError());
    copy_rows_out(c, dD, T, t0, t1, out);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (d_D_keep) *d_D_keep = dD;
}

// Upper-triangle entries D[i][j], i < j, of the tracks i in [t0, t1) (all columns j > i): the unit of work when several GPUs split the
// pairwise distances of one global clustering -- a track's rows all live in one contiguous block, so every entry is produced by the
// same chain of additions the single-GPU call runs.  mirror: also D[j][i] = D[i][j] (whole matrix only: t0 = 0, t1 = T), as the
// reference stores it.  Entries j <= i of the computed rows are zero without mirror.
void pair_mean_dist_dev(Ctx* c, const PairInput& in, int N, int dim, const int32_t* row_start, int T, const PairOutput& out, double** d_D_keep,
                        int t0, int t1, int metric, bool mirror)
{
    if (t1 < 0) t1 = T;
    PVF_REQUIRE(0 <= t0 && t0 <= t1 && t1 <= T, "pair_mean_dist: bad track range");
    PVF_REQUIRE(N > 0 && T > 0 && dim > 0 && dim <= 4096, "pair_mean_dist: bad sizes");
    PVF_REQUIRE(row_start[0] == 0 && row_start[T] == N, "pair_mean_dist: row_start must cover [0, N)");
    PVF_REQUIRE(metric == 0 || metric == 1, "pair_mean_dist: metric 0 (euclidean) or 1 (cosine)");
    PVF_REQUIRE(!mirror || (t0 == 0 && t1 == T), "pair_mean_dist: only a whole matrix can be mirrored");
    PVF_REQUIRE((in.X != nullptr) != (in.emb != nullptr), "pair_mean_dist: one input table");
    if (in.emb) {
        PVF_REQUIRE(in.emb_stride >= (int64_t)dim * 4 && in.emb_stride % 4 == 0 && in.n_src >= 1 && in.decimals <= 15, "pair_mean_dist: bad float32 rows");
        if (in.order) for (int k = 0; k < N; ++k) PVF_REQUIRE(in.order[k] >= 0 && in.order[k] < in.n_src, "pair_mean_dist: row order out of range");
        else PVF_REQUIRE(in.n_src >= N, "pair_mean_dist: fewer float32 rows than N");
    }
    if (dim == 128) { pair_mean_dist_mfma(c, in, N, row_start, T, out, d_D_keep, t0, t1, metric, mirror); return; }
    PVF_REQUIRE(metric == 0, "pair_mean_dist: cosine is implemented for 128-D rows");
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t xb = al((size_t)N * dim * sizeof(double));
    const size_t sb = al((size_t)N * T * sizeof(double)), db = (size_t)T * T * sizeof(double), rb = al((size_t)(T + 1) * sizeof(int32_t));
    const size_t tmpb = al(stage_tmp_bytes(in, dim)), ordb = al(in.order ? (size_t)N * 4 : 0);
    c->s_clu0.ensure(2 * xb + sb + rb + tmpb + ordb + 512);
    c->s_clu1.ensure(db + (size_t)T * 64 + 4096);
    uint8_t* p = c->s_clu0.as<uint8_t>();
    double* dX = reinterpret_cast<double*>(p); p += xb;
    double* dXt = reinterpret_cast<double*>(p); p += xb;
    double* dS = reinterpret_cast<double*>(p); p += sb;
    int32_t* dR = reinterpret_cast<int32_t*>(p); p += rb;
    uint8_t* dTmp = p; p += tmpb;
    int32_t* dOrder = reinterpret_cast<int32_t*>(p);
    double* dD = c->s_clu1.as<double>();
    stage_table(c, in, N, dim, dX, dTmp, dOrder);
    HIP_CHECK(hipMemcpyAsync(dR, row_start, (size_t)(T + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, "pdist");
        hipLaunchKernelGGL(transpose_k, dim3((unsigned)(((size_t)N * dim + 255) / 256)), dim3(256), 0, c->stream, dX, N, dim, dXt);
        const int a0 = row_start[t0], a1 = row_start[t1];
        if (a1 > a0)
            hipLaunchKernelGGL(row_track_sums_k, dim3(a1 - a0), dim3(256), (dim + PD_CHUNK) * sizeof(double), c->stream, dX, dXt, N, dim, dR, T, dS, a0);
        if (t1 > t0)
            hipLaunchKernelGGL(track_pair_mean_k, dim3((T + 255) / 256, t1 - t0), dim3(256), 0, c->stream, dS, dR, T, dD, t0);
        if (mirror) mirror_upper_dev(c, dD, T);
    }
    HIP_CHECK(hipGetLastError());
    copy_rows_out(c, dD, T, t0, t1, out);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (d_D_keep) *d_D_keep = dD;
}

// ---------------------------------------------------------------------------------------------------
struct HacState {
    double* D; int T;
    double* rmin; int* rarg; int* alive; int* dirty; double* size;
    double* best;   // [0]=bi [1]=bj [2]=dist [3]=done
    double* log;    // [(T-1)][4]
    int* n_merges;
    double threshold;
};

__global__ void __launch_bounds__(256) hac_row_min_k(HacState h, int only_dirty)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (only_dirty && !h.dirty[r]) return;
    double bv = INFINITY; int bi = 0x7fffffff;
    if (h.alive[r]) {
        for (int j = r + 1 + tid; j < h.T; j += 256) {
            if (!h.alive[j]) continue;
            const double v = h.D[(size_t)r * h.T + j];
            if (v < bv) { bv = v; bi = j; }
        }
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double ov = sv[tid + off]; const int oi = si[tid + off];
            if (ov < sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) { h.rmin[r] = sv[0]; h.rarg[r] = si[0]; h.dirty[r] = 0; }
}

__global__ void __launch_bounds__(1024) hac_argmin_k(HacState h)
{
    __shared__ double sv[1024];
    __shared__ int si[1024];
    const int tid = threadIdx.x;
    if (h.best[3] != 0.0) return;
    double bv = INFINITY; int bi = 0x7fffffff;
    for (int r = tid; r < h.T; r += 1024) {
        const double v = h.rmin[r];
        if (v < bv) { bv = v; bi = r; }   // ascending r per thread => first occurrence kept
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) {
            const double ov = sv[tid + off]; const int oi = si[tid + off];
            if (ov < sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (si[0] == 0x7fffffff || !(sv[0] <= h.threshold)) { h.best[3] = 1.0; return; }
        const int i = si[0], j = h.rarg[i];
        h.best[0] = i; h.best[1] = j; h.best[2] = sv[0];
        const int m = *h.n_merges;
        h.log[4 * m] = i; h.log[4 * m + 1] = j; h.log[4 * m + 2] = sv[0]; h.log[4 * m + 3] = h.size[i] + h.size[j];
        *h.n_merges = m + 1;
    }
}

__global__ void __launch_bounds__(256) hac_merge_k(HacState h)
{
    if (h.best[3] != 0.0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int bi = (int)h.best[0], bj = (int)h.best[1];
    if (k >= h.T) return;
    const double si = h.size[bi], sj = h.size[bj];
    if (k == bi) { h.dirty[bi] = 1; return; }
    if (k == bj) { h.dirty[bj] = 1; return; }
    if (!h.alive[k]) return;
    const double v = (si * h.D[(size_t)bi * h.T + k] + sj * h.D[(size_t)bj * h.T + k]) / (si + sj);
    h.D[(size_t)bi * h.T + k] = v;
    h.D[(size_t)k * h.T + bi] = v;
    if (k < bi) {
        const int ra = h.rarg[k];
        if (ra == bi || ra == bj) h.dirty[k] = 1;
        else if (v < h.rmin[k] || (v == h.rmin[k] && bi < ra)) { h.rmin[k] = v; h.rarg[k] = bi; }
    } else if (k < bj) {
        if (h.rarg[k] == bj) h.dirty[k] = 1;
    }
}

__global__ void hac_finish_merge_k(HacState h)
{
    if (h.best[3] != 0.0) return;
    const int bi = (int)h.best[0], bj = (int)h.best[1];
    h.size[bi] = h.size[bi] + h.size[bj];
    h.alive[bj] = 0;
}

__global__ void __launch_bounds__(256) hac_init_k(HacState h, const int32_t* __restrict__ row_start)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { h.best[0] = h.best[1] = h.best[2] = h.best[3] = 0.0; *h.n_merges = 0; }
    if (k >= h.T) return;
    h.alive[k] = 1; h.dirty[k] = 0; h.size[k] = (double)(row_start[k + 1] - row_start[k]);
}

// The whole agglomeration in ONE workgroup: row minima and their arguments live in LDS (14 bytes per track + one alive bit: up to 10 240
// tracks), a merge is argmin over LDS -> size-weighted update of row / column mi in HBM -> re-scan of the rows whose cached minimum died.
// Same decisions as the kernels above (first minimum in row-major order, rows re-scanned from scratch), no launch or host round trip
// inside the loop.  A merge is two dependent trips to HBM, whatever T: a thread requests the <= U entries it owns of rows mi and mj (and
// the two sizes) in one batch before it uses any, the new minimum of row mi comes out of that same pass (its new entries are in
// registers), and the other rows to re-scan are read HAC_G - 1 at a time, again one batch per thread, and reduced together.
#define HAC_PERSIST_MAX_T 10200
#define HAC_G 4                          // rows reduced together after a merge: row mi + 3 re-scanned rows, then 3 per further round
static inline size_t hac_persist_lds(int T) { return ((size_t)T * 14 + 3) / 4 * 4 + (size_t)((T + 31) / 32) * 4; }

// (value, index) minimum over a wave, smaller index on equal values, left in every lane.  Two DPP reductions (row_shr 1 / 2 / 4 / 8 inside
// the rows of 16 lanes, then row_bcast:15 and row_bcast:31 carry the row results up to lane 63): first the value, then the smallest index
// among the lanes that hold it -- no trips through the LDS crossbar, no compare-and-select chains.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64(double v)
{
    return __hiloint2double(dpp_i32<CTRL, ROWS>(__double2hiint(v)), dpp_i32<CTRL, ROWS>(__double2loint(v)));
}
__device__ __forceinline__ void wave_argmin(double& v, int& i)
{
    double m = v;
    m = __builtin_fmin(m, dpp_f64<0x111, 0xf>(m));
    m = __builtin_fmin(m, dpp_f64<0x112, 0xf>(m));
    m = __builtin_fmin(m, dpp_f64<0x114, 0xf>(m));
    m = __builtin_fmin(m, dpp_f64<0x118, 0xf>(m));
    m = __builtin_fmin(m, dpp_f64<0x142, 0xa>(m));
    m = __builtin_fmin(m, dpp_f64<0x143, 0xc>(m));
    const double vmin = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 63), __builtin_amdgcn_readlane(__double2loint(m), 63));
    int c = (v == vmin) ? i : 0x7fffffff;
    c = min(c, dpp_i32<0x111, 0xf>(c));
    c = min(c, dpp_i32<0x112, 0xf>(c));
    c = min(c, dpp_i32<0x114, 0xf>(c));
    c = min(c, dpp_i32<0x118, 0xf>(c));
    c = min(c, dpp_i32<0x142, 0xa>(c));
    c = min(c, dpp_i32<0x143, 0xc>(c));
    v = vmin;
    i = __builtin_amdgcn_readlane(c, 63);
}
__device__ __forceinline__ void pick16(const double* wv, const int* wi, double& v, int& i)
{
    v = wv[0]; i = wi[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
        const double ov = wv[w]; const int oi = wi[w];
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

template <int U>
__global__ void __launch_bounds__(1024) hac_persist_k(HacState h)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int T = h.T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* rmin = reinterpret_cast<double*>(hsm);
    int* rarg = reinterpret_cast<int*>(hsm + (size_t)T * 8);
    uint16_t* dlist = reinterpret_cast<uint16_t*>(rarg + T);                     // rows to re-scan after this merge
    uint32_t* abits = reinterpret_cast<uint32_t*>(hsm + ((size_t)T * 14 + 3) / 4 * 4);   // bit k: track k has not been merged away
    __shared__ double wvA[16];
    __shared__ int wiA[16];
    __shared__ double wvB[HAC_G][16];
    __shared__ int wiB[HAC_G][16];
    __shared__ int s_nd;
    double* const D = h.D;
    for (int r = tid; r < T; r += 1024) { rmin[r] = h.rmin[r]; rarg[r] = h.rarg[r]; }
    for (int w = tid; w < (T + 31) / 32; w += 1024) abits[w] = (32 * w + 32 <= T) ? 0xffffffffu : ((1u << (T - 32 * w)) - 1u);
    __syncthreads();
    auto alive = [&](int k) { return (abits[k >> 5] >> (k & 31)) & 1u; };
    int merges = 0;
    while (merges < T - 1) {
        // ---- the closest pair: first minimum of the row minima
        double bv = INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = tid + 1024 * u;
            if (r < T) { const double v = rmin[r]; if (v < bv) { bv = v; bi = r; } }   // ascending r per thread: first occurrence kept
        }
        wave_argmin(bv, bi);
        if (lane == 0) { wvA[wave] = bv; wiA[wave] = bi; }
        if (tid == 0) s_nd = 0;
        __syncthreads();
        pick16(wvA, wiA, bv, bi);
        if (bi == 0x7fffffff || !(bv <= h.threshold)) break;      // uniform: every thread holds the same (bv, bi)
        // (the pair comes out of LDS: tell the compiler it is wave-uniform, so that row addresses are scalar arithmetic)
        const int mi = __builtin_amdgcn_readfirstlane(bi), mj = __builtin_amdgcn_readfirstlane(rarg[bi]);
        // ---- row / column mi <- size-weighted mean of rows mi and mj; everything a thread needs is requested before anything is used
        double a[U], b[U];
        uint32_t live = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = tid + 1024 * u;
            if (k < T && k != mi && k != mj && alive(k)) live |= 1u << u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = tid + 1024 * u;
            a[u] = 0.0; b[u] = 0.0;
            if (live >> u & 1) {
                a[u] = D[(size_t)mi * T + k];
                b[u] = D[(size_t)mj * T + k];
            }
        }
        const double szi = h.size[mi], szj = h.size[mj];
        if (tid == 0) {
            h.log[4 * merges] = mi; h.log[4 * merges + 1] = mj; h.log[4 * merges + 2] = bv; h.log[4 * merges + 3] = szi + szj;
        }
        // the quotient by the (uniform) sum of the sizes: reciprocal once, then q = n y corrected twice through the exact residual
        // n - den q (Markstein: with y the correctly rounded reciprocal the result is the correctly rounded quotient, i.e. what the
        // division instruction sequence returns -- 35 instructions per entry otherwise; checked on 1e8 cases, tests/test_host_logic.py)
        const double den = szi + szj, yrc = 1.0 / den;
        double v0 = INFINITY; int j0 = 0x7fffffff;             // the new first minimum of row mi over its alive columns k > mi
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!(live >> u & 1)) continue;
            const int k = tid + 1024 * u;
            const double num = szi * a[u] + szj * b[u];
            double v = num * yrc;
            v = __builtin_fma(__builtin_fma(-den, v, num), yrc, v);
            v = __builtin_fma(__builtin_fma(-den, v, num), yrc, v);
            D[(size_t)mi * T + k] = v;
            D[(size_t)k * T + mi] = v;
            if (k < mi) {
                const int ra = rarg[k];
                if (ra == mi || ra == mj) dlist[atomicAdd(&s_nd, 1)] = (uint16_t)k;
                else if (v < rmin[k] || (v == rmin[k] && mi < ra)) { rmin[k] = v; rarg[k] = mi; }
            } else {
                if (k < mj && rarg[k] == mj) dlist[atomicAdd(&s_nd, 1)] = (uint16_t)k;
                if (v < v0) { v0 = v; j0 = k; }
            }
        }
        if (tid == 0) {
            atomicAnd(&abits[mj >> 5], ~(1u << (mj & 31)));
            rmin[mj] = INFINITY; rarg[mj] = 0x7fffffff;
        }
        __syncthreads();
        const int nd = __builtin_amdgcn_readfirstlane(s_nd);
        if (tid == 0) h.size[mi] = szi + szj;               // (after the barrier: every thread has read the two sizes)
        // ---- row mi (from registers) and the rows whose cached minimum died: first minimum over the alive columns j > r
        uint32_t am = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = tid + 1024 * u;
            if (j < T && alive(j)) am |= 1u << u;
        }
        int q = 0;
        bool first = true;
        do {
            int row[HAC_G];
            double sv[HAC_G]; int sj[HAC_G];
            int nrow = 0;
            row[0] = mi; nrow = 1;                                              // slot 0: row mi in the first round, unused afterwards
            while (nrow < HAC_G && q < nd) row[nrow++] = __builtin_amdgcn_readfirstlane((int)dlist[q++]);
            double ld[HAC_G - 1][U];                                            // (slot 0 is row mi's, never loaded)
#pragma unroll
            for (int g = 1; g < HAC_G; ++g)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = tid + 1024 * u;
                    ld[g - 1][u] = INFINITY;
                    if (g < nrow && (am >> u & 1) && j > row[g]) ld[g - 1][u] = D[(size_t)row[g] * T + j];
                }
#pragma unroll
            for (int g = 0; g < HAC_G; ++g) {
                if (g >= nrow) continue;
                if (g == 0) { if (!first) continue; sv[g] = v0; sj[g] = j0; }
                else {
                    sv[g] = INFINITY; sj[g] = 0x7fffffff;
#pragma unroll
                    for (int u = 0; u < U; ++u) if (ld[g - 1][u] < sv[g]) { sv[g] = ld[g - 1][u]; sj[g] = tid + 1024 * u; }
                }
                wave_argmin(sv[g], sj[g]);
                if (lane == 0) { wvB[g][wave] = sv[g]; wiB[g][wave] = sj[g]; }
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < HAC_G; ++g)
                if (g < nrow && tid == g && (g > 0 || first)) {
                    double v; int i;
                    pick16(wvB[g], wiB[g], v, i);
                    rmin[row[g]] = v; rarg[row[g]] = i;
                }
            __syncthreads();
            first = false;
        } while (q < nd);
        ++merges;
    }
    if (tid == 0) *h.n_merges = merges;
}

int hac_dev(Ctx* c, double* d_D, const int32_t* row_start, int T, double threshold, int32_t* labels, double* merge_log)
{
    for (int i = 0; i < T; ++i) labels[i] = i;
    if (T < 2) return 0;
    const size_t need = (size_t)T * (8 + 4 + 4 + 4 + 8) + 4 * 8 + (size_t)T * 4 * 8 + 64 + (size_t)(T + 1) * 4 + 512;
    c->s_misc.ensure(need);
    uint8_t* p = c->s_misc.as<uint8_t>();
    auto take = [&](size_t bytes) { uint8_t* q = p; p += (bytes + 63) / 64 * 64; return q; };
    // note: the bump sizes above leave slack for the 64-byte rounding
    c->s_misc.ensure(need + 64 * 10);
    p = c->s_misc.as<uint8_t>();
    HacState h;
    h.D = d_D; h.T = T; h.threshold = threshold;
    h.rmin = (double*)take((size_t)T * 8);
    h.size = (double*)take((size_t)T * 8);
    h.best = (double*)take(4 * 8);
    h.log = (double*)take((size_t)T * 4 * 8);
    h.rarg = (int*)take((size_t)T * 4);
    h.alive = (int*)take((size_t)T * 4);
    h.dirty = (int*)take((size_t)T * 4);
    h.n_merges = (int*)take(64);
    int32_t* dR = (int32_t*)take((size_t)(T + 1) * 4);
    HIP_CHECK(hipMemcpyAsync(dR, row_start, (size_t)(T + 1) * 4, hipMemcpyHostToDevice, c->stream));
    ProfScope ps(c, "hac");
    hipLaunchKernelGGL(hac_init_k, dim3((T + 255) / 256), dim3(256), 0, c->stream, h, dR);
    hipLaunchKernelGGL(hac_row_min_k, dim3(T), dim3(256), 0, c->stream, h, 0);
    double hbest[4];
    if (T <= HAC_PERSIST_MAX_T) {
        const size_t lds = hac_persist_lds(T);           // rmin f64, rarg i32, re-scan list u16, alive bits
        static std::atomic<uint64_t> attr_on{0};          // per device (a function attribute belongs to the device it was set on)
        const uint64_t dev_bit = 1ull << (c->device & 63);
        if (!(attr_on.load() & dev_bit)) {
            HIP_CHECK(hipFuncSetAttribute((const void*)hac_persist_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hac_persist_lds(1024)));
            HIP_CHECK(hipFuncSetAttribute((const void*)hac_persist_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hac_persist_lds(3072)));
            HIP_CHECK(hipFuncSetAttribute((const void*)hac_persist_k<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hac_persist_lds(HAC_PERSIST_MAX_T)));
            attr_on.fetch_or(dev_bit);
        }
        if (T <= 1024) hipLaunchKernelGGL(hac_persist_k<1>, dim3(1), dim3(1024), lds, c->stream, h);
        else if (T <= 3072) hipLaunchKernelGGL(hac_persist_k<3>, dim3(1), dim3(1024), lds, c->stream, h);
        else hipLaunchKernelGGL(hac_persist_k<10>, dim3(1), dim3(1024), lds, c->stream, h);
    } else
    for (int it = 0; it < T - 1; ++it) {
        hipLaunchKernelGGL(hac_argmin_k, dim3(1), dim3(1024), 0, c->stream, h);
        hipLaunchKernelGGL(hac_merge_k, dim3((T + 255) / 256), dim3(256), 0, c->stream, h);
        hipLaunchKernelGGL(hac_finish_merge_k, dim3(1), dim3(1), 0, c->stream, h);
        hipLaunchKernelGGL(hac_row_min_k, dim3(T), dim3(256), 0, c->stream, h, 1);
        if ((it & 63) == 63) {
            HIP_CHECK(hipMemcpyAsync(hbest, h.best, sizeof hbest, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (hbest[3] != 0.0) break;
        }
    }
    HIP_CHECK(hipGetLastError());
    int n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, h.n_merges, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<double> log((size_t)std::max(n, 1) * 4);
    if (n > 0) {
        HIP_CHECK(hipMemcpyAsync(log.data(), h.log, (size_t)n * 4 * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    for (int m = 0; m < n; ++m) {
        const int a = (int)log[4 * m], b = (int)log[4 * m + 1];
        for (int k = 0; k < T; ++k) if (labels[k] == b) labels[k] = a;
        if (merge_log) memcpy(merge_log + 4 * m, &log[4 * m], 4 * sizeof(double));
    }
    return n;
}

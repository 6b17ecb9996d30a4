// detect.hip -- K1..K4: image pyramid, FHOG, filter scoring, NMS  (replaces dlib.get_frontal_face_detector()(rgb, 1);
// reference pyannote/video/face/face.py:54,66).  All arithmetic follows the orders stated in oracle/pvo_fhog.c and
// oracle/pvo_detect.c so that boxes are bit-identical; the code itself is written for gfx950 (wave64, LDS tiles).
#include "detect_ml.h"
#include <algorithm>
#include <thread>
#include <cmath>
#include <cstdlib>

// =====================================================================================================
// K1: bilinear resize (pyramid_up / pyramid_down<6>), uint8 RGB HWC, double coordinates, (v + 0.5) truncation
// =====================================================================================================
// One lane = one output column, walking RS consecutive output rows.  The horizontal blend of a source row,
//   H_s = (1 - lr) * S[s][left] + lr * S[s][right],
// depends only on (s, column) and consecutive output rows share source rows, so each H_s is evaluated once and kept in registers;
// the value written is
//   v = (1 - tb) * H_top + tb * H_bottom ; out = (uint8)(v + 0.5)      (oracle/pvo_image.c).
// A wave first requests EVERY source row its RS output rows need (one coalesced dword per lane and row: the 64 columns of a wave span
// < 256 source bytes for scales <= 1.25) and parks them in LDS; after that single wait the strip runs on LDS + VALU only.
//
// Round 5: the arithmetic above is untouched (the same fp64 products and sums in the same order: bit-exact by construction); what went
// away is what stood around it -- 69 -> about 45 vector instructions per 64-pixel row (hipcc -S):
//   * source rows are requested through a buffer descriptor over the frame (a dword past the frame reads as 0, never used): one
//     buffer_load with a SCALAR row offset per row instead of a 64-bit address, a compare against the frame's last word and two selects
//     per lane and row (7 vector instructions per requested row -> 0);
//   * a lane reads its two source pixels as six BYTES straight out of the parked dwords (ds_read_u8: the LDS unit extracts them) and
//     converts them with v_cvt_f64_u32 -- round 3's form converted the window to floats (4 v_cvt_f32_ubyte), parked those (16-byte
//     write), read six back and converted again (6 v_cvt_f64_f32): 13 -> 8 vector instructions per new source row, the same six LDS reads;
//   * the walk follows the SOURCE rows and its body exists twice with the two H register triples in swapped roles: nothing is ever
//     copied (round 3's two-entry cache moved 3 doubles and ran 6 selects per output row);
//   * the 64 result pixels leave as 48 aligned dwords through LDS: a lane writes its 3 bytes (ds_write_b8), lanes 0..47 read one dword
//     each and store it -- no packing, no ds_bpermute pair, no funnel shift (8 vector instructions -> 0, two more LDS instructions);
//   * (later in round 5) the output image sits behind a descriptor as well -- the row offset is a scalar operand of the store, a lane
//     past the segment's end stores out of range instead of being masked off -- and the vertical weights are read by the products from
//     scalar registers: 42 -> about 36 vector instructions per row.  That bought 2 %: with the three pipes measured one at a time
//     (profiles/r05_resize_bounds_experiment.txt: loads and stores alone 4.29 ms, arithmetic alone 4.91 ms, together 6.2 ms per 125
//     frames) the kernel is no longer bound by one of them -- HBM at 62 % of what streams, vector issue at 53 %, LDS at about half.
// Needs 4-byte aligned output rows (level images of the batched path have a padded row pitch).
#define RESIZE_MAXS 22
// per output row of a resize stage, computed once on the host with the oracle's double arithmetic (y = r * y_scale; top = floor(y);
// bottom = min(top + 1, ih - 1); tb = y - top; tb1 = 1 - tb): the row loop reads it through the scalar cache instead of redoing
// wave-uniform double arithmetic in every lane.  32 bytes = one aligned s_load_dwordx8.
struct RowTab { int32_t top, bottom; double tb, tb1, pad; };
// ... and per output column (round 6): left source column and the two horizontal weights, likewise host data.  HOW the source coordinates
// are generated (c * x_scale as restated in oracle/pvo_image.c, or by accumulation as dlib's loop may do it: oracle/EXT_REGISTER.md E1)
// is therefore decided in ONE place on the host (resize_coords) for rows and columns alike, and the kernel does not change with it.
struct ColTab { int32_t left, pad; double lr, lr1, pad2; };

template <int RS, int NSTRIP>
__global__ void __launch_bounds__(256) resize_rows_k(const uint8_t* const* __restrict__ in_ptrs, const uint8_t* __restrict__ in_base,
                                                     size_t in_stride, int in_rb, int ih, int iw, uint8_t* __restrict__ out,
                                                     size_t out_stride, int out_rb, int oh, int ow,
                                                     const RowTab* __restrict__ rows, const ColTab* __restrict__ cols)
{
    constexpr int MAXS = (RS == 16) ? RESIZE_MAXS : (RS * 5) / 4 + 3;     // source rows of a strip at the pyramid's largest step (6/5 down)
    __shared__ uint32_t s_rows[4][MAXS][64];                     // a wave's source rows: 256-byte windows, dword per lane
    __shared__ uint32_t s_out[4][64];                            // a wave's output row: 64 pixels x 3 bytes = 48 dwords
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = blockIdx.x * 256 + wave * 64;               // first column of this wave's segment
    const int b = blockIdx.z;
    if (c0 >= ow) return;                                      // wave-uniform
    const int c = min(c0 + lane, ow - 1);                      // lanes past the row end recompute the last column (never stored)
    const uint8_t* in = in_ptrs ? in_ptrs[b] : in_base + (size_t)b * in_stride;
    uint8_t* ob = out + (size_t)b * out_stride + (size_t)c0 * 3;
    const ColTab ct = cols[c];                                 // (once per wave and segment: 32 bytes per lane, coalesced)
    const int left = ct.left;
    const int left0 = __builtin_amdgcn_readfirstlane(left);    // lane 0 holds column c0: the smallest source column of the wave
    const bool has_right = (left + 1 <= iw - 1);
    const double lr = ct.lr, lr1 = ct.lr1;
    const int seg_bytes = min(ow - c0, 64) * 3;
    const bool stores = (4 * lane < seg_bytes);
    // the frame behind a buffer descriptor whose base is 4-byte aligned (`delta` = what the alignment cut off): a row's window starts at
    // the aligned dword that holds the wave's first source byte; dwords past the frame's end read as zero
    const unsigned delta = (unsigned)((uintptr_t)in & 3u);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((uintptr_t)in - delta), 0, (int)(delta + (unsigned)ih * (unsigned)in_rb), 0x00020000);
    const int lane4 = 4 * lane;
    // a wave writes NSTRIP strips of RS output rows, one below the other; the next strip's source rows are requested (into registers)
    // before the current strip's arithmetic starts and go to LDS when it is done: the wave's memory phase lies under its compute phase
    int r0 = blockIdx.y * (RS * NSTRIP), r_end = 0, s_first = 0;
    unsigned win0 = 0;                                                       // byte offset of row s_first's first needed byte
    uint32_t t[MAXS];
    auto request = [&](int q0, int& sf, unsigned& w0) {
        const int qe = min(q0 + RS, oh);
        sf = rows[q0].top;
        const int nr = rows[qe - 1].bottom - sf + 1;
        w0 = delta + (unsigned)sf * (unsigned)in_rb + 3u * (unsigned)left0;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            t[k] = 0;
            if (k < nr) t[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, (int)((w0 + (unsigned)k * (unsigned)in_rb) & ~3u), 0);     // wave-uniform test and offset
        }
    };
    request(r0, s_first, win0);
    // this lane's two source pixels inside a window, in bytes from the window's first NEEDED byte (the row's alignment offset is added
    // per row); a lane without a right neighbour blends its own pixel with itself (oracle/pvo_image.c)
    const uint8_t* wl = reinterpret_cast<const uint8_t*>(&s_rows[wave][0][0]) + 3 * (left - left0);
    const uint8_t* wr = wl + (has_right ? 3 : 0);
    uint8_t* so = reinterpret_cast<uint8_t*>(&s_out[wave][0]) + 3 * lane;
    const uint32_t* sd = &s_out[wave][0] + lane;
    auto hblend = [&](int srow, double* hh) {
        const int k = srow - s_first;                                              // wave-uniform
        const unsigned off = 256u * (unsigned)k + ((win0 + (unsigned)k * (unsigned)in_rb) & 3u);
        // six single-byte LDS reads (volatile: the compiler would fuse neighbours into 16-bit reads and spend vector instructions taking them apart again)
        typedef const volatile __attribute__((address_space(3))) uint8_t* lds_bytes;
        lds_bytes pl = (lds_bytes)(wl + off);
        lds_bytes pr = (lds_bytes)(wr + off);
        const uint32_t l0 = pl[0], l1 = pl[1], l2 = pl[2], q0 = pr[0], q1 = pr[1], q2 = pr[2];
        hh[0] = lr1 * (double)l0 + lr * (double)q0;
        hh[1] = lr1 * (double)l1 + lr * (double)q1;
        hh[2] = lr1 * (double)l2 + lr * (double)q2;
    };
    // the output image behind a descriptor too: the row's offset is a scalar operand of the store (no 64-bit address arithmetic per row)
    // and a lane past the segment's end stores out of range, which the descriptor drops (no exec-mask branch per row)
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)ob, 0, (int)((unsigned)oh * (unsigned)out_rb - 3u * (unsigned)c0), 0x00020000);
    const int st_off = stores ? lane4 : 0x7fffff00;
    auto emit = [&](int r, double tb, double tb1, const double* ht, const double* hb) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = tb1 * ht[k] + tb * hb[k];
            so[k] = (uint8_t)(v + 0.5);
        }
        __builtin_amdgcn_wave_barrier();                               // (LDS serves a wave's accesses in order; this only pins the compiler's order)
        const uint32_t dw = *sd;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_raw_buffer_store_b32(dw, ro, st_off, r * out_rb, 0);
    };
    // The walk goes down the SOURCE rows: with H of row s at hand the row below it (s + 1, or s itself on the image's last row: the table's
    // `bottom`) is blended into the other register triple and every output row whose `top` is s is written; then the two triples swap
    // ROLES -- the loop body exists twice, once per role, so no value is ever copied (round 3: a two-entry cache keyed by the output
    // row's (top, bottom), 3 doubles moved and 6 selects per output row).  Every H is evaluated once, in row order, as before.
    int r = 0, srow = 0;
    double hA[3], hB[3];
    // Table entries come through the scalar cache (constant address space: this kernel's stores cannot touch the table).  A source row's
    // walk step asks for the entries of the next TWO output rows before it blends (their latency hides behind that arithmetic) and
    // uses them as they are: the two weights stay in scalar registers and are read from there by the products.  (Carried from one
    // output row to the next -- round 4's prefetch -- the compiler keeps them in vector registers: two 64-bit moves per output row on
    // the pipe this kernel is bound by.)  The pyramid never writes more than two rows per source row (2x up: two; 6/5 down: one or
    // none); a third and later row fetch their entries on the spot.
    const __attribute__((address_space(4))) RowTab* crows = (const __attribute__((address_space(4))) RowTab*)rows;
    int top_r = 0;                                                              // `top` of output row r, the next one to write
    auto step = [&](const double* hc, double* hn) {
        const int ra = min(r, oh - 1), rb = min(r + 1, oh - 1), rc = min(r + 2, oh - 1);
        const double tbA = crows[ra].tb, tb1A = crows[ra].tb1, tbB = crows[rb].tb, tb1B = crows[rb].tb1;
        const int topB = crows[rb].top, topC = crows[rc].top;
        hblend(min(srow + 1, ih - 1), hn);
        if (top_r == srow) {                                                    // (r < r_end here: the walk's loop tests it)
            emit(r, tbA, tb1A, hc, hn);
            ++r;
            top_r = topB;
            if (r < r_end && top_r == srow) {
                emit(r, tbB, tb1B, hc, hn);
                ++r;
                top_r = topC;
                while (r < r_end && top_r == srow) {
                    emit(r, crows[r].tb, crows[r].tb1, hc, hn);
                    ++r;
                    top_r = crows[min(r, oh - 1)].top;
                }
            }
        }
        ++srow;
    };
#pragma unroll 1
    for (int strip = 0; strip < NSTRIP; ++strip) {
#pragma unroll
        for (int k = 0; k < MAXS; ++k) s_rows[wave][k][lane] = t[k];
        __builtin_amdgcn_wave_barrier();
        r_end = min(r0 + RS, oh);
        const int q0 = r0 + RS;
        const bool more = (strip + 1 < NSTRIP) && (q0 < oh);
        int n_first = 0;
        unsigned n_win0 = 0;
        if (more) request(q0, n_first, n_win0);
        r = r0; srow = s_first; top_r = s_first;
        hblend(srow, hA);
        while (r < r_end) {
            step(hA, hB);
            if (r >= r_end) break;
            step(hB, hA);
        }
        if (!more) break;
        __builtin_amdgcn_wave_barrier();                               // this strip's LDS reads stay before the next strip's writes
        r0 = q0; s_first = n_first; win0 = n_win0;
    }
}

// source coordinate of output index i of a resize stage, i = 0 .. n_out - 1, scale = (n_in - 1) / max(n_out - 1, 1).
// PVF_RESIZE_COORDS=accumulate: the coordinate is carried from index to index (v = -scale; v += scale per step) instead of formed as
// i * scale -- the other way dlib's resize_image may generate it (oracle/EXT_REGISTER.md E1; the oracle has the same switch,
// PVO_RESIZE_COORDS, and tests/test_gpu_parity.py runs the pyramid under both).  Read per plan: a context's plans are built once.
static std::vector<double> resize_coords(int n_in, int n_out)
{
    const double scale = (n_in - 1) / (double)std::max(n_out - 1, 1);
    const char* mode = getenv("PVF_RESIZE_COORDS");
    const bool accumulate = mode && strcmp(mode, "accumulate") == 0;
    PVF_REQUIRE(!mode || accumulate || strcmp(mode, "mul") == 0, "PVF_RESIZE_COORDS must be mul or accumulate");
    std::vector<double> v((size_t)n_out);
    double a = -scale;
    for (int i = 0; i < n_out; ++i) { a += scale; v[i] = accumulate ? a : i * scale; }
    return v;
}

static void fill_row_table(std::vector<RowTab>& t, int ih, int oh)
{
    const std::vector<double> ys = resize_coords(ih, oh);
    const size_t base = t.size();
    t.resize(base + oh);
    for (int r = 0; r < oh; ++r) {
        const double y = ys[r];
        RowTab& e = t[base + r];
        e.top = (int)std::floor(y);
        e.bottom = std::min(e.top + 1, ih - 1);
        e.tb = y - e.top;
        e.tb1 = 1 - e.tb;
        e.pad = 0;
    }
}

static void fill_col_table(std::vector<ColTab>& t, int iw, int ow)
{
    const std::vector<double> xs = resize_coords(iw, ow);
    const size_t base = t.size();
    t.resize(base + ow);
    for (int c = 0; c < ow; ++c) {
        ColTab& e = t[base + c];
        e.left = (int)std::floor(xs[c]);
        e.lr = xs[c] - e.left;
        e.lr1 = 1 - e.lr;
        e.pad = 0; e.pad2 = 0;
    }
}

static void launch_resize_rows(Ctx* c, const uint8_t* const* in_ptrs, const uint8_t* in_base, size_t in_stride, int in_rb, int ih, int iw,
                               uint8_t* out, size_t out_stride, int out_rb, int oh, int ow, int batch, const RowTab* d_rows, const ColTab* d_cols)
{
    const double x_scale = (iw - 1) / (double)std::max(ow - 1, 1);
    const double y_scale = (ih - 1) / (double)std::max(oh - 1, 1);
    constexpr int RS = 16;
    PVF_REQUIRE(x_scale <= 1.25 && y_scale <= 1.25 && (RS - 1) * y_scale + 3 <= RESIZE_MAXS, "resize_rows: scale outside the pyramid's range (2x up, 6/5 down)");
    PVF_REQUIRE(out_rb % 4 == 0 && out_stride % 4 == 0 && ((uintptr_t)out & 3) == 0 && out_rb >= (ow * 3 + 3) / 4 * 4, "resize: output rows must be 4-byte aligned");
    // two strips per wave, the second one's source rows in flight behind the first one's arithmetic: 6.20 -> 6.02 ms per 125 frames of 1080p
    // (4 strips 6.14, 8 strips 6.40: the grid of the small levels gets too coarse).  profiles/r05_resize_bounds_experiment.txt
    constexpr int NSTRIP = 2;
    static const int thin = getenv("PVF_RESIZE_THIN") ? atoi(getenv("PVF_RESIZE_THIN")) : 0;
    if (thin) {
        // strips of 8 output rows: 13 source rows of LDS per wave instead of 22 (13.3 KB per block instead of 23.5) -- room for a block of
        // the embedder beside six of these on a CU (profiles/r06_coissue_experiment.txt)
        dim3 grid8((ow + 255) / 256, (oh + 8 * NSTRIP - 1) / (8 * NSTRIP), batch);
        hipLaunchKernelGGL((resize_rows_k<8, NSTRIP>), grid8, dim3(256), 0, c->det_stream, in_ptrs, in_base, in_stride, in_rb, ih, iw, out, out_stride, out_rb, oh, ow, d_rows, d_cols);
        return;
    }
    dim3 grid((ow + 255) / 256, (oh + RS * NSTRIP - 1) / (RS * NSTRIP), batch);
    hipLaunchKernelGGL((resize_rows_k<RS, NSTRIP>), grid, dim3(256), 0, c->det_stream, in_ptrs, in_base, in_stride, in_rb, ih, iw, out, out_stride, out_rb, oh, ow, d_rows, d_cols);
}

static void pyramid_up_dims(int ih, int iw, int* oh, int* ow)
{
    const double right = ((iw - 1) + 1.25) * 2.0;
    const double bottom = ((ih - 1) + 0.75) * 2.0;
    *ow = (int)std::floor(right + 0.5) + 1;
    *oh = (int)std::floor(bottom + 0.5) + 1;
}

// =====================================================================================================
// K3: filter scoring.  score[f](r,c) = fmaf chain over (m, n, p) -- identical order to the oracle (oracle/pvo_detect.c).
// =====================================================================================================
// (ScoreParams, CandRec: detect_ml.h)

// =====================================================================================================
// host side: level schedule, rectangle mapping, canonical sort, NMS
// =====================================================================================================
static inline long iround(double v) { return (long)std::floor(v + 0.5); }
static void rect_down6(long r[4]) { const double ratio = (6 - 1.0) / 6; for (int i = 0; i < 4; ++i) r[i] = iround((r[i] - 0.3) * ratio + 0.3); }
static void rect_up6(long r[4]) { const double ratio = 6 / (6 - 1.0); for (int i = 0; i < 4; ++i) r[i] = iround((r[i] - 0.3) * ratio + 0.3); }
static void rect_down2i(long r[4])
{
    r[0] = iround(r[0] / 2.0 - 1.25); r[1] = iround(r[1] / 2.0 - 0.75);
    r[2] = iround(r[2] / 2.0 - 1.25); r[3] = iround(r[3] / 2.0 - 0.75);
}
static int detector_levels(int h, int w, const DetectorModel& m)
{
    long r[4] = {0, 0, w - 1, h - 1};
    int levels = 0;
    do { rect_down6(r); ++levels; } while ((r[2] - r[0] + 1) >= m.min_w && (r[3] - r[1] + 1) >= m.min_h && levels < m.max_levels);
    return levels;
}
static void fhog_to_image(long px, long py, int cell, int pad_r, int pad_c, long* ox, long* oy)
{
    long x = (px + 1 - (pad_c - 1) / 2) * cell + 1;
    long y = (py + 1 - (pad_r - 1) / 2) * cell + 1;
    x += (x >= 0) ? cell / 2 : -(cell / 2);
    y += (y >= 0) ? cell / 2 : -(cell / 2);
    *ox = x; *oy = y;
}

struct LevelDims { int h, w; };
static std::vector<LevelDims> level_schedule(int h, int w, int upsample, const DetectorModel& m, std::vector<LevelDims>* ups)
{
    int ch = h, cw = w;
    for (int u = 0; u < upsample; ++u) {
        int nh, nw;
        pyramid_up_dims(ch, cw, &nh, &nw);
        ch = nh; cw = nw;
        if (ups) ups->push_back({ch, cw});
    }
    const int levels = detector_levels(ch, cw, m);
    std::vector<LevelDims> out;
    out.push_back({ch, cw});
    for (int l = 1; l < levels; ++l) { ch = (5 * ch) / 6; cw = (5 * cw) / 6; out.push_back({ch, cw}); }
    return out;
}

// table of frame pointers for the kernels of one batch.  Two tables alternate (det_slot): the pinned host copy of a batch must
// survive until its asynchronous upload has run, and det_run_many has the next batch in flight before it collects this one.
static void upload_frame_ptrs(Ctx* c, const std::vector<Frame>& frames, const uint8_t*** d_ptrs)
{
    DevBuf& d = c->s_fptr[c->det_slot & 1];
    HostBuf& h = c->h_fptr[c->det_slot & 1];
    d.ensure(frames.size() * sizeof(void*));
    h.ensure(frames.size() * sizeof(void*));
    const uint8_t** hp = h.as<const uint8_t*>();
    for (size_t i = 0; i < frames.size(); ++i) hp[i] = frames[i].d;
    HIP_CHECK(hipMemcpyAsync(d.p, hp, frames.size() * sizeof(void*), hipMemcpyHostToDevice, c->det_stream));
    *d_ptrs = d.as<const uint8_t*>();
}

// =====================================================================================================
// All pyramid levels in one launch per stage.  HBM is large (288 GB): the whole pyramid of a batch stays resident
// (83 MB of images + 58 MB of features per 1080p frame), so after the (sequentially dependent) resize chain the fused FHOG pass
// and the scoring each run as ONE grid that covers every level -- 2 launches instead of 60 per batch, and the small levels
// fill the CUs the big ones leave idle.
// A block finds its level with a short scan of block-start offsets passed by value (kernarg / scalar cache).
// =====================================================================================================

// Logical block id of the multi-level kernels.  A contiguous-range-per-XCD order (so that vertically adjacent tiles meet in one
// L2) was measured 5-8 % SLOWER for the histogram / feature / scoring kernels (the ranges differ in cost per block, and the
// re-reads it saves are served by the MALL anyway), so blocks keep the hardware's round-robin order.
__device__ __forceinline__ int ml_block(const MlStarts& st) { return (int)blockIdx.x; }
static inline int ml_grid(int total) { return total; }

__device__ __forceinline__ int ml_level(const MlStarts& st, int g)
{
    int l = 0;
    while (l + 1 < st.nl && g >= st.b0[l + 1]) ++l;
    return l;
}

// ---------------------------------------------------------------------------------------------------
// K2: image rows -> 31-plane features in ONE pass; gradients and cell histograms never leave the chip.
//
// A strip is 64 histogram columns wide (lane L <-> histogram column hx0 + L) and a chunk of the level high; it is walked top to bottom,
// one pixel row at a time.  Per row and lane
//   - the image rows y-1, y, y+1 of the lane's own 8 pixel columns (each row loaded once, 32 bytes per lane) become 8 (magnitude, bin)
//     pairs: the colour channel with the largest |g|^2, its orientation bin from a table, the exact root;
//   - the lane's own 8 pairs and the 8 of the lane to its right are the 16 columns of its cell's window: their votes go to the bins of
//     the cell whose UPPER half the row lies in and of the cell whose LOWER half it lies in.
// A cell therefore receives its votes in row-major order of its own 16 x 16 window -- the order dlib's scatter loop produces
// (oracle/pvo_fhog.c) -- while every pixel's gradient is computed exactly once.  The bins live in LDS, [bin][lane] (conflict-free), the
// even and the odd cell rows' in two halves of ONE array a constant distance apart: a vote reads, adds to and writes the lane's own word
// in both cells with one ds_read2st64 / ds_write2st64 pair off one address, the chains of the two cells a row votes into are
// independent, and LDS executes a wave's accesses in order, so no wait sits between one vote's write and the next vote's read.
// (ds_add_f32 gives the same sums -- it is an IEEE add -- but the LDS atomic unit retires so few lanes per clock that the kernel ran 5 x
// slower with it.)  When a cell is complete its 18 bins move into registers; a finished cell row's features (4-way block normalisation
// over the 3 x 3 neighbourhood of cell energies: rows from the two previous cell rows kept in registers, columns from the neighbouring
// lanes) are written straight to the feature map.
// HBM traffic: the level images once (83 MB per 1080p frame) + the features once (58 MB).
//
// Strip geometry: feature column x needs the histograms x+1 .. x+3, so a strip of 64 lanes yields 61 feature columns
// (lane 63 only supplies gradients to lane 62; lanes 0 and 62 only supply cell energies).  Row chunks of `chunk_rows` feature
// rows re-walk 24 + 2 pixel rows of their upper neighbour (3 cell rows of histogram context).
#define FUSED_OUT 61
// value of the lane to the right / left (v_mov_b32_dpp wave_shl:1 / wave_shr:1; the last / first lane, which has no source, gets 0).
// The direction is checked once per context on the device (dpp_probe_k): a mismatch is an error, not a fallback.
// (bound_ctrl: a lane without a source reads 0 -- no zeroed destination to prepare, one v_mov less per move)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }

// ---------------------------------------------------------------------------------------------------
// The two halves of that work run in waves of their own (round 6).  Rounds 2-5 held the image rows, two rows' worth of gradients, the
// previous cell row's bins and the feature epilogue in ONE wave (fhog_fused_ml_k: 236 registers, two waves per SIMD), and those two spent
// their time differently -- 8 x ~35 vector instructions per pixel row for the gradients against a chain of 16 dependent LDS
// read-add-writes for the votes -- so whenever both sat in their chains the vector pipe idled.
// Here a block of 8 waves works on 4 strips; each strip has
//   a GRADIENT wave (role 0): image rows in registers, per pixel the channel with the largest |g|^2, its orientation bin from the
//     table and the exact root -- (magnitude, bin) of its 8 pixels per row go into a two-slot ring in LDS;
//   a VOTE wave (role 1): reads its own and its right neighbour's 8 pairs from the ring (the neighbour's come from LDS, not from 16 DPP
//     moves), casts the 16 votes into the two cell rows in the SAME order as before (a cell's bins receive their terms in row-major
//     order of its 16 x 16 window: bit-identical by construction), keeps the finished rows' bins and energies and writes the features.
// Neither role needs more than 128 registers => 4 waves per SIMD, two of each kind when the hardware deals a block's waves out to the
// SIMDs round robin (waves 0-3 gradient, 4-7 vote): the vote waves' LDS waits are filled by the gradient waves' arithmetic.
// Hand-over without block barriers: per strip two flags per slot in LDS, ready (row + 1, written by the gradient wave after the row's
// pairs) and done (row + 1, written by the vote wave once its reads are issued).  LDS executes one wave's instructions in order, so a
// wave that sees the flag sees the data written before it; a vote wave reads flag and data in one go and repeats the lot if the flag
// was not up yet (the gradient wave runs ahead: rare).
// The gradient itself is shorter too: the two differences of a channel are formed into the halves of ONE register (SDWA byte selects,
// the second one written into the upper word), |g|^2 is one v_dot2 of that register with itself, and the register of the winning
// channel IS the table index after a bit permutation (the table is laid out for it: 9-bit wrapped differences, 8 x 8 tiles).
#define FS_PAIRS 4
#define FS_COLS 65                                   // lanes of a ring row: 64 + the column lane 63 reads as its right neighbour (zeros)
#define FS_SLOT_DW (8 * FS_COLS * 2)                 // dwords of a slot: [pixel][lane](magnitude bits, bin)
#define FS_BINS_DW (2 * 18 * 64)                     // a vote wave's bins: [parity of the cell row][bin][lane]
#define FS_FLAG_DW (4 * 64)                          // a strip's flags, a word per lane: ready[2 slots][64], done[2 slots][64]
#define FS_LDS_BYTES (FS_PAIRS * (FS_BINS_DW + 2 * FS_SLOT_DW + FS_FLAG_DW) * 4)
static_assert(FS_COLS * 8 == 520, "the ring's column pitch is written into the instruction strings below");
typedef __attribute__((address_space(3))) uint32_t* lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_u32)p; }

// a gradient wave's hand-over: the 8 pairs of a row, then the row's flag.  One statement: LDS executes a wave's instructions in order,
// the compiler must not move or split them.
__device__ __forceinline__ void ring_write(uint32_t data_at, uint32_t flag_at, const u32x2* v, uint32_t row1)
{
    asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %0, %3 offset:520\n\tds_write_b64 %0, %4 offset:1040\n\tds_write_b64 %0, %5 offset:1560\n\t"
                 "ds_write_b64 %0, %6 offset:2080\n\tds_write_b64 %0, %7 offset:2600\n\tds_write_b64 %0, %8 offset:3120\n\t"
                 "ds_write_b64 %0, %9 offset:3640\n\tds_write_b32 %1, %10"
                 :: "v"(data_at), "v"(flag_at), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(row1) : "memory");
}
// a vote wave's: the row's flag and, behind it, its own and its right neighbour's 8 pairs
__device__ __forceinline__ uint32_t ring_read(uint32_t data_at, uint32_t flag_at, u32x2* own, u32x2* nb)
{
    uint32_t ready;
    asm volatile("ds_read_b32 %0, %18\n\t"
                 "ds_read_b64 %1, %17\n\tds_read_b64 %9, %17 offset:8\n\tds_read_b64 %2, %17 offset:520\n\tds_read_b64 %10, %17 offset:528\n\t"
                 "ds_read_b64 %3, %17 offset:1040\n\tds_read_b64 %11, %17 offset:1048\n\tds_read_b64 %4, %17 offset:1560\n\tds_read_b64 %12, %17 offset:1568\n\t"
                 "ds_read_b64 %5, %17 offset:2080\n\tds_read_b64 %13, %17 offset:2088\n\tds_read_b64 %6, %17 offset:2600\n\tds_read_b64 %14, %17 offset:2608\n\t"
                 "ds_read_b64 %7, %17 offset:3120\n\tds_read_b64 %15, %17 offset:3128\n\tds_read_b64 %8, %17 offset:3640\n\tds_read_b64 %16, %17 offset:3648\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(ready), "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3]), "=&v"(own[4]), "=&v"(own[5]), "=&v"(own[6]), "=&v"(own[7]),
                   "=&v"(nb[0]), "=&v"(nb[1]), "=&v"(nb[2]), "=&v"(nb[3]), "=&v"(nb[4]), "=&v"(nb[5]), "=&v"(nb[6]), "=&v"(nb[7])
                 : "v"(data_at), "v"(flag_at) : "memory");
    return ready;
}
__device__ __forceinline__ void flag_write(uint32_t flag_at, uint32_t v)
{
    asm volatile("ds_write_b32 %0, %1" :: "v"(flag_at), "v"(v) : "memory");
}

template <int A, int B> __device__ __forceinline__ uint32_t sub_into_upper(uint32_t acc, uint32_t x, uint32_t y);
// acc[31:16] = byte A of x - byte B of y (low 16 bits of the difference), acc[15:0] kept.  (The wait state after it: a VALU that writes
// part of a register needs one before the register is read on this family, and the compiler cannot see into the statement.)
#define PVF_SUBW1(A, B)                                                                                                                  \
    template <> __device__ __forceinline__ uint32_t sub_into_upper<A, B>(uint32_t acc, uint32_t x, uint32_t y)                           \
    {                                                                                                                                    \
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #A " src1_sel:BYTE_" #B "\n\ts_nop 0"   \
            : "+v"(acc) : "v"(x), "v"(y));                                                                                               \
        return acc;                                                                                                                      \
    }
PVF_SUBW1(0, 0) PVF_SUBW1(1, 1) PVF_SUBW1(2, 2) PVF_SUBW1(3, 3)
#undef PVF_SUBW1
typedef short s16x2 __attribute__((ext_vector_type(2)));

// (magnitude bits, bin) of pixel P of the centre row from the lane's three 32-byte row windows
template <int P>
__device__ __forceinline__ u32x2 split_grad_px(const uint32_t* up, const uint32_t* ce, const uint32_t* dn, const __amdgpu_buffer_rsrc_t& lut_rs,
                                               bool valid)
{
    uint32_t pk[3];
    int cv[3];
#define PVF_CH(K)                                                                                                                         \
    {                                                                                                                                     \
        constexpr int IU = 4 + 3 * P + K, IL = 1 + 3 * P + K, IR = 7 + 3 * P + K;                                                         \
        const uint32_t cx = ((ce[IR >> 2] >> (8 * (IR & 3))) & 0xffu) - ((ce[IL >> 2] >> (8 * (IL & 3))) & 0xffu);                          \
        pk[K] = sub_into_upper<(IU & 3), (IU & 3)>(cx, dn[IU >> 2], up[IU >> 2]);                                                          \
        const s16x2 v = __builtin_bit_cast(s16x2, pk[K]);                                                                                 \
        cv[K] = __builtin_amdgcn_sdot2(v, v, 0, false);                                                                                   \
    }
    PVF_CH(0) PVF_CH(1) PVF_CH(2)
#undef PVF_CH
    const int bv = max(cv[0], max(cv[1], cv[2]));
    const uint32_t pb = (cv[0] == bv) ? pk[0] : ((cv[1] == bv) ? pk[1] : pk[2]);         // first channel with the largest |g|^2
    // table offset: X = cx mod 512 (bits 0-8 of pb), Y = cy mod 512 (bits 16-24): (X & 7) | Y << 3 | (X >> 3) << 12
    const uint32_t off = (pb & 7u) | ((pb >> 13) & 0xff8u) | ((pb << 9) & 0x3f000u);
    u32x2 r;
    r.y = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(lut_rs, off, 0, 0);
    const float m = sqrt_exact_small((float)bv);
    r.x = valid ? __float_as_uint(m) : 0u;
    return r;
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
fhog_split_ml_k(MlStarts st, const LvDesc* __restrict__ lv, int B, const uint8_t* __restrict__ img_base, float* __restrict__ feat_base,
                const uint8_t* __restrict__ lutw, int oy, int ox)
{
    constexpr int RSRC_FLAGS = 0x00020000;
    extern __shared__ __attribute__((aligned(16))) uint32_t fs_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 3, role = wave >> 2;
    // rings (the column lane 63 reads beyond its strip must hold zeros: a bin offset from there is an LDS address) and flags
    for (int i = threadIdx.x; i < FS_PAIRS * (2 * FS_SLOT_DW + FS_FLAG_DW); i += 512) fs_lds[FS_PAIRS * FS_BINS_DW + i] = 0u;
    __syncthreads();
    const int g = ml_block(st);
    if (g >= st.b0[st.nl]) return;
    const int l = ml_level(st, g);
    const LvDesc d = lv[l];
    const int task = (g - st.b0[l]) * FS_PAIRS + pair;
    if (task >= d.fused_tasks) return;                             // wave-uniform, the same for both waves of a strip
    const int sx = task % d.strips;
    const int t2 = task / d.strips;
    const int cy = t2 % d.chunks;
    const int b = t2 / d.chunks;
    const int y0 = cy * d.chunk_rows;                              // first feature (hog) row of this chunk
    const int R = min(d.chunk_rows, d.hog_nr - y0);
    const int g_first = y0 + 1, g_last = y0 + R + 3;               // bands (= cell rows whose upper half they hold)
    const int y_begin = 8 * g_first - 12;
    // byte addresses in LDS: this lane's column of the strip's ring (slot s: + s * 4 * FS_SLOT_DW) and its word of the flags
    const uint32_t ring_at = lds_addr(fs_lds + FS_PAIRS * FS_BINS_DW + pair * 2 * FS_SLOT_DW) + 8u * lane;
    uint32_t* const flags = fs_lds + FS_PAIRS * (FS_BINS_DW + 2 * FS_SLOT_DW) + pair * FS_FLAG_DW + lane;      // ready: [64 * s], done: [128 + 64 * s]
    const uint32_t flag_at = lds_addr(flags);

    if (role == 0) {
        // ---------------- gradient wave ----------------
        const int hx = FUSED_OUT * sx + 1 + lane;                  // histogram column of this lane
        const int x_first = 8 * hx - 12;                           // image column of its first pixel
        const uint8_t* im = img_base + d.img_off + (size_t)b * d.img_stride;
        const int rb = d.rb;
        unsigned xmask = 0;                                        // gradients exist for 1 <= x < visible_nc (oracle/pvo_fhog.c)
#pragma unroll
        for (int p = 0; p < 8; ++p) if (x_first + p >= 1 && x_first + p < d.visible_nc) xmask |= 1u << p;
        const bool strip_inside = __builtin_amdgcn_ballot_w64(xmask != 0xffu) == 0;      // wave-uniform
        // byte offsets of the two 16-byte halves of the lane's 32-byte window in an image row (8-aligned): bytes [3 x_first - 4, 3 x_first + 28),
        // pixel p, channel k at byte 4 + 3 p + k.  Columns left of the image give negative offsets, which are huge as unsigned: out of range
        // for the row's buffer descriptor => zeros (such pixels are never valid); a row outside the image gets an empty descriptor.
        const int voff = 3 * x_first - 4, voff2 = voff + 16;
        const __amdgpu_buffer_rsrc_t lut_rs = __builtin_amdgcn_make_buffer_rsrc((void*)lutw, 0, 1 << 18, RSRC_FLAGS);
        // Image rows as 8 dwords per lane: the three around the row whose gradients are formed and one on its way.  (Eight buffers, rows
        // requested six steps ahead, measured the same 7.8 ms per 125 frames: the rows' latency is not what this wave waits for.)
        uint32_t rw[4][8];
        auto load_row = [&](int yi, uint32_t* dst) {
            const int bytes = (yi >= 0 && yi < d.h) ? rb : 0;       // wave-uniform
            const int yc = min(max(yi, 0), d.h - 1);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(im + (size_t)yc * rb), 0, bytes, RSRC_FLAGS);
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, voff2, 0, 0);
            dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = c.x; dst[5] = c.y; dst[6] = c.z; dst[7] = c.w;
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) load_row(y_begin - 1 + q, rw[q]);
        const int nrows = 8 * (g_last - g_first + 1);
        for (int k0 = 0; k0 < nrows; k0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + i, y = y_begin + k;
                // rows y - 1, y, y + 1 in rw[i], rw[i + 1], rw[i + 2] (mod 4); rw[i + 3] holds row y + 2, requested two steps ago
                const uint32_t* up = rw[i & 3];
                const uint32_t* ce = rw[(i + 1) & 3];
                const uint32_t* dn = rw[(i + 2) & 3];
                // asked for early: the slot's previous row (k - 2) must have been read before the slot is written again
                const int freed = (int)__hip_atomic_load(flags + 128 + 64 * (i & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool ok = (y >= 1 && y < d.visible_nr);
                u32x2 mb[8];
                if (strip_inside && ok) {
                    // every pixel of the row has a gradient (a strip that touches neither side of the image, a row inside it: most of them): no select per pixel
                    mb[0] = split_grad_px<0>(up, ce, dn, lut_rs, true); mb[1] = split_grad_px<1>(up, ce, dn, lut_rs, true);
                    mb[2] = split_grad_px<2>(up, ce, dn, lut_rs, true); mb[3] = split_grad_px<3>(up, ce, dn, lut_rs, true);
                    mb[4] = split_grad_px<4>(up, ce, dn, lut_rs, true); mb[5] = split_grad_px<5>(up, ce, dn, lut_rs, true);
                    mb[6] = split_grad_px<6>(up, ce, dn, lut_rs, true); mb[7] = split_grad_px<7>(up, ce, dn, lut_rs, true);
                } else {
                    mb[0] = split_grad_px<0>(up, ce, dn, lut_rs, ok && (xmask & 1u));
                    mb[1] = split_grad_px<1>(up, ce, dn, lut_rs, ok && (xmask & 2u));
                    mb[2] = split_grad_px<2>(up, ce, dn, lut_rs, ok && (xmask & 4u));
                    mb[3] = split_grad_px<3>(up, ce, dn, lut_rs, ok && (xmask & 8u));
                    mb[4] = split_grad_px<4>(up, ce, dn, lut_rs, ok && (xmask & 16u));
                    mb[5] = split_grad_px<5>(up, ce, dn, lut_rs, ok && (xmask & 32u));
                    mb[6] = split_grad_px<6>(up, ce, dn, lut_rs, ok && (xmask & 64u));
                    mb[7] = split_grad_px<7>(up, ce, dn, lut_rs, ok && (xmask & 128u));
                }
                load_row(y + 3, rw[i & 3]);                          // row y - 1 is done with
                int fr = freed;
                while (fr < k - 1) {
                    __builtin_amdgcn_s_sleep(1);
                    fr = (int)__hip_atomic_load(flags + 128 + 64 * (i & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ring_write(ring_at + (i & 1) * 4 * FS_SLOT_DW, flag_at + 256 * (i & 1), mb, (uint32_t)(k + 1));
            }
        }
        return;
    }

    // ---------------- vote wave ----------------
    float* accE = reinterpret_cast<float*>(fs_lds) + pair * FS_BINS_DW + lane;      // bin k of this lane: accE[64 * k]
    float* accO = accE + 18 * 64;
#pragma unroll
    for (int k = 0; k < 18; ++k) { accE[64 * k] = 0.0f; accO[64 * k] = 0.0f; }
    float hprev[18];
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int k = 0; k < 18; ++k) hprev[k] = 0.f;
    int krow = 0;                                                  // rows of this task read so far
    // One band = 8 pixel rows: the upper half of cell row gb (bins in accU) and the lower half of cell row gb - 1 (bins in accL).
    // The first band's lower half (cell row y0) and the last band's upper half (cell row y0 + R + 3) belong to cells this chunk
    // does not need; they are accumulated all the same (no branches in the vote loop): the first is read and dropped, the last
    // is never read.  Rows without gradients vote with magnitude 0 (x + 0 = x: nothing changes).
    auto band = [&](int gb, float* accU, float* accL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the row's pairs: own 8 pixels and the right neighbour's 8
            u32x2 own[8], nb[8];
            for (;;) {
                const int ready = (int)ring_read(ring_at + (i & 1) * 4 * FS_SLOT_DW, flag_at + 256 * (i & 1), own, nb);
                if (__builtin_amdgcn_readfirstlane(ready) > krow) break;
                __builtin_amdgcn_s_sleep(1);
            }
            ++krow;
            flag_write(flag_at + 512 + 256 * (i & 1), (uint32_t)krow);
            const float fy = ((float)i + 0.5f) / 8.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int p = j & 7;
                const u32x2 v = (j < 8) ? own[p] : nb[p];
                const float mv = __uint_as_float(v.x);
                const int bv = (int)(v.y << 8);                    // BYTE offset of the bin's row of 64 lanes
                const float fx = ((float)p + 0.5f) / 8.0f;
                const float wx = (j < 8) ? fx : 1.0f - fx;
                float* const pl = reinterpret_cast<float*>(reinterpret_cast<char*>(accL) + bv);
                float* const pu = reinterpret_cast<float*>(reinterpret_cast<char*>(accU) + bv);
                const float vl = *pl, vu = *pu;
                *pl = vl + ((1.0f - fy) * wx) * mv;
                *pu = vu + (fy * wx) * mv;
            }
        }
        // cell row c = gb - 1 is complete (for c = y0: a half-filled cell that only has to be cleared; its energy is shifted out of
        // e0..e2 before the first feature row is formed): energy in the oracle's order, straight from LDS
        const int c = gb - 1;
        float e = 0.0f;
#pragma unroll
        for (int o = 0; o < 9; ++o) { const float s2 = accL[64 * o] + accL[64 * (o + 9)]; e = e + s2 * s2; }
        e2 = e1; e1 = e0; e0 = e;
        if (c >= y0 + 3) {
            // features of the centre cell row c - 1 (histograms in hprev), hog row yh = c - 3; norms: rows c-2, c-1, c x lanes L-1, L, L+1
            float n[9];
            n[1] = e2; n[4] = e1; n[7] = e0;
            n[0] = __uint_as_float(from_prev_lane(__float_as_uint(e2))); n[2] = __uint_as_float(from_next_lane(__float_as_uint(e2)));
            n[3] = __uint_as_float(from_prev_lane(__float_as_uint(e1))); n[5] = __uint_as_float(from_next_lane(__float_as_uint(e1)));
            n[6] = __uint_as_float(from_prev_lane(__float_as_uint(e0))); n[8] = __uint_as_float(from_next_lane(__float_as_uint(e0)));
            const int x = FUSED_OUT * sx + lane - 1, yh = c - 3;
            if (lane >= 1 && lane <= FUSED_OUT && x < d.hog_nc) {
                // fhog_dev.h: cell_features, statement for statement, but every plane group leaves as soon as its four values exist and the
                // scheduler may not pull later groups' arithmetic in front of it (sched_barrier): with all 32 outputs alive at once the
                // kernel does not fit its 128 registers, the allocator spills around the stores, and a reload (scratch_load -> vmcnt(0))
                // waits for every store issued before it.  Plane group k of the strip's 61 cells is one run of memory (detect_ml.h:
                // feat_at): every store writes whole lines.
                float4* dst = reinterpret_cast<float4*>(feat_base + d.feat_off + (size_t)b * d.feat_stride + feat_at(yh + oy, 0, x + ox, d.fwp));
                const size_t step = (size_t)d.fwp;
                const float eps = 0.0001f;
                const float z1[4] = {n[4], n[1], n[3], n[0]}, z2[4] = {n[5], n[2], n[4], n[1]}, z3[4] = {n[7], n[4], n[6], n[3]}, z4[4] = {n[8], n[5], n[7], n[4]};
                float nn[4], nv[4], t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nn[k] = 0.2f * sqrtf((((z1[k] + z2[k]) + z3[k]) + z4[k]) + eps);
                    nv[k] = 0.1f / nn[k];
                }
                float o[20];                                          // the contrast-sensitive planes 0 .. 17, then 18, 19 of the next part
#pragma unroll
                for (int g = 0; g < 18; g += 3) {
                    float hh[3][4];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) hh[j][k] = fminf(hprev[g + j], nn[k]) * nv[k];
                        o[g + j] = (hh[j][0] + hh[j][1]) + (hh[j][2] + hh[j][3]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = t[k] + ((hh[0][k] + hh[1][k]) + hh[2][k]);
                    // planes 0..3 are complete after g = 3, 4..7 after g = 6, 8..11 after g = 9, 12..15 after g = 15
                    if (g == 3) { *dst = make_float4(o[0], o[1], o[2], o[3]); dst += step; __builtin_amdgcn_sched_barrier(0); }
                    if (g == 6) { *dst = make_float4(o[4], o[5], o[6], o[7]); dst += step; __builtin_amdgcn_sched_barrier(0); }
                    if (g == 9) { *dst = make_float4(o[8], o[9], o[10], o[11]); dst += step; __builtin_amdgcn_sched_barrier(0); }
                    if (g == 15) { *dst = make_float4(o[12], o[13], o[14], o[15]); dst += step; __builtin_amdgcn_sched_barrier(0); }
                }
                float ci[9];                                          // the contrast-insensitive planes 18 .. 26
#pragma unroll
                for (int g = 0; g < 9; ++g) {
                    const float s2 = hprev[g] + hprev[g + 9];
                    float hh[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) hh[k] = fminf(s2, nn[k]) * nv[k];
                    ci[g] = (hh[0] + hh[1]) + (hh[2] + hh[3]);
                    if (g == 1) { *dst = make_float4(o[16], o[17], ci[0], ci[1]); dst += step; __builtin_amdgcn_sched_barrier(0); }
                    if (g == 5) { *dst = make_float4(ci[2], ci[3], ci[4], ci[5]); dst += step; __builtin_amdgcn_sched_barrier(0); }
                }
                const float tscale = (float)(2 * 0.2357);
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = t[k] * tscale;
                *dst = make_float4(ci[6], ci[7], ci[8], t[0]); dst += step;
                *dst = make_float4(t[1], t[2], t[3], 0.0f);
            }
        }
        // the finished cell row becomes the centre of the next feature row: its bins move to registers, the LDS words are cleared
#pragma unroll
        for (int k = 0; k < 18; ++k) { hprev[k] = accL[64 * k]; accL[64 * k] = 0.0f; }
    };
    for (int gb = g_first; gb <= g_last; ++gb) {
        if (gb & 1) band(gb, accO, accE);                           // upper half -> the odd cell row gb, lower half -> the even row gb - 1
        else band(gb, accE, accO);
    }
}

// writes the zero border of the feature maps (the padding ring around the hog cells: (frows-1)/2 cells above / left, the rest below / right).
// Needed once per (buffer, plan): the fused kernel only ever writes hog cells, so the ring stays valid across batches.
__global__ void __launch_bounds__(256) feat_ring_zero_k(MlStarts st, const LvDesc* __restrict__ lv, int B, float* __restrict__ feat_base, int oy, int ox)
{
    const int g = ml_block(st);
    if (g >= st.b0[st.nl]) return;
    const int l = ml_level(st, g);
    const LvDesc d = lv[l];
    const int local = g - st.b0[l];
    const int xb = local % d.feat_bx;
    const int py = (local / d.feat_bx) % d.fh;
    const int b = local / (d.feat_bx * d.fh);
    const int px = xb * 256 + threadIdx.x;
    if (px >= d.fwp) return;                                       // (the zero columns behind the map's last one included)
    const int x = px - ox, y = py - oy;
    if (x >= 0 && y >= 0 && x < d.hog_nc && y < d.hog_nr) return;
    float4* dst = reinterpret_cast<float4*>(feat_base + d.feat_off + (size_t)b * d.feat_stride + feat_at(py, 0, px, d.fwp));
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[(size_t)k * d.fwp] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------------
// K3: the five filters on every window position of every level, on the fp32 matrix cores.  N = 5 filters would fill 5 of 16 MFMA columns,
// so three neighbouring output columns share one 16-column tile (column = 5 * shift + filter; K grows from 10 to 12 cells per filter row;
// the zero entries of B are exact no-ops in the fmaf chain); K of a filter row is the packed run of 12 cells x 31 planes (the feature map's
// pad plane is dropped in the slab) = 93 k-steps of 4; B fragments come packed four k-steps per lane ([m][cell column][2][64 lanes][4]) and
// are fetched one cell column ahead; every accumulator receives its terms in (m, n, p) order  =>  bit-identical to the oracle's chain.
// A block of TWO waves owns one column strip (96 output columns) of one level of one frame and walks it top to bottom, one feature row
// per step, staged ONCE into a double-buffered slab both waves read (each fetches half of row s + 1 while row s is multiplied; one
// barrier per step).  At step s the ten output rows s - 9 .. s are alive (row r takes filter row m = s - r); wave w owns the rows
// r = w (mod 2), i.e. five of them at every step -- the two waves always have the same work, whatever the phase.  A wave's slot q holds
// the row with m = p + 2 q (p = parity of s - w): on odd steps slot 4 completes (m = 9), is written out, and the slots move up by one
// (40 register moves per 1860 MFMAs).  A feature row leaves HBM / L2 once per strip and feeds five MFMAs per fragment read.  (Round 2's
// form -- one wave per 4 output rows x 96 columns with a slab of its own -- staged every row 3.25 times: 3.8 x the feature maps from
// beyond L2, now 1.09 x; a four-wave form with three rows per wave and one shared slab was 12 % slower: its waves hold 1 to 3 rows' worth
// of work per step and the barrier waits for the longest.  DESIGN.md section 3.)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
score_roll_k(MlStarts st, const LvDesc* __restrict__ lv, int B, const float* __restrict__ feat_base,
             const float4* __restrict__ Bg4, ScoreParams sp, int* __restrict__ counts, CandRec* __restrict__ cands)
{
    constexpr int FR = 10, FC = 10, NK = 12, PITCH = 31, MT = 2, WCOLS = MT * 48, SEG = WCOLS + 11, R = 5;
    constexpr int LAST_STEPS = 93 - 8 * (NK - 1);
    constexpr int NSTW = (SEG * 8 + 127) / 128;     // 16-byte pieces of a row segment per thread (7)
    constexpr int SLAB = (SEG * PITCH + 3) / 4 * 4; // floats per slab buffer
    constexpr int RSRC_FLAGS = 0x00020000;
    extern __shared__ __attribute__((aligned(16))) float s_seg[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = (int)blockIdx.x;
    if (g >= st.b0[st.nl]) return;
    const int l = ml_level(st, g);
    const LvDesc d = lv[l];
    const int local = g - st.b0[l];
    const int bx = local % d.score_bx;
    const int sg = (local / d.score_bx) % d.roll_nseg;
    const int b = __builtin_amdgcn_readfirstlane(local / (d.score_bx * d.roll_nseg));
    const int fw = d.fw;
    const int c_base = bx * WCOLS;
    const int c1 = fw - (FC - FC / 2 - 1);
    // this block's piece of the strip: output rows r_base .. r_base + out_rows - 1 (by their top feature row), i.e. the feature rows
    // r_base .. r_base + fh - 1; below, rows and steps are counted from r_base
    const int r_base = sg * d.roll_rows;
    const int out_rows = min(d.roll_rows, d.fh - (FR - 1) - r_base);
    const int fh = out_rows + FR - 1;
    const int fwp = d.fwp;
    const float* fb = feat_base + d.feat_off + (size_t)b * d.feat_stride + feat_at(r_base, 0, c_base, fwp);
    const int seg_cells = (fw - c_base < SEG) ? fw - c_base : SEG;
    const int i = lane & 15, kq = lane >> 4;
    const int lane16 = lane * 16;
    const bool two_tiles = (c_base + 48 + FC / 2 < c1);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)Bg4, 0, FR * NK * 2 * 64 * 16, RSRC_FLAGS);
    f32x4 acc[R][MT];
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int tt = 0; tt < MT; ++tt) acc[q][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this thread's share of a feature row: cell `tid` of the segment (threads 107 .. 127: none), its eight 16-byte pieces -- one per
    // plane group, each group a run of memory of its own (detect_ml.h: feat_at) -- fetched and parked in two halves so that only four
    // pieces are held in registers at a time.  The range of a plane group's descriptor is the segment's cells: cells past the segment (or
    // any cell of a row past the piece's last) come back as zeros.
    constexpr int NH = 4;
    static_assert(NSTW <= 8 && SEG <= 128, "one cell per thread");
    u32x4 sv[NH];
    const int tid16 = (int)threadIdx.x * 16;
    auto load_part = [&](int fr, int half) {
        const int bytes = (fr < fh) ? seg_cells * 16 : 0;
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            // a descriptor per plane group (its base is scalar arithmetic): a scalar OFFSET would count against the range
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(fb + ((size_t)fr * 8 + half * NH + u) * fwp * 4), 0, bytes, RSRC_FLAGS);
            sv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid16, 0, 0);
        }
    };
    auto fill_part = [&](float* seg, int half) {
        if ((int)threadIdx.x < SEG) {
#pragma unroll
            for (int u = 0; u < NH; ++u) {
                const int q = half * NH + u;
                uint32_t* dd = reinterpret_cast<uint32_t*>(seg + (int)threadIdx.x * PITCH + 4 * q);
                dd[0] = sv[u].x; dd[1] = sv[u].y; dd[2] = sv[u].z;
                if (q != 7) dd[3] = sv[u].w;            // (plane 31 is padding; its slot belongs to the next cell)
            }
        }
    };
    auto emit = [&](int r_out, const f32x4 (&a)[MT]) {
        const int jc = lane & 15;
        if (jc < 15) {
            const int sft = jc / 5, f = jc % 5;
            const float th = sp.thresh[f];
            const int r = r_base + r_out + FR / 2;
#pragma unroll
            for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int pos = 4 * (lane >> 4) + reg;
                    const int cc = c_base + tt * 48 + 3 * pos + sft + FC / 2;
                    const float v = a[tt][reg];
                    if (cc < c1 && v >= th && (tt == 0 || two_tiles)) {
                        const int idx = atomicAdd(&counts[b], 1);
                        if (idx < sp.cap) {
                            CandRec rec;
                            rec.score = v - th; rec.filter = f; rec.level = l; rec.r = r; rec.c = cc;
                            cands[(size_t)b * sp.cap + idx] = rec;
                        }
                    }
                }
        }
    };
    load_part(0, 0); fill_part(s_seg, 0);
    load_part(0, 1); fill_part(s_seg, 1);
    __syncthreads();
    for (int s = 0; s < fh; ++s) {
        const float* seg = s_seg + (s & 1) * SLAB;
        float* nxt = s_seg + ((s + 1) & 1) * SLAB;
        load_part(s + 1, 0);                        // (past the last row: zeros, never used)
        const int p = (s - wave) & 1;
        const int r0 = s - p;                       // slot q holds output row r0 - 2 q, which takes filter row p + 2 q at this step
        bool on[R];
        bool all_on = two_tiles, any_on = false;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = r0 - 2 * q;
            on[q] = (r >= 0 && r < out_rows);
            all_on = all_on && on[q];
            any_on = any_on || on[q];
        }
        if (any_on) {
            const int bop = p * NK * 2048;
            const float* a0 = seg + (3 * i) * PITCH + kq;
            u32x4 bn[R][2];
            float an[8 * MT];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                bn[q][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bop + 2 * q * NK * 2048, 0);
                bn[q][1] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bop + 2 * q * NK * 2048 + 1024, 0);
            }
#pragma unroll
            for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) an[pq * MT + tt] = a0[(tt * 48) * PITCH + 4 * pq];
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                float ac[8 * MT];
                float bc[R][8];
#pragma unroll
                for (int x = 0; x < 8 * MT; ++x) ac[x] = an[x];
#pragma unroll
                for (int q = 0; q < R; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t b0 = bn[q][h].x, b1 = bn[q][h].y, b2 = bn[q][h].z, b3 = bn[q][h].w;
                        bc[q][4 * h] = __uint_as_float(b0); bc[q][4 * h + 1] = __uint_as_float(b1);
                        bc[q][4 * h + 2] = __uint_as_float(b2); bc[q][4 * h + 3] = __uint_as_float(b3);
                    }
                if (n + 1 < NK) {
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        bn[q][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bop + (2 * q * NK + n + 1) * 2048, 0);
                        bn[q][1] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bop + (2 * q * NK + n + 1) * 2048 + 1024, 0);
                    }
#pragma unroll
                    for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                        for (int tt = 0; tt < MT; ++tt) an[pq * MT + tt] = a0[(tt * 48) * PITCH + 32 * (n + 1) + 4 * pq];
                }
                if (n == NK / 2) { fill_part(nxt, 0); load_part(s + 1, 1); }     // (the slab being filled is not the one being read)
                __builtin_amdgcn_sched_barrier(0);
                const int steps = (n == NK - 1) ? LAST_STEPS : 8;      // (n is an unrolled constant)
                if (all_on) {
#pragma unroll
                    for (int pq = 0; pq < steps; ++pq)
#pragma unroll
                        for (int q = 0; q < R; ++q)
#pragma unroll
                            for (int tt = 0; tt < MT; ++tt)
                                acc[q][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT + tt], bc[q][pq], acc[q][tt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        if (!on[q]) continue;
                        if (two_tiles) {
#pragma unroll
                            for (int pq = 0; pq < steps; ++pq)
#pragma unroll
                                for (int tt = 0; tt < MT; ++tt)
                                    acc[q][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT + tt], bc[q][pq], acc[q][tt], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int pq = 0; pq < steps; ++pq)
                                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT], bc[q][pq], acc[q][0], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        else { fill_part(nxt, 0); load_part(s + 1, 1); }
        if (p == 1) {
            // slot 4 has taken its last filter row (m = 9): write it out, move the slots up, slot 0 starts the row s + 1
            if (on[R - 1]) emit(r0 - 2 * (R - 1), acc[R - 1]);
#pragma unroll
            for (int q = R - 1; q > 0; --q)
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) acc[q][tt] = acc[q - 1][tt];
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) acc[0][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        fill_part(nxt, 1);
        __syncthreads();
    }
}

struct MlPlan {
    int h = 0, w = 0, upsample = -1, B = 0;
    std::vector<LvDesc> lv;
    std::vector<LevelDims> ups;
    MlStarts feat, fused, walk;
    int feat_blocks = 0, fused_blocks = 0, walk_blocks = 0;
    size_t img_bytes = 0, feat_floats = 0, up_bytes = 0;
    LvDesc* d_lv = nullptr;
    RowTab* d_rowtab = nullptr;                // row tables of every resize stage: upsampling stages first, then level l from level l - 1
    ColTab* d_coltab = nullptr;                // column tables, the same stages
    std::vector<size_t> up_tab, lv_tab;        // offsets (entries) into d_rowtab
    std::vector<size_t> up_ctab, lv_ctab;      // ... into d_coltab
    const void* ring_valid_for = nullptr;      // feature buffer whose zero border was written for this plan (fused FHOG)
    ScreenPlan screen;                         // work items of the screening pass (screen.hip), built when it is first used
    bool screen_built = false;
    ~MlPlan() { if (d_lv) (void)hipFree(d_lv); if (d_rowtab) (void)hipFree(d_rowtab); if (d_coltab) (void)hipFree(d_coltab); }
    MlPlan() = default;
    MlPlan(const MlPlan&) = delete;
    MlPlan& operator=(const MlPlan&) = delete;
};

// plans live in their context (one per frame size / upsampling / batch size) and die with it
struct MlPlanCache {
    std::map<std::vector<int>, std::unique_ptr<MlPlan>> plans;
    int dpp_probe = -1;                        // 1: v_mov_b32_dpp wave_shl/wave_shr move data as the fused kernel expects
    int sqrt_probe = -1;                       // 1: sqrt_exact_small is correctly rounded on 0 .. 2 * 255^2
};
void ml_plans_free(Ctx* c)
{
    delete c->ml_plans;
    c->ml_plans = nullptr;
}

static MlPlan* ml_plan(Ctx* c, int h, int w, int upsample, int B)
{
    if (!c->ml_plans) c->ml_plans = new MlPlanCache();
    const std::vector<int> key{h, w, upsample, B};
    auto it = c->ml_plans->plans.find(key);
    if (it != c->ml_plans->plans.end()) return it->second.get();
    const DetectorModel& m = c->det;
    std::unique_ptr<MlPlan> pp(new MlPlan());
    MlPlan& p = *pp;
    p.h = h; p.w = w; p.upsample = upsample; p.B = B;
    std::vector<LevelDims> dims = level_schedule(h, w, upsample, m, &p.ups);
    PVF_REQUIRE((int)dims.size() <= ML_MAX, "too many pyramid levels");
    auto al = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    for (size_t u = 0; u + 1 < p.ups.size(); ++u) p.up_bytes = std::max(p.up_bytes, (size_t)p.ups[u].h * al((size_t)p.ups[u].w * 3, 64) * B);
    p.feat.nl = p.fused.nl = p.walk.nl = (int)dims.size();
    // Feature rows per FHOG task.  A task re-walks 3 cell rows above its first feature row, so tasks are as tall as the batch allows: the
    // tallest that still leave two tasks per strip slot of the device (n_cu x 2 blocks x 4 strips).  Measured on 125 1080p frames, detector
    // alone, ms per batch: 16 rows 7.39, 32 rows 6.84, 64 rows 6.59, 96 rows 6.54, 160 rows 6.50, whole levels 6.52
    // (profiles/r06_fhog_split_experiment.txt).
    int chunk_big = 8;
    for (int cand : {160, 128, 96, 64, 48, 32, 24, 16}) {
        long long tasks = 0;
        for (const LevelDims& ld : dims) {
            const int hog_nr = (int)((double)ld.h / 8.0 + 0.5) - 2, hog_nc = (int)((double)ld.w / 8.0 + 0.5) - 2;
            if (hog_nr > 0 && hog_nc > 0) tasks += (long long)((hog_nc + FUSED_OUT - 1) / FUSED_OUT) * std::max(2, (hog_nr + cand - 1) / cand) * B;
        }
        if (tasks >= 2LL * c->n_cu * 2 * 4) { chunk_big = cand; break; }
    }
    if (getenv("PVF_FHOG_CHUNK") && atoi(getenv("PVF_FHOG_CHUNK")) > 0) chunk_big = atoi(getenv("PVF_FHOG_CHUNK"));
    for (size_t l = 0; l < dims.size(); ++l) {
        LvDesc d;
        memset(&d, 0, sizeof d);
        d.h = dims[l].h; d.w = dims[l].w;
        d.cells_nr = (int)((double)d.h / 8.0 + 0.5); d.cells_nc = (int)((double)d.w / 8.0 + 0.5);
        d.visible_nr = std::min(d.cells_nr * 8, d.h) - 1; d.visible_nc = std::min(d.cells_nc * 8, d.w) - 1;
        d.hog_nr = d.cells_nr - 2; d.hog_nc = d.cells_nc - 2;
        const bool feat_ok = d.hog_nr > 0 && d.hog_nc > 0;
        d.fh = feat_ok ? d.hog_nr + m.frows - 1 : 0; d.fw = feat_ok ? d.hog_nc + m.fcols - 1 : 0;
        d.valid_score = (d.fh >= m.frows && d.fw >= m.fcols) ? 1 : 0;
        d.rb = (int)al((size_t)d.w * 3, 64);
        d.img_off = (long long)p.img_bytes; d.img_stride = (long long)d.h * d.rb;
        p.img_bytes += (size_t)d.img_stride * B;
        d.fwp = feat_ok ? d.fw + FEAT_PAD_COLS : 0;
        d.feat_off = (long long)p.feat_floats; d.feat_stride = (long long)d.fh * d.fwp * PVF_FHOG_STRIDE;
        p.feat_floats += (size_t)d.feat_stride * B;
        d.feat_bx = feat_ok ? (d.fwp + 255) / 256 : 0;
        const int out_c = d.fw - 9;
        d.score_bx = d.valid_score ? (out_c + 95) / 96 : 0;        // column strips of 96 output columns (K3)
        // fused FHOG tasks: strips of 61 feature columns x chunks of feature rows (smaller chunks for the small levels: more tasks)
        d.strips = feat_ok ? (d.hog_nc + FUSED_OUT - 1) / FUSED_OUT : 0;
        {
            // chunks of about chunk_big feature rows, all of a level's the same height (a level lower than two chunks: two halves)
            const int nch = std::max(2, (d.hog_nr + chunk_big - 1) / chunk_big);
            d.chunk_rows = std::max((d.hog_nr + nch - 1) / nch, 1);
        }
        d.chunks = feat_ok ? (d.hog_nr + d.chunk_rows - 1) / d.chunk_rows : 0;
        d.fused_tasks = d.strips * d.chunks * B;
        p.feat.b0[l] = p.feat_blocks; p.feat_blocks += d.feat_bx * d.fh * B;
        p.fused.b0[l] = p.fused_blocks; p.fused_blocks += (d.fused_tasks + 3) / 4;
        p.lv.push_back(d);
    }
    const int nl = (int)dims.size();
    {
        // K3 v5: a block per column strip and piece of <= roll_max output rows.  A piece re-reads the 9 feature rows above it and starts
        // with 9 steps of partial work, so pieces are as tall as the launch allows: no block longer than about half of what a block slot
        // (4 per CU) works through in the whole launch -- 1080p x 128 frames: whole strips (270 rows); 4K x 32 frames: two pieces per strip
        // on the largest level (measured: 64-row pieces cost 3 % at 1080p, whole 540-row strips 4 % at 4K)
        long long total = 0;
        for (const LvDesc& d : p.lv) if (d.valid_score) total += (long long)d.score_bx * (d.fh - 9 + 9) * B;
        const int slots = 4 * 256;
        int roll_max = (int)std::max<long long>(64, total / slots / 2);
        if (getenv("PVF_SCORE_SEG")) roll_max = std::max(2, atoi(getenv("PVF_SCORE_SEG")));
        for (int l = 0; l < nl; ++l) {
            LvDesc& d = p.lv[l];
            const int out_r = d.fh - 9;
            d.roll_nseg = d.valid_score ? (out_r + roll_max - 1) / roll_max : 0;
            d.roll_rows = d.valid_score ? ((out_r + d.roll_nseg - 1) / d.roll_nseg + 1) / 2 * 2 : 0;
            if (d.valid_score) d.roll_nseg = (out_r + d.roll_rows - 1) / d.roll_rows;      // (rounding the height up may have emptied the last piece)
            p.walk.b0[l] = p.walk_blocks; p.walk_blocks += d.score_bx * d.roll_nseg * B;
        }
    }
    p.feat.b0[nl] = p.feat_blocks; p.walk.b0[nl] = p.walk_blocks;
    p.fused.b0[nl] = p.fused_blocks;
    HIP_CHECK(hipMalloc((void**)&p.d_lv, sizeof(LvDesc) * nl));
    HIP_CHECK(hipMemcpy(p.d_lv, p.lv.data(), sizeof(LvDesc) * nl, hipMemcpyHostToDevice));
    {
        std::vector<RowTab> tab;
        int ch = h;
        for (size_t u = 0; u < p.ups.size(); ++u) { p.up_tab.push_back(tab.size()); fill_row_table(tab, ch, p.ups[u].h); ch = p.ups[u].h; }
        p.lv_tab.push_back(0);
        for (int l = 1; l < nl; ++l) { p.lv_tab.push_back(tab.size()); fill_row_table(tab, p.lv[l - 1].h, p.lv[l].h); }
        HIP_CHECK(hipMalloc((void**)&p.d_rowtab, sizeof(RowTab) * std::max<size_t>(tab.size(), 1)));
        if (!tab.empty()) HIP_CHECK(hipMemcpy(p.d_rowtab, tab.data(), sizeof(RowTab) * tab.size(), hipMemcpyHostToDevice));
        std::vector<ColTab> ctab;
        int cw2 = w;
        for (size_t u = 0; u < p.ups.size(); ++u) { p.up_ctab.push_back(ctab.size()); fill_col_table(ctab, cw2, p.ups[u].w); cw2 = p.ups[u].w; }
        p.lv_ctab.push_back(0);
        for (int l = 1; l < nl; ++l) { p.lv_ctab.push_back(ctab.size()); fill_col_table(ctab, p.lv[l - 1].w, p.lv[l].w); }
        HIP_CHECK(hipMalloc((void**)&p.d_coltab, sizeof(ColTab) * std::max<size_t>(ctab.size(), 1)));
        if (!ctab.empty()) HIP_CHECK(hipMemcpy(p.d_coltab, ctab.data(), sizeof(ColTab) * ctab.size(), hipMemcpyHostToDevice));
    }
    MlPlan* raw = pp.get();
    c->ml_plans->plans[key] = std::move(pp);
    return raw;
}

// builds the whole pyramid of the batch into s_pyr (level images at plan->lv[l].img_off); returns the plan
static MlPlan* ml_build_pyramid(Ctx* c, const std::vector<Frame>& frames, int upsample)
{
    const int B = (int)frames.size();
    const int h = frames[0].h, w = frames[0].w;
    for (auto& f : frames) PVF_REQUIRE(f.h == h && f.w == w, "batched frames must share one size");
    MlPlan* p = ml_plan(c, h, w, upsample, B);
    c->s_pyr.ensure(p->img_bytes + p->up_bytes + 256);
    uint8_t* base = c->s_pyr.as<uint8_t>();
    uint8_t* up_tmp = base + ((p->img_bytes + 63) / 64) * 64;
    const uint8_t** d_ptrs = nullptr;
    upload_frame_ptrs(c, frames, &d_ptrs);
    ProfScope ps(c, "pyramid", c->det_stream);
    auto al = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    const uint8_t* cur = nullptr;
    int ch = h, cw = w, crb = w * 3;
    size_t cstride = 0;
    for (size_t u = 0; u < p->ups.size(); ++u) {
        const bool last = (u + 1 == p->ups.size());
        uint8_t* dst = last ? base + p->lv[0].img_off : up_tmp;
        const int drb = (int)al((size_t)p->ups[u].w * 3, 64);
        const size_t dstride = (size_t)p->ups[u].h * drb;
        launch_resize_rows(c, cur ? nullptr : d_ptrs, cur, cstride, crb, ch, cw, dst, dstride, drb, p->ups[u].h, p->ups[u].w, B, p->d_rowtab + p->up_tab[u], p->d_coltab + p->up_ctab[u]);
        cur = dst; cstride = dstride; crb = drb; ch = p->ups[u].h; cw = p->ups[u].w;
    }
    if (!cur) {
        for (int b = 0; b < B; ++b)
            HIP_CHECK(hipMemcpy2DAsync(base + p->lv[0].img_off + (size_t)b * p->lv[0].img_stride, (size_t)p->lv[0].rb, frames[b].d, (size_t)w * 3,
                                       (size_t)w * 3, (size_t)h, hipMemcpyDeviceToDevice, c->det_stream));
    }
    for (size_t l = 1; l < p->lv.size(); ++l)
        launch_resize_rows(c, nullptr, base + p->lv[l - 1].img_off, (size_t)p->lv[l - 1].img_stride, p->lv[l - 1].rb, p->lv[l - 1].h, p->lv[l - 1].w,
                           base + p->lv[l].img_off, (size_t)p->lv[l].img_stride, p->lv[l].rb, p->lv[l].h, p->lv[l].w, B, p->d_rowtab + p->lv_tab[l], p->d_coltab + p->lv_ctab[l]);
    return p;
}

// What the fused FHOG kernel takes for granted about the device, checked once per context (a mismatch is an error, not a fallback):
// `v_mov_b32_dpp ... wave_shl:1 / wave_shr:1` with bound_ctrl hand lane i the value of lane i + 1 / i - 1 (lanes without a source: 0), and
// sqrt_exact_small (fhog_dev.h: the hardware estimate stepped UP when one exact residual asks for it) is the correctly rounded root of
// every squared gradient magnitude 0 .. 2 * 255^2.
__global__ void dpp_probe_k(int* out)
{
    const int lane = threadIdx.x;
    out[lane] = (int)from_next_lane((uint32_t)(lane + 100));
    out[64 + lane] = (int)from_prev_lane((uint32_t)(lane + 100));
}
#define SQRT_PROBE_N (2 * 255 * 255 + 1)
__global__ void sqrt_probe_k(float* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < SQRT_PROBE_N) out[i] = sqrt_exact_small((float)i);
}
static void check_dpp_direction(Ctx* c)
{
    MlPlanCache* pc = c->ml_plans;
    if (pc->dpp_probe < 0) {
        int* d = nullptr;                                // (a buffer of its own, once per context: every scratch buffer is somebody's state)
        HIP_CHECK(hipMalloc((void**)&d, 128 * sizeof(int) + SQRT_PROBE_N * sizeof(float)));
        float* ds = reinterpret_cast<float*>(d + 128);
        hipLaunchKernelGGL(dpp_probe_k, dim3(1), dim3(64), 0, c->det_stream, d);
        hipLaunchKernelGGL(sqrt_probe_k, dim3((SQRT_PROBE_N + 255) / 256), dim3(256), 0, c->det_stream, ds);
        int h[128];
        std::vector<float> hs(SQRT_PROBE_N);
        HIP_CHECK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->det_stream));
        HIP_CHECK(hipMemcpyAsync(hs.data(), ds, SQRT_PROBE_N * sizeof(float), hipMemcpyDeviceToHost, c->det_stream));
        HIP_CHECK(hipStreamSynchronize(c->det_stream));
        (void)hipFree(d);
        bool ok = true;
        for (int i = 0; i < 64; ++i) {
            ok = ok && h[i] == (i < 63 ? i + 101 : 0);
            ok = ok && h[64 + i] == (i > 0 ? i + 99 : 0);
        }
        pc->dpp_probe = ok ? 1 : 0;
        bool sq = true;
        for (int i = 0; i < SQRT_PROBE_N; ++i) sq = sq && hs[i] == (float)std::sqrt((double)i);      // (exact: double sqrt, one rounding)
        pc->sqrt_probe = sq ? 1 : 0;
    }
    PVF_REQUIRE(pc->dpp_probe == 1, "v_mov_b32_dpp wave_shl / wave_shr do not move data as the FHOG kernel expects on this device");
    PVF_REQUIRE(pc->sqrt_probe == 1, "v_sqrt_f32 is not within one step below the correctly rounded root on this device: the FHOG kernel's one-sided correction does not hold");
}

// pyramid + FHOG features of every level of the batch (s_feat at plan->lv[l].feat_off); returns the plan
static MlPlan* ml_features(Ctx* c, const std::vector<Frame>& frames, int upsample)
{
    const DetectorModel& m = c->det;
    const int B = (int)frames.size();
    MlPlan* p = ml_build_pyramid(c, frames, upsample);
    const int oy = (m.frows - 1) / 2, ox = (m.fcols - 1) / 2;
    const void* feat_before = c->s_feat.p;
    c->s_feat.ensure(p->feat_floats * sizeof(float) + 64);
    if (p->fused_blocks == 0) return p;
    check_dpp_direction(c);
    ProfScope ps(c, "fhog", c->det_stream);
    // the fused kernel writes hog cells only: the zero padding ring is written when this plan first meets this buffer (other plans
    // and the stage-access entries share s_feat, so the owner is tracked)
    if (c->s_feat.p != feat_before || c->feat_ring_owner != (const void*)p) {
        hipLaunchKernelGGL(feat_ring_zero_k, dim3(ml_grid(p->feat_blocks)), dim3(256), 0, c->det_stream, p->feat, p->d_lv, B, c->s_feat.as<float>(), oy, ox);
        c->feat_ring_owner = (const void*)p;
    }
    static std::atomic<uint64_t> attr_set{0};             // per device (a function attribute belongs to the device it was set on)
    const uint64_t bit = 1ull << (c->device & 63);
    if (!(attr_set.load() & bit)) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fhog_split_ml_k), hipFuncAttributeMaxDynamicSharedMemorySize, FS_LDS_BYTES));
        attr_set.fetch_or(bit);
    }
    hipLaunchKernelGGL(fhog_split_ml_k, dim3(ml_grid(p->fused_blocks)), dim3(512), FS_LDS_BYTES, c->det_stream, p->fused, p->d_lv, B,
                       c->s_pyr.as<uint8_t>(), c->s_feat.as<float>(), orientation_lut_wrapped(c), oy, ox);
    return p;
}

static void det_run_batch_ml(Ctx* c, const std::vector<Frame>& frames, int upsample, const ScoreParams& sp0, int* d_counts, CandRec* d_cands)
{
    const DetectorModel& m = c->det;
    const int B = (int)frames.size();
    MlPlan* p = ml_features(c, frames, upsample);
    if (p->walk_blocks == 0) return;
    if (c->det_screen && !c->det_screen_suspended) {
        // the f16 screening pass + the exact chain for what it lists (screen.hip); its counters sit behind the B candidate counts
        if (!p->screen_built) { screen_plan_build(p->screen, p->lv, B); p->screen_built = true; }
        if (p->screen.usable) {
            ProfScope ps(c, "score_screened", c->det_stream);
            screen_launch(c, p->screen, p->d_lv, B, c->s_feat.as<float>(), sp0, d_counts, d_cands, d_counts + B);
            ++c->screen_batches;
            return;
        }
    }
    ProfScope ps(c, "score", c->det_stream);
    const size_t lds = (size_t)2 * (((2 * 48 + 11) * 31 + 3) / 4 * 4) * sizeof(float);      // one slab of 107 packed cells of 31 planes, double-buffered
    const float4* b4 = reinterpret_cast<const float4*>(m.d_bmfma4);
    hipLaunchKernelGGL(score_roll_k, dim3(p->walk_blocks), dim3(128), lds, c->det_stream, p->walk, p->d_lv, B, c->s_feat.as<float>(), b4, sp0, d_counts, d_cands);
}

void det_pyramid_level(Ctx* c, const Frame& f, int upsample, int level, std::vector<uint8_t>* out, int* oh, int* ow)
{
    PVF_REQUIRE(c->det.loaded, "detector not loaded");
    std::vector<Frame> fr{f};
    MlPlan* p = ml_build_pyramid(c, fr, upsample);
    PVF_REQUIRE(level >= 0 && level < (int)p->lv.size(), "pyramid level out of range");
    *oh = p->lv[level].h; *ow = p->lv[level].w;
    if (out) {
        out->resize((size_t)(*oh) * (*ow) * 3);
        HIP_CHECK(hipMemcpy2DAsync(out->data(), (size_t)(*ow) * 3, c->s_pyr.as<uint8_t>() + p->lv[level].img_off, (size_t)p->lv[level].rb,
                                   (size_t)(*ow) * 3, (size_t)(*oh), hipMemcpyDeviceToHost, c->det_stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
}

// the resize chain of a whole batch and nothing else (measurement: tools/probes/coissue_probe.py runs it beside the embedder)
void det_pyramid_batch(Ctx* c, const std::vector<Frame>& frames, int upsample)
{
    PVF_REQUIRE(c->det.loaded, "detector not loaded");
    PVF_REQUIRE(!frames.empty(), "pyramid batch: no frames");
    (void)ml_build_pyramid(c, frames, upsample);
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
}

// features of one pyramid level as the batched detector computes them (parity tests of the multi-level FHOG kernels)
void det_level_features(Ctx* c, const Frame& f, int upsample, int level, std::vector<float>* out, int* fh, int* fw)
{
    PVF_REQUIRE(c->det.loaded, "detector not loaded");
    std::vector<Frame> fr{f};
    MlPlan* p = ml_features(c, fr, upsample);
    PVF_REQUIRE(level >= 0 && level < (int)p->lv.size(), "pyramid level out of range");
    const LvDesc& d = p->lv[level];
    *fh = d.fh; *fw = d.fw;
    if (out) {
        // the caller's view is [row][column][32 planes]; the device's is [row][plane group][column][4] with padded rows (feat_at)
        out->resize((size_t)d.fh * d.fw * PVF_FHOG_STRIDE);
        std::vector<float> dev((size_t)d.feat_stride);
        if (!dev.empty())
            HIP_CHECK(hipMemcpyAsync(dev.data(), c->s_feat.as<float>() + d.feat_off, dev.size() * sizeof(float), hipMemcpyDeviceToHost, c->det_stream));
        HIP_CHECK(hipStreamSynchronize(c->det_stream));
        for (int y = 0; y < d.fh; ++y)
            for (int j = 0; j < 8; ++j)
                for (int x = 0; x < d.fw; ++x)
                    memcpy(out->data() + ((size_t)y * d.fw + x) * PVF_FHOG_STRIDE + 4 * j, dev.data() + feat_at(y, j, x, d.fwp), 4 * sizeof(float));
    }
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
}

static bool raw_less(const RawDet& x, const RawDet& y)
{
    if (x.score != y.score) return x.score > y.score;
    if (x.filter != y.filter) return x.filter < y.filter;
    if (x.level != y.level) return x.level < y.level;
    if (x.r != y.r) return x.r < y.r;
    return x.c < y.c;
}

static void decode_candidates(const DetectorModel& m, int upsample, const CandRec* q, int n, std::vector<RawDet>& v);

// the screening pass's counters of a finished batch: a list that overflowed or a feature above the assumed bound => the call is repeated
// on the dense kernel (api.hip: with_candidate_room)
static void screen_verdict(Ctx* c, const int* ctl)
{
    if (!c->det_screen || c->det_screen_suspended) return;
    c->screen_listed += std::min(ctl[SCR_FLAGGED], c->screen_list_cap);
    if (ctl[SCR_FLAGGED] > c->screen_list_cap || ctl[SCR_VIOLATION] != 0) {
        HIP_CHECK(hipStreamSynchronize(c->det_stream));                 // (det_run_many: the next batch is in flight on the shared scratch)
        throw ScreenRetry();
    }
}

void det_run_batch(Ctx* c, const std::vector<Frame>& frames, int upsample, double adjust, std::vector<std::vector<RawDet>>& raw_sorted)
{
    const DetectorModel& m = c->det;
    PVF_REQUIRE(m.loaded, "detector not loaded");
    PVF_REQUIRE(!frames.empty(), "no frames");
    const int B = (int)frames.size();
    const int cap = c->det_cand_cap;
    const size_t cnt_bytes = (((size_t)(B + SCR_CTL_INTS) * sizeof(int) + 63) / 64) * 64;     // per-frame counts + the screening pass's counters
    c->s_cand.ensure((size_t)B * cap * sizeof(CandRec) + cnt_bytes + 64);
    int* d_counts = c->s_cand.as<int>();
    CandRec* d_cands = reinterpret_cast<CandRec*>(c->s_cand.as<uint8_t>() + cnt_bytes);
    HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)(B + SCR_CTL_INTS) * sizeof(int), c->det_stream));
    ScoreParams sp;
    for (int f = 0; f < 8; ++f) sp.thresh[f] = f < m.n_filters ? (float)((double)m.thresh[f] + adjust) : 3.0e38f;
    sp.n_filters = m.n_filters; sp.cap = cap;
    det_run_batch_ml(c, frames, upsample, sp, d_counts, d_cands);
    HIP_CHECK(hipGetLastError());
    c->h_cand.ensure((size_t)B * cap * sizeof(CandRec) + cnt_bytes + 64);
    int* h_counts = c->h_cand.as<int>();
    CandRec* h_cands = reinterpret_cast<CandRec*>(c->h_cand.as<uint8_t>() + cnt_bytes);
    HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, (size_t)(B + SCR_CTL_INTS) * sizeof(int), hipMemcpyDeviceToHost, c->det_stream));
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
    screen_verdict(c, h_counts + B);
    raw_sorted.assign(B, {});
    for (int b = 0; b < B; ++b) {
        const int n = h_counts[b];
        if (n > cap) throw CandOverflow(n);
        if (n == 0) continue;
        HIP_CHECK(hipMemcpyAsync(h_cands + (size_t)b * cap, d_cands + (size_t)b * cap, (size_t)n * sizeof(CandRec), hipMemcpyDeviceToHost,
                                 c->det_stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
    for (int b = 0; b < B; ++b) decode_candidates(m, upsample, h_cands + (size_t)b * cap, h_counts[b], raw_sorted[b]);
}

// ---- many batches with the host work of batch j hidden behind the kernels of batch j + 1 ------------------------------------
// Per batch the host has to wait for the candidate counts, copy the candidates, map them to image rectangles and sort them;
// done batch by batch that leaves the GPU idle for ~1 ms out of every ~13.  Here the kernels of the next batch are queued
// before the results of the current one are collected.  Scratch (pyramid, planes, features) is shared: stream order keeps
// batch j + 1 from touching it before batch j is done; only the candidate buffers and the frame-pointer tables alternate.
static const int DET_PREFETCH = 512;      // candidates per frame copied back unconditionally (more are fetched on demand)

static void decode_candidates(const DetectorModel& m, int upsample, const CandRec* q, int n, std::vector<RawDet>& v)
{
    const int bw = m.fcols - 2 * m.padding, bh = m.frows - 2 * m.padding;
    v.resize(n);
    for (int i = 0; i < n; ++i) {
        long rect[4];
        const long cl = q[i].c - bw / 2, ct = q[i].r - bh / 2;
        fhog_to_image(cl, ct, m.cell, m.frows, m.fcols, &rect[0], &rect[1]);
        fhog_to_image(cl + bw - 1, ct + bh - 1, m.cell, m.frows, m.fcols, &rect[2], &rect[3]);
        for (int k = 0; k < q[i].level; ++k) rect_up6(rect);
        for (int u = 0; u < upsample; ++u) rect_down2i(rect);
        v[i] = RawDet{q[i].score, q[i].filter, q[i].level, q[i].r, q[i].c, (int32_t)rect[0], (int32_t)rect[1], (int32_t)rect[2], (int32_t)rect[3]};
    }
    std::sort(v.begin(), v.end(), raw_less);
}

void det_run_many(Ctx* c, const std::vector<Frame>& frames, int batch, int upsample, double adjust,
                  std::vector<std::vector<RawDet>>& raw_sorted, bool nms)
{
    const DetectorModel& m = c->det;
    PVF_REQUIRE(m.loaded, "detector not loaded");
    PVF_REQUIRE(!frames.empty() && batch > 0, "no frames");
    const int N = (int)frames.size();
    raw_sorted.assign(N, {});
    // equal batches: a 250-frame shot at batch 128 runs as 125 + 125, not 128 + 122.  Launch plans (and the layout of the feature
    // maps, zero border included) are per batch size: two sizes per shot meant two plans taking turns on the feature buffer and a
    // pass re-zeroing the border before EVERY batch (2.6 ms per 1000 frames, profiles/r03_rocprof_kernel_stats.txt: feat_ring_zero_k)
    {
        const int nb = (N + batch - 1) / batch;
        batch = (N + nb - 1) / nb;
    }
    const int cap = c->det_cand_cap, PF = DET_PREFETCH;
    ScoreParams sp;
    for (int f = 0; f < 8; ++f) sp.thresh[f] = f < m.n_filters ? (float)((double)m.thresh[f] + adjust) : 3.0e38f;
    sp.n_filters = m.n_filters; sp.cap = cap;
    const size_t cnt_bytes = (((size_t)(batch + SCR_CTL_INTS) * sizeof(int) + 63) / 64) * 64;     // per-frame counts + the screening pass's counters
    for (int k = 0; k < 2; ++k) {
        c->s_cand2[k].ensure(cnt_bytes + (size_t)batch * cap * sizeof(CandRec));
        c->h_cand2[k].ensure(cnt_bytes + (size_t)batch * PF * sizeof(CandRec));
        if (!c->det_ev[k]) HIP_CHECK(hipEventCreateWithFlags(&c->det_ev[k], hipEventDisableTiming));
    }
    auto submit = [&](int o, int slot) {
        std::vector<Frame> fr(frames.begin() + o, frames.begin() + std::min(N, o + batch));
        const int B = (int)fr.size();
        int* d_counts = c->s_cand2[slot].as<int>();
        CandRec* d_cands = reinterpret_cast<CandRec*>(c->s_cand2[slot].as<uint8_t>() + cnt_bytes);
        HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)(B + SCR_CTL_INTS) * sizeof(int), c->det_stream));
        c->det_slot = slot;
        det_run_batch_ml(c, fr, upsample, sp, d_counts, d_cands);
        HIP_CHECK(hipGetLastError());
        uint8_t* hb = c->h_cand2[slot].as<uint8_t>();
        HIP_CHECK(hipMemcpyAsync(hb, d_counts, (size_t)(B + SCR_CTL_INTS) * sizeof(int), hipMemcpyDeviceToHost, c->det_stream));
        HIP_CHECK(hipMemcpy2DAsync(hb + cnt_bytes, (size_t)PF * sizeof(CandRec), d_cands, (size_t)cap * sizeof(CandRec),
                                   (size_t)PF * sizeof(CandRec), (size_t)B, hipMemcpyDeviceToHost, c->det_stream));
        HIP_CHECK(hipEventRecord(c->det_ev[slot], c->det_stream));
    };
    auto collect = [&](int o, int slot) {
        const int B = std::min(N, o + batch) - o;
        HIP_CHECK(hipEventSynchronize(c->det_ev[slot]));
        const uint8_t* hb = c->h_cand2[slot].as<uint8_t>();
        const int* h_counts = reinterpret_cast<const int*>(hb);
        const CandRec* h_cands = reinterpret_cast<const CandRec*>(hb + cnt_bytes);
        const CandRec* d_cands = reinterpret_cast<const CandRec*>(c->s_cand2[slot].as<uint8_t>() + cnt_bytes);
        screen_verdict(c, h_counts + B);
        for (int b = 0; b < B; ++b)
            if (h_counts[b] > cap) {
                HIP_CHECK(hipStreamSynchronize(c->det_stream));         // the next batch is in flight on the shared scratch: let it finish before the call is repeated
                throw CandOverflow(h_counts[b]);
            }
        // rare: frames with more candidates than were copied back ahead are fetched whole, one after the other
        std::vector<std::vector<CandRec>> big(B);
        bool any_big = false;
        for (int b = 0; b < B; ++b)
            if (h_counts[b] > PF) {
                big[b].resize(h_counts[b]);
                HIP_CHECK(hipMemcpyAsync(big[b].data(), d_cands + (size_t)b * cap, (size_t)h_counts[b] * sizeof(CandRec), hipMemcpyDeviceToHost, c->det_stream));
                any_big = true;
            }
        if (any_big) HIP_CHECK(hipStreamSynchronize(c->det_stream));
        // rectangles, order and suppression per frame: independent, so the frames of a batch are shared out over a few threads.  For every
        // batch but a call's last this runs while the next batch's kernels are queued; the last one's is what the GPU waits for
        auto frames_of = [&](int b0, int b1) {
            std::vector<RawDet> sorted_raw;
            for (int b = b0; b < b1; ++b) {
                const int n = h_counts[b];
                std::vector<RawDet>& dst = nms ? sorted_raw : raw_sorted[o + b];
                decode_candidates(m, upsample, n <= PF ? h_cands + (size_t)b * PF : big[b].data(), n, dst);
                if (nms) det_nms(m, sorted_raw, raw_sorted[o + b]);
            }
        };
        const int nt = std::max(1, std::min(8, B / 12));
        if (nt == 1) frames_of(0, B);
        else {
            std::vector<std::thread> th;
            for (int i = 0; i < nt; ++i) th.emplace_back(frames_of, B * i / nt, B * (i + 1) / nt);
            for (auto& t : th) t.join();
        }
    };
    submit(0, 0);
    int slot = 0;
    for (int o = 0; o < N; o += batch) {
        if (o + batch < N) submit(o + batch, slot ^ 1);
        collect(o, slot);
        slot ^= 1;
    }
    c->det_slot = 0;
}

static bool boxes_overlap(const RawDet& a, const RawDet& b, double iou, double covered)
{
    const long il = std::max(a.l, b.l), it = std::max(a.t, b.t), ir = std::min(a.rr, b.rr), ib = std::min(a.b, b.b);
    if (il > ir || it > ib) return false;
    const double inner = (double)(ir - il + 1) * (double)(ib - it + 1);
    const long ol = std::min(a.l, b.l), ot = std::min(a.t, b.t), orr = std::max(a.rr, b.rr), ob = std::max(a.b, b.b);
    const double outer = (double)(orr - ol + 1) * (double)(ob - ot + 1);
    const double aa = (double)(a.rr - a.l + 1) * (double)(a.b - a.t + 1);
    const double ab = (double)(b.rr - b.l + 1) * (double)(b.b - b.t + 1);
    return inner / outer > iou || inner / aa > covered || inner / ab > covered;
}

void det_nms(const DetectorModel& m, const std::vector<RawDet>& sorted, std::vector<RawDet>& out)
{
    out.clear();
    for (const RawDet& d : sorted) {
        bool hit = false;
        for (const RawDet& k : out) if (boxes_overlap(k, d, m.nms_iou, m.nms_covered)) { hit = true; break; }
        if (!hit) out.push_back(d);
    }
}

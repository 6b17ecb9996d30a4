// chip.hip -- K6: image chips (dlib chip_details / extract_image_chips / transform_image + interpolate_bilinear).
// Used by the face embedder (150x150 aligned chips; reference face.py:74-75) and by the correlation tracker
// (64x64 translation chip, 32 scale chips of 23x23; reference tracking.py:203,251).
// Geometry (a handful of doubles per chip) is planned on the host; pixels never leave HBM.
#include "pvf_internal.h"
#include <algorithm>
#include <cmath>

static void rect_down2(double r[4])
{
    r[0] = r[0] / 2.0 - 1.25; r[1] = r[1] / 2.0 - 0.75;
    r[2] = r[2] / 2.0 - 1.25; r[3] = r[3] / 2.0 - 0.75;
}
static double drect_area(const double r[4]) { return (r[0] > r[2] || r[1] > r[3]) ? 0.0 : (r[2] - r[0]) * (r[3] - r[1]); }
static void rot(double cx, double cy, double x, double y, double cs, double sn, double* ox, double* oy)
{
    const double dx = x - cx, dy = y - cy;
    *ox = cs * dx - sn * dy + cx;
    *oy = sn * dx + cs * dy + cy;
}
static void pyr_down2_dims(int ih, int iw, int* oh, int* ow)
{
    if (ih <= 8 || iw <= 8) { *oh = 0; *ow = 0; return; }
    *oh = (ih - 3) / 2; *ow = (iw - 3) / 2;
}

ChipJob chip_plan(const Frame& f, const ChipDetails& d)
{
    ChipJob j;
    memset(&j, 0, sizeof j);
    j.img = f.d; j.h = f.h; j.w = f.w; j.rows = d.rows; j.cols = d.cols; j.empty = true;
    const double size = (double)d.rows * d.cols;
    const double R[4] = {d.l, d.t, d.r, d.b};
    double grow = 2;
    double rect[4] = {R[0], R[1], R[2], R[3]};
    rect_down2(rect);
    while (drect_area(rect) > size) { rect_down2(rect); grow = grow * 2 + 2; }
    const double cx = (R[0] + R[2]) / 2, cy = (R[1] + R[3]) / 2;
    double xs[4], ys[4];
    rot(cx, cy, R[0], R[1], d.cs, d.sn, &xs[0], &ys[0]);
    rot(cx, cy, R[2], R[1], d.cs, d.sn, &xs[1], &ys[1]);
    rot(cx, cy, R[0], R[3], d.cs, d.sn, &xs[2], &ys[2]);
    rot(cx, cy, R[2], R[3], d.cs, d.sn, &xs[3], &ys[3]);
    double bl = xs[0], bt = ys[0], br = xs[0], bb = ys[0];
    for (int i = 1; i < 4; ++i) {
        if (xs[i] < bl) bl = xs[i];
        if (xs[i] > br) br = xs[i];
        if (ys[i] < bt) bt = ys[i];
        if (ys[i] > bb) bb = ys[i];
    }
    bl -= grow; bt -= grow; br += grow; bb += grow;
    if (bl < 0) bl = 0;
    if (bt < 0) bt = 0;
    if (br > f.w - 1) br = f.w - 1;
    if (bb > f.h - 1) bb = f.h - 1;
    if (bl > br || bt > bb) return j;
    const int bx0 = (int)std::floor(bl + 0.5), by0 = (int)std::floor(bt + 0.5);
    const int bx1 = (int)std::floor(br + 0.5), by1 = (int)std::floor(bb + 0.5);
    const int sw = bx1 - bx0 + 1, sh = by1 - by0 + 1;
    if (sw <= 0 || sh <= 0) return j;
    int level = -1;
    double lr[4] = {R[0] - bx0, R[1] - by0, R[2] - bx0, R[3] - by0};
    for (;;) {
        double nxt[4] = {lr[0], lr[1], lr[2], lr[3]};
        rect_down2(nxt);
        if (!(drect_area(nxt) > size)) break;
        ++level;
        memcpy(lr, nxt, sizeof nxt);
    }
    const double lcx = (lr[0] + lr[2]) / 2, lcy = (lr[1] + lr[3]) / 2;
    double tlx, tly, trx, try_, blx, bly;
    rot(lcx, lcy, lr[0], lr[1], d.cs, d.sn, &tlx, &tly);
    rot(lcx, lcy, lr[2], lr[1], d.cs, d.sn, &trx, &try_);
    rot(lcx, lcy, lr[0], lr[3], d.cs, d.sn, &blx, &bly);
    j.m[0] = (trx - tlx) / (double)(d.cols - 1);
    j.m[2] = (try_ - tly) / (double)(d.cols - 1);
    j.m[1] = (blx - tlx) / (double)(d.rows - 1);
    j.m[3] = (bly - tly) / (double)(d.rows - 1);
    j.b[0] = tlx; j.b[1] = tly;
    j.bx0 = bx0; j.by0 = by0; j.sw = sw; j.sh = sh; j.levels = level + 1;
    j.empty = false;
    return j;
}

struct DevPyrJob { const uint8_t* src; int stride_w, x0, y0; uint8_t* dst; int dh, dw; };
struct DevXfJob { const uint8_t* src; int stride_w, x0, y0, sw, sh; double m[4], b[2]; };

// pyramid_down<2>: separable 1-4-6-4-1 in integers, /256 truncating (exact: every partial sum is an integer below 2^16, any order
// of the additions gives the same value).  A block of 256 threads makes a tile of 32 x 8 output pixels of one job: first the
// HORIZONTAL sums of the 19 source rows the tile needs (19 x 32 x 3 integers, parked in LDS; a sum's five source pixels are 15
// consecutive bytes = four unaligned dwords at +0, +4, +8, +11), then every thread adds five of them for its output pixel.
// Round 4 had every output pixel read its 5 x 5 source pixels itself, byte by byte: 75 loads per pixel; now 9.5 dword loads.
#define PD_TW 32
#define PD_TH 8
__global__ void __launch_bounds__(256) pyr_down2_k(const DevPyrJob* __restrict__ jobs)
{
    __shared__ uint16_t hs[2 * PD_TH + 3][PD_TW][4];            // horizontal sums (<= 16 x 255): [source row][output column][channel]
    const DevPyrJob j = jobs[blockIdx.z];
    const int c0 = blockIdx.x * PD_TW, r0 = blockIdx.y * PD_TH;
    if (j.dst == nullptr || r0 >= j.dh || c0 >= j.dw) return;   // block-uniform
    const int tid = threadIdx.x;
    const int rows_in = min(2 * PD_TH + 3, 2 * (j.dh - r0) + 3);     // source rows 2 r0 .. 2 r0 + rows_in - 1 exist for the output rows of this tile
    for (int idx = tid; idx < (2 * PD_TH + 3) * PD_TW; idx += 256) {
        const int sr = idx / PD_TW, oc = idx % PD_TW;
        if (sr >= rows_in || c0 + oc >= j.dw) continue;
        const uint8_t* p = j.src + ((size_t)(j.y0 + 2 * r0 + sr) * j.stride_w + (j.x0 + 2 * (c0 + oc))) * 3;
        const uint32_t d0 = *reinterpret_cast<const uint32_t*>(p), d1 = *reinterpret_cast<const uint32_t*>(p + 4);
        const uint32_t d2 = *reinterpret_cast<const uint32_t*>(p + 8), d3 = *reinterpret_cast<const uint32_t*>(p + 11);
        // pixel q, channel k = byte 3 q + k of the 15 (d3 holds bytes 11 .. 14)
        const uint32_t h0 = (d0 & 0xffu) + 4 * (d0 >> 24) + 6 * ((d1 >> 16) & 0xffu) + 4 * ((d2 >> 8) & 0xffu) + ((d3 >> 8) & 0xffu);
        const uint32_t h1 = ((d0 >> 8) & 0xffu) + 4 * (d1 & 0xffu) + 6 * (d1 >> 24) + 4 * ((d2 >> 16) & 0xffu) + ((d3 >> 16) & 0xffu);
        const uint32_t h2 = ((d0 >> 16) & 0xffu) + 4 * ((d1 >> 8) & 0xffu) + 6 * (d2 & 0xffu) + 4 * (d2 >> 24) + (d3 >> 24);
        *reinterpret_cast<uint2*>(&hs[sr][oc][0]) = make_uint2(h0 | (h1 << 16), h2);
    }
    __syncthreads();
    const int c = c0 + (tid & (PD_TW - 1)), rl = tid >> 5, r = r0 + rl;
    if (r >= j.dh || c >= j.dw) return;
    const int oc = tid & (PD_TW - 1);
    uint32_t acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        acc[k] = hs[2 * rl][oc][k] + 4u * hs[2 * rl + 1][oc][k] + 6u * hs[2 * rl + 2][oc][k] + 4u * hs[2 * rl + 3][oc][k] + hs[2 * rl + 4][oc][k];
    uint8_t* o = j.dst + ((size_t)r * j.dw + c) * 3;
    o[0] = (uint8_t)(acc[0] / 256); o[1] = (uint8_t)(acc[1] / 256); o[2] = (uint8_t)(acc[2] / 256);
}

// transform_image + interpolate_bilinear (black outside, truncating store)
__global__ void __launch_bounds__(256) transform_k(const DevXfJob* __restrict__ jobs, uint8_t* __restrict__ out, int rows, int cols)
{
    const DevXfJob j = jobs[blockIdx.z];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    uint8_t* o = out + ((size_t)blockIdx.z * rows * cols + (size_t)r * cols + c) * 3;
    if (j.src == nullptr) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
    const double px = j.m[0] * c + j.m[1] * r + j.b[0];
    const double py = j.m[2] * c + j.m[3] * r + j.b[1];
    const double fx = floor(px), fy = floor(py);
    if (!(fx >= 0 && fy >= 0 && fx + 1 < j.sw && fy + 1 < j.sh)) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
    const int left = (int)fx, top = (int)fy;
    const double lr = px - left, tb = py - top;
    // the two pixels of a source row are 6 consecutive bytes: two (unaligned) dwords at +0 and +2 cover exactly them
    const uint8_t* ptl = j.src + ((size_t)(j.y0 + top) * j.stride_w + (j.x0 + left)) * 3;
    const uint8_t* pbl = ptl + (size_t)j.stride_w * 3;
    const uint32_t t0 = *reinterpret_cast<const uint32_t*>(ptl), t1 = *reinterpret_cast<const uint32_t*>(ptl + 2);
    const uint32_t b0 = *reinterpret_cast<const uint32_t*>(pbl), b1 = *reinterpret_cast<const uint32_t*>(pbl + 2);
    const uint32_t tl3[3] = {t0 & 0xffu, (t0 >> 8) & 0xffu, (t0 >> 16) & 0xffu}, tr3[3] = {t0 >> 24, (t1 >> 16) & 0xffu, t1 >> 24};
    const uint32_t bl3[3] = {b0 & 0xffu, (b0 >> 8) & 0xffu, (b0 >> 16) & 0xffu}, br3[3] = {b0 >> 24, (b1 >> 16) & 0xffu, b1 >> 24};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double tl = tl3[k], tr = tr3[k], bl = bl3[k], br = br3[k];
        const double v = (1 - tb) * ((1 - lr) * tl + lr * tr) + tb * ((1 - lr) * bl + lr * br);
        o[k] = (uint8_t)v;
    }
}

// plain affine sampling straight from the frame (tracker scale space): jobs carry m,b; sub image = whole frame
void transform_batch(Ctx* c, const std::vector<ChipJob>& jobs, uint8_t* d_out)
{
    const int n = (int)jobs.size();
    if (n == 0) return;
    const int rows = jobs[0].rows, cols = jobs[0].cols;
    std::vector<DevXfJob> xf(n);
    for (int i = 0; i < n; ++i) {
        const ChipJob& j = jobs[i];
        DevXfJob& x = xf[i];
        x.src = j.img; x.stride_w = j.w; x.x0 = 0; x.y0 = 0; x.sw = j.w; x.sh = j.h;
        memcpy(x.m, j.m, sizeof x.m); memcpy(x.b, j.b, sizeof x.b);
    }
    c->s_chip.ensure(n * sizeof(DevXfJob));
    void* hb = c->stage.take(n * sizeof(DevXfJob));
    memcpy(hb, xf.data(), n * sizeof(DevXfJob));
    HIP_CHECK(hipMemcpyAsync(c->s_chip.p, hb, n * sizeof(DevXfJob), hipMemcpyHostToDevice, c->stream));
    c->stage.sent(c->stream);
    ProfScope ps(c, "chip");
    hipLaunchKernelGGL(transform_k, dim3((cols + 63) / 64, rows, n), dim3(64), 0, c->stream, c->s_chip.as<DevXfJob>(), d_out, rows, cols);
}

void chip_extract_batch(Ctx* c, const std::vector<ChipJob>& jobs, uint8_t* d_out)
{
    const int n = (int)jobs.size();
    if (n == 0) return;
    const int rows = jobs[0].rows, cols = jobs[0].cols;
    // lay out the pyramid arena
    int max_levels = 0;
    for (auto& j : jobs) if (!j.empty) max_levels = std::max(max_levels, j.levels);
    std::vector<std::vector<DevPyrJob>> pj(max_levels, std::vector<DevPyrJob>(n));
    std::vector<DevXfJob> xf(n);
    size_t arena = 0;
    std::vector<size_t> offs; // per (job, level) offset, resolved after the arena is allocated
    struct Slot { int job, level; size_t off; int h, w; };
    std::vector<Slot> slots;
    for (int i = 0; i < n; ++i) {
        const ChipJob& j = jobs[i];
        int ch = j.sh, cw = j.sw;
        for (int l = 0; l < j.levels && !j.empty; ++l) {
            int nh, nw;
            pyr_down2_dims(ch, cw, &nh, &nw);
            slots.push_back({i, l, arena, nh, nw});
            arena += ((size_t)std::max(nh, 0) * std::max(nw, 0) * 3 + 63) / 64 * 64;
            ch = nh; cw = nw;
        }
    }
    c->s_chip_pyr.ensure(arena + 64);
    uint8_t* base = c->s_chip_pyr.as<uint8_t>();
    for (int l = 0; l < max_levels; ++l)
        for (int i = 0; i < n; ++i) pj[l][i] = DevPyrJob{nullptr, 0, 0, 0, nullptr, 0, 0};
    std::vector<int> max_h(max_levels, 0), max_w(max_levels, 0);
    for (int i = 0; i < n; ++i) {
        const ChipJob& j = jobs[i];
        DevXfJob& x = xf[i];
        memset(&x, 0, sizeof x);
        if (j.empty) continue;
        memcpy(x.m, j.m, sizeof x.m); memcpy(x.b, j.b, sizeof x.b);
        x.src = j.img; x.stride_w = j.w; x.x0 = j.bx0; x.y0 = j.by0; x.sw = j.sw; x.sh = j.sh;
    }
    {
        // resolve slots in order: level l of job i reads level l-1 (or the frame)
        std::vector<const uint8_t*> prev_ptr(n, nullptr);
        std::vector<int> prev_h(n, 0), prev_w(n, 0);
        for (const Slot& s : slots) {
            const ChipJob& j = jobs[s.job];
            DevPyrJob& p = pj[s.level][s.job];
            if (s.level == 0) { p.src = j.img; p.stride_w = j.w; p.x0 = j.bx0; p.y0 = j.by0; }
            else { p.src = prev_ptr[s.job]; p.stride_w = prev_w[s.job]; p.x0 = 0; p.y0 = 0; }
            p.dst = (s.h > 0 && s.w > 0) ? base + s.off : nullptr;
            p.dh = s.h; p.dw = s.w;
            max_h[s.level] = std::max(max_h[s.level], s.h);
            max_w[s.level] = std::max(max_w[s.level], s.w);
            prev_ptr[s.job] = base + s.off; prev_h[s.job] = s.h; prev_w[s.job] = s.w;
            if (s.level == j.levels - 1) {
                DevXfJob& x = xf[s.job];
                if (s.h > 0 && s.w > 0) { x.src = base + s.off; x.stride_w = s.w; x.x0 = 0; x.y0 = 0; x.sw = s.w; x.sh = s.h; }
                else x.src = nullptr;
            }
        }
    }
    // upload descriptors: [levels][n] pyr jobs then [n] xf jobs
    const size_t pyr_bytes = (size_t)max_levels * n * sizeof(DevPyrJob);
    const size_t total = pyr_bytes + n * sizeof(DevXfJob) + 64;
    c->s_chip.ensure(total);
    uint8_t* hb = reinterpret_cast<uint8_t*>(c->stage.take(total));
    for (int l = 0; l < max_levels; ++l) memcpy(hb + (size_t)l * n * sizeof(DevPyrJob), pj[l].data(), n * sizeof(DevPyrJob));
    const size_t xf_off = (pyr_bytes + 15) / 16 * 16;
    memcpy(hb + xf_off, xf.data(), n * sizeof(DevXfJob));
    HIP_CHECK(hipMemcpyAsync(c->s_chip.p, hb, xf_off + n * sizeof(DevXfJob), hipMemcpyHostToDevice, c->stream));
    c->stage.sent(c->stream);
    ProfScope ps(c, "chip");
    for (int l = 0; l < max_levels; ++l) {
        if (max_h[l] <= 0 || max_w[l] <= 0) continue;
        hipLaunchKernelGGL(pyr_down2_k, dim3((max_w[l] + PD_TW - 1) / PD_TW, (max_h[l] + PD_TH - 1) / PD_TH, n), dim3(256), 0, c->stream,
                           reinterpret_cast<const DevPyrJob*>(c->s_chip.as<uint8_t>() + (size_t)l * n * sizeof(DevPyrJob)));
    }
    hipLaunchKernelGGL(transform_k, dim3((cols + 63) / 64, rows, n), dim3(64), 0, c->stream,
                       reinterpret_cast<const DevXfJob*>(c->s_chip.as<uint8_t>() + xf_off), d_out, rows, cols);
    // (no synchronisation: the descriptors went through a staging buffer of their own, the device scratch is reused in stream order)
}

// api.hip -- extern "C" entry points declared in include/pvface.h (context/model/frame calls live in ctx.hip).
#include "pvf_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <thread>

#define API_BEGIN try {
#define API_END                                                        \
    return 0;                                                          \
    }                                                                  \
    catch (const std::exception& e) { pvf_set_error(e.what()); return -1; } \
    catch (...) { pvf_set_error("unknown error"); return -2; }

// every compute entry point runs under its context's api_mu: a context serialises its own calls, whichever threads make them
#define ENTER(c, h)                                                \
    Ctx* c = pvf_ctx(h);                                           \
    std::lock_guard<std::recursive_mutex> _api_lock(c->api_mu);    \
    HIP_CHECK(hipSetDevice(c->device))
// the detector side (its own stream, scratch and lock: runs beside the calls above)
#define ENTER_DET(c, h)                                            \
    Ctx* c = pvf_ctx(h);                                           \
    std::lock_guard<std::recursive_mutex> _det_lock(c->det_mu);    \
    HIP_CHECK(hipSetDevice(c->device))
// debug entries that use scratch of both sides
#define ENTER_BOTH(c, h)                                           \
    Ctx* c = pvf_ctx(h);                                           \
    std::lock_guard<std::recursive_mutex> _det_lock(c->det_mu);    \
    std::lock_guard<std::recursive_mutex> _api_lock(c->api_mu);    \
    HIP_CHECK(hipSetDevice(c->device))

// ---- S1 -------------------------------------------------------------------------------------------
// a frame with more raw candidates than the candidate slots hold (a threshold lowered far below the operating point, a pathological
// image): the slots are enlarged to what the frame asked for and the call runs again -- the scanner itself has no limit (dlib collects
// every window above the threshold)
template <class F>
static void with_candidate_room(Ctx* c, F&& run)
{
    for (;;) {
        try { run(); return; }
        catch (const CandOverflow& e) {
            PVF_REQUIRE(e.needed <= (1 << 22), "detector: more than 4M candidates in one frame");
            int cap = c->det_cand_cap;
            while (cap < e.needed) cap *= 2;
            c->det_cand_cap = cap;
        }
        catch (const ScreenRetry&) {
            // the screening pass listed more windows than its list holds (a threshold far below the operating point) or met a feature above
            // the bound its error analysis assumes: this call runs on the dense exact kernel
            PVF_REQUIRE(!c->det_screen_suspended, "detector: the dense scoring kernel asked for a retry");
            c->det_screen_suspended = true;
            ++c->screen_retries;
            struct Resume { Ctx* c; ~Resume() { c->det_screen_suspended = false; } } resume{c};
            with_candidate_room(c, run);
            return;
        }
    }
}

extern "C" int32_t pvf_detector_screening(pvf_handle h, int32_t on, int32_t list_cap)
{
    API_BEGIN
    ENTER_DET(c, h);
    PVF_REQUIRE(list_cap >= 0 && list_cap <= (1 << 26), "pvf_detector_screening: list_cap must be 0 (keep) .. 2^26");
    c->det_screen = on != 0;
    if (list_cap > 0) c->screen_list_cap = list_cap;
    API_END
}

extern "C" int32_t pvf_detector_screening_stats(pvf_handle h, int64_t* batches, int64_t* listed, int64_t* retries, double* bounds, double* pipe_err)
{
    API_BEGIN
    ENTER_DET(c, h);
    if (batches) *batches = c->screen_batches;
    if (listed) *listed = c->screen_listed;
    if (retries) *retries = c->screen_retries;
    if (bounds) for (int f = 0; f < c->det.n_filters; ++f) bounds[f] = c->det.screen_bound[f];
    if (pipe_err) *pipe_err = c->screen_pipe_err;
    API_END
}

extern "C" int32_t pvf_detect_batch(pvf_handle h, const pvf_handle* frames, int32_t n_frames, int32_t upsample, double adjust,
                                    pvf_rect_i32* out, float* scores, int32_t* counts, int32_t cap)
{
    API_BEGIN
    ENTER_DET(c, h);
    PVF_REQUIRE(n_frames > 0 && frames && out && counts && cap > 0, "pvf_detect_batch: bad arguments");
    PVF_REQUIRE(upsample >= 0 && upsample <= 2, "pvf_detect_batch: upsample must be 0..2");
    std::vector<Frame> fr(n_frames);
    for (int i = 0; i < n_frames; ++i) fr[i] = c->frame(frames[i]);
    std::vector<std::vector<RawDet>> raw;
    with_candidate_room(c, [&]() { det_run_batch(c, fr, upsample, adjust, raw); });
    std::vector<RawDet> kept;
    for (int i = 0; i < n_frames; ++i) {
        det_nms(c->det, raw[i], kept);
        const int n = std::min<int>((int)kept.size(), cap);
        counts[i] = n;
        for (int k = 0; k < n; ++k) {
            out[(size_t)i * cap + k] = pvf_rect_i32{kept[k].l, kept[k].t, kept[k].rr, kept[k].b};
            if (scores) scores[(size_t)i * cap + k] = kept[k].score;
        }
    }
    API_END
}

extern "C" int32_t pvf_detect_many(pvf_handle h, const pvf_handle* frames, int32_t n_frames, int32_t batch, int32_t upsample, double adjust,
                                   pvf_rect_i32* out, float* scores, int32_t* counts, int32_t cap)
{
    API_BEGIN
    ENTER_DET(c, h);
    PVF_REQUIRE(n_frames > 0 && batch > 0 && frames && out && counts && cap > 0, "pvf_detect_many: bad arguments");
    PVF_REQUIRE(upsample >= 0 && upsample <= 2, "pvf_detect_many: upsample must be 0..2");
    std::vector<Frame> fr(n_frames);
    for (int i = 0; i < n_frames; ++i) fr[i] = c->frame(frames[i]);
    std::vector<std::vector<RawDet>> kept_all;
    with_candidate_room(c, [&]() { det_run_many(c, fr, batch, upsample, adjust, kept_all, true); });
    for (int i = 0; i < n_frames; ++i) {
        const std::vector<RawDet>& kept = kept_all[i];
        const int n = std::min<int>((int)kept.size(), cap);
        counts[i] = n;
        for (int k = 0; k < n; ++k) {
            out[(size_t)i * cap + k] = pvf_rect_i32{kept[k].l, kept[k].t, kept[k].rr, kept[k].b};
            if (scores) scores[(size_t)i * cap + k] = kept[k].score;
        }
    }
    API_END
}

extern "C" int32_t pvf_detect(pvf_handle h, pvf_handle frame, int32_t upsample, double adjust, pvf_rect_i32* out, float* scores,
                              int32_t cap, int32_t* n)
{
    int32_t cnt = 0;
    const int32_t rc = pvf_detect_batch(h, &frame, 1, upsample, adjust, out, scores, &cnt, cap);
    if (rc == 0 && n) *n = cnt;
    return rc;
}

extern "C" int32_t pvf_debug_detect_raw(pvf_handle h, pvf_handle frame, int32_t upsample, double adjust, float* scores, int32_t* meta,
                                        int32_t cap, int32_t* n)
{
    API_BEGIN
    ENTER_DET(c, h);
    std::vector<Frame> fr{c->frame(frame)};
    std::vector<std::vector<RawDet>> raw;
    with_candidate_room(c, [&]() { det_run_batch(c, fr, upsample, adjust, raw); });
    const int k = std::min<int>((int)raw[0].size(), cap);
    *n = (int)raw[0].size();
    for (int i = 0; i < k; ++i) {
        const RawDet& d = raw[0][i];
        scores[i] = d.score;
        int32_t* m = meta + (size_t)i * 8;
        m[0] = d.filter; m[1] = d.level; m[2] = d.r; m[3] = d.c; m[4] = d.l; m[5] = d.t; m[6] = d.rr; m[7] = d.b;
    }
    API_END
}

extern "C" int32_t pvf_debug_detect_raw_many(pvf_handle h, const pvf_handle* frames, int32_t n_frames, int32_t batch, int32_t upsample, double adjust,
                                             int32_t* counts, int32_t* rows, int64_t cap, int64_t* total)
{
    API_BEGIN
    ENTER_DET(c, h);
    PVF_REQUIRE(n_frames > 0 && batch > 0 && frames && counts && total && (rows || cap == 0), "pvf_debug_detect_raw_many: bad arguments");
    PVF_REQUIRE(upsample >= 0 && upsample <= 2, "pvf_debug_detect_raw_many: upsample must be 0..2");
    std::vector<Frame> fr(n_frames);
    for (int i = 0; i < n_frames; ++i) fr[i] = c->frame(frames[i]);
    std::vector<std::vector<RawDet>> raw;
    with_candidate_room(c, [&]() { det_run_many(c, fr, batch, upsample, adjust, raw, false); });
    int64_t k = 0;
    for (int i = 0; i < n_frames; ++i) {
        counts[i] = (int32_t)raw[i].size();
        for (const RawDet& d : raw[i]) {
            if (k < cap) {
                int32_t* m = rows + (size_t)k * 5;
                m[0] = d.level; m[1] = d.filter; m[2] = d.r; m[3] = d.c;
                memcpy(&m[4], &d.score, 4);
            }
            ++k;
        }
    }
    *total = k;
    API_END
}

extern "C" int32_t pvf_debug_pyramid_level(pvf_handle h, pvf_handle frame, int32_t upsample, int32_t level, uint8_t* out, int32_t* oh,
                                           int32_t* ow)
{
    API_BEGIN
    ENTER_DET(c, h);
    std::vector<uint8_t> buf;
    int hh = 0, ww = 0;
    det_pyramid_level(c, c->frame(frame), upsample, level, out ? &buf : nullptr, &hh, &ww);
    *oh = hh; *ow = ww;
    if (out) memcpy(out, buf.data(), buf.size());
    API_END
}

extern "C" int32_t pvf_debug_pyramid_batch(pvf_handle h, const pvf_handle* frames, int32_t n, int32_t upsample)
{
    API_BEGIN
    ENTER_DET(c, h);
    PVF_REQUIRE(frames && n > 0, "pvf_debug_pyramid_batch: bad arguments");
    std::vector<Frame> fr;
    for (int i = 0; i < n; ++i) fr.push_back(c->frame(frames[i]));
    det_pyramid_batch(c, fr, upsample);
    API_END
}

extern "C" int32_t pvf_debug_level_features(pvf_handle h, pvf_handle frame, int32_t upsample, int32_t level, float* out, int32_t* fh, int32_t* fw)
{
    API_BEGIN
    ENTER_DET(c, h);
    std::vector<float> buf;
    int a = 0, b = 0;
    det_level_features(c, c->frame(frame), upsample, level, out ? &buf : nullptr, &a, &b);
    *fh = a; *fw = b;
    if (out) memcpy(out, buf.data(), buf.size() * sizeof(float));
    API_END
}

extern "C" int32_t pvf_debug_fhog(pvf_handle h, const uint8_t* img, int32_t ih, int32_t iw, int32_t cell, int32_t pad_r, int32_t pad_c,
                                  float* out, int32_t* fh, int32_t* fw)
{
    API_BEGIN
    ENTER_BOTH(c, h);
    int a = 0, b = 0;
    if (!out) { fhog_dims(ih, iw, cell, pad_r, pad_c, &a, &b); *fh = a; *fw = b; return 0; }
    std::vector<float> buf;
    fhog_debug(c, img, ih, iw, cell, pad_r, pad_c, buf, &a, &b);
    *fh = a; *fw = b;
    memcpy(out, buf.data(), buf.size() * sizeof(float));
    API_END
}

// ---- S2 -------------------------------------------------------------------------------------------
static pvf_handle tracker_new(Ctx* c)
{
    std::unique_ptr<Tracker> t(new Tracker());          // device state: taken by the first start_track (dsst_start_many) or shared by a clone
    const uint64_t id = c->next_id++;
    c->trackers[id] = std::move(t);
    return id;
}

extern "C" int32_t pvf_tracker_create(pvf_handle h, pvf_handle* trk)
{
    API_BEGIN
    ENTER(c, h);
    *trk = tracker_new(c);
    API_END
}

extern "C" int32_t pvf_tracker_create_many(pvf_handle h, int32_t n, pvf_handle* trks)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(n >= 0, "negative count");
    for (int i = 0; i < n; ++i) trks[i] = tracker_new(c);
    API_END
}

extern "C" int32_t pvf_tracker_clone_many(pvf_handle h, const pvf_handle* src, int32_t n, pvf_handle* dst)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(n >= 0, "negative count");
    std::vector<Tracker*> s(n), d(n);
    for (int i = 0; i < n; ++i) s[i] = &c->tracker(src[i]);
    for (int i = 0; i < n; ++i) { dst[i] = tracker_new(c); d[i] = &c->tracker(dst[i]); }
    dsst_clone_many(c, s, d);
    API_END
}

// released tracker states are kept for reuse up to this many (2.39 MB each); beyond it they go back to the allocator, so a long shot's
// burst of trackers does not stay resident for the rest of the video
static const size_t TRACKER_POOL_KEEP = 4096;
static void pool_tracker_state(Ctx* c, Tracker& t)
{
    double* d_state = t.d_state;
    t.d_state = nullptr;
    if (t.share) {                                   // clones: the last owner returns the buffer
        int* sh = t.share;
        t.share = nullptr;
        if (--*sh > 0) return;
        delete sh;
    }
    if (!d_state) return;
    if (c->tracker_pool.size() < TRACKER_POOL_KEEP) { c->tracker_pool.push_back(d_state); return; }
    HIP_CHECK(hipStreamSynchronize(c->stream));      // kernels that still read the state have finished
    HIP_CHECK(hipFree(d_state));
}

extern "C" int32_t pvf_tracker_destroy_many(pvf_handle h, const pvf_handle* trks, int32_t n)
{
    API_BEGIN
    ENTER(c, h);
    for (int i = 0; i < n; ++i) {
        auto it = c->trackers.find(trks[i]);
        PVF_REQUIRE(it != c->trackers.end(), "unknown tracker handle");
        pool_tracker_state(c, *it->second);
        c->trackers.erase(it);
    }
    API_END
}

extern "C" int32_t pvf_tracker_destroy(pvf_handle h, pvf_handle trk)
{
    API_BEGIN
    ENTER(c, h);
    auto it = c->trackers.find(trk);
    PVF_REQUIRE(it != c->trackers.end(), "unknown tracker handle");
    pool_tracker_state(c, *it->second);
    c->trackers.erase(it);
    API_END
}
extern "C" int32_t pvf_tracker_start_many(pvf_handle h, const pvf_handle* trks, const pvf_handle* frames, const double* boxes, int32_t n)
{
    API_BEGIN
    ENTER(c, h);
    std::vector<Tracker*> t(n);
    std::vector<Frame> f(n);
    for (int i = 0; i < n; ++i) { t[i] = &c->tracker(trks[i]); f[i] = c->frame(frames[i]); }
    dsst_start_many(c, t, f, boxes);
    API_END
}
extern "C" int32_t pvf_tracker_update_many(pvf_handle h, const pvf_handle* trks, const pvf_handle* frames, int32_t n, double* psr,
                                           double* boxes_out)
{
    API_BEGIN
    ENTER(c, h);
    std::vector<Tracker*> t(n);
    std::vector<Frame> f(n);
    for (int i = 0; i < n; ++i) { t[i] = &c->tracker(trks[i]); f[i] = c->frame(frames[i]); }
    dsst_update_many(c, t, f, psr, boxes_out);
    API_END
}
extern "C" int32_t pvf_tracker_start(pvf_handle h, pvf_handle trk, pvf_handle frame, const double box[4])
{
    return pvf_tracker_start_many(h, &trk, &frame, box, 1);
}
extern "C" int32_t pvf_tracker_update_many_deferred(pvf_handle h, const pvf_handle* trks, const pvf_handle* frames, int32_t n, double* psr,
                                                    double* boxes_out)
{
    API_BEGIN
    ENTER(c, h);
    std::vector<Tracker*> t(n);
    std::vector<Frame> f(n);
    for (int i = 0; i < n; ++i) { t[i] = &c->tracker(trks[i]); f[i] = c->frame(frames[i]); }
    dsst_update_many(c, t, f, psr, boxes_out, 1);
    API_END
}

extern "C" int32_t pvf_tracker_commit_many(pvf_handle h, const pvf_handle* trks, const pvf_handle* frames, int32_t n)
{
    API_BEGIN
    ENTER(c, h);
    std::vector<Tracker*> t(n);
    std::vector<Frame> f(n);
    for (int i = 0; i < n; ++i) { t[i] = &c->tracker(trks[i]); f[i] = c->frame(frames[i]); }
    dsst_update_many(c, t, f, nullptr, nullptr, 2);
    API_END
}

extern "C" int32_t pvf_tracker_update(pvf_handle h, pvf_handle trk, pvf_handle frame, double* psr)
{
    return pvf_tracker_update_many(h, &trk, &frame, 1, psr, nullptr);
}
extern "C" int32_t pvf_tracker_position(pvf_handle h, pvf_handle trk, double box[4])
{
    API_BEGIN
    ENTER(c, h);
    memcpy(box, c->tracker(trk).pos, 4 * sizeof(double));
    API_END
}
extern "C" int32_t pvf_debug_tracker_state(pvf_handle h, pvf_handle trk, double* F, double* A, double* B)
{
    API_BEGIN
    ENTER(c, h);
    Tracker& t = c->tracker(trk);
    PVF_REQUIRE(t.d_state, "tracker has no state yet (start_track first)");
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (F && c->s_trk1.p) HIP_CHECK(hipMemcpy(F, c->s_trk1.p, (size_t)32 * 64 * 64 * 2 * sizeof(double), hipMemcpyDeviceToHost));
    if (A) HIP_CHECK(hipMemcpy(A, t.d_state + TRK_A, (size_t)32 * 64 * 64 * 2 * sizeof(double), hipMemcpyDeviceToHost));
    if (B) HIP_CHECK(hipMemcpy(B, t.d_state + TRK_B, (size_t)64 * 64 * sizeof(double), hipMemcpyDeviceToHost));
    API_END
}

extern "C" int32_t pvf_shot_dfd(pvf_handle h, const pvf_handle* frames, int32_t n, int32_t width, int32_t height, const float* tables,
                                double* dfd, uint8_t* gray_out, float* flow_out)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(n >= 1 && frames && tables && (n == 1 || dfd), "shot: frames, tables and an output array");
    std::vector<Frame> f(n);
    for (int i = 0; i < n; ++i) f[i] = c->frame(frames[i]);
    shot_dfd(c, f, width, height, tables, dfd, gray_out, flow_out);
    API_END
}

// ---- S3 (host) --------------------------------------------------------------------------------------
static double darea(double l, double t, double r, double b) { return (l > r || t > b) ? 0.0 : (r - l) * (b - t); }

void overlap_matrix_host(const double* a, int na, const double* b, int nb, double ratio, double* out)
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            const double* p = a + 4 * i;
            const double* q = b + 4 * j;
            const double il = std::max(p[0], q[0]), it = std::max(p[1], q[1]);
            const double ir = std::min(p[2], q[2]), ib = std::min(p[3], q[3]);
            double ov = darea(il, it, ir, ib);
            if (ov < ratio * darea(p[0], p[1], p[2], p[3]) || ov < ratio * darea(q[0], q[1], q[2], q[3])) ov = 0.0;
            out[(size_t)i * nb + j] = ov;
        }
}

// Kuhn-Munkres exactly as `munkres` 1.1.4 runs it (reference tracking.py:35,121,172 -> Munkres().compute(cost)): the same six
// steps on the same matrix in the same scan orders, because among equally good assignments the step order decides which one
// comes out (and with it the track ids).  Two details of that package are easy to miss and are kept:
//   - step 4 resumes its search for an uncovered zero at the row of the last primed zero and at the column of that row's
//     starred zero, wrapping around cyclically (Munkres.__step4 / __find_a_zero(i0, j0));
//   - inside the first row that has an uncovered zero the search does not stop at the first hit: it keeps scanning the row
//     (cyclically from j0) and returns the LAST uncovered zero of that row.
// Pinned against the package itself in tests/test_reference_pins.py (ties included).
namespace {
struct Hungarian {
    int n;
    std::vector<double> C;
    std::vector<char> mark, rowc, colc;
    std::vector<int> path;
    int z0r = 0, z0c = 0;
    explicit Hungarian(const double* cost, int n_) : n(n_), C(cost, cost + (size_t)n_ * n_), mark((size_t)n_ * n_, 0), rowc(n_, 0), colc(n_, 0), path(4 * n_ + 8, 0) {}
    void clear_covers() { std::fill(rowc.begin(), rowc.end(), 0); std::fill(colc.begin(), colc.end(), 0); }
    int reduce_rows()                                       // __step1
    {
        for (int i = 0; i < n; ++i) {
            double mn = C[(size_t)i * n];
            for (int j = 1; j < n; ++j) mn = std::min(mn, C[(size_t)i * n + j]);
            for (int j = 0; j < n; ++j) C[(size_t)i * n + j] -= mn;
        }
        return 2;
    }
    int star_initial()                                      // __step2
    {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (C[(size_t)i * n + j] == 0 && !colc[j] && !rowc[i]) { mark[(size_t)i * n + j] = 1; colc[j] = 1; rowc[i] = 1; break; }
        clear_covers();
        return 3;
    }
    int cover_starred()                                     // __step3
    {
        int count = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (mark[(size_t)i * n + j] == 1 && !colc[j]) { colc[j] = 1; ++count; }
        return count >= n ? 7 : 4;
    }
    // __find_a_zero(i0, j0): rows cyclically from i0; in the first row holding an uncovered zero, the last such zero in the
    // cyclic column order that starts at j0
    void find_a_zero(int i0, int j0, int* row, int* col) const
    {
        *row = -1; *col = -1;
        int i = i0;
        bool done = false;
        while (!done) {
            int j = j0;
            for (;;) {
                if (C[(size_t)i * n + j] == 0 && !rowc[i] && !colc[j]) { *row = i; *col = j; done = true; }
                j = (j + 1) % n;
                if (j == j0) break;
            }
            i = (i + 1) % n;
            if (i == i0) done = true;
        }
    }
    int prime_zeros()                                       // __step4
    {
        int row = 0, col = 0;
        for (;;) {
            int r, cc;
            find_a_zero(row, col, &r, &cc);
            if (r < 0) return 6;
            row = r; col = cc;
            mark[(size_t)row * n + col] = 2;
            int star = -1;
            for (int j = 0; j < n; ++j) if (mark[(size_t)row * n + j] == 1) { star = j; break; }
            if (star < 0) { z0r = row; z0c = col; return 5; }
            col = star;
            rowc[row] = 1; colc[col] = 0;
        }
    }
    int augment()                                           // __step5
    {
        int count = 0;
        path[0] = z0r; path[1] = z0c;
        for (;;) {
            int row = -1;
            for (int i = 0; i < n; ++i) if (mark[(size_t)i * n + path[2 * count + 1]] == 1) { row = i; break; }
            if (row < 0) break;
            ++count; path[2 * count] = row; path[2 * count + 1] = path[2 * (count - 1) + 1];
            int col = -1;
            for (int j = 0; j < n; ++j) if (mark[(size_t)path[2 * count] * n + j] == 2) { col = j; break; }
            ++count; path[2 * count] = path[2 * (count - 1)]; path[2 * count + 1] = col;
        }
        for (int k = 0; k <= count; ++k) {
            char& m = mark[(size_t)path[2 * k] * n + path[2 * k + 1]];
            m = (m == 1) ? 0 : 1;
        }
        clear_covers();
        for (auto& m : mark) if (m == 2) m = 0;
        return 3;
    }
    int shift_costs()                                       // __step6 (+= then -= on an element that gets both, like the package)
    {
        double mn = 0; bool have = false;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (!rowc[i] && !colc[j] && (!have || C[(size_t)i * n + j] < mn)) { mn = C[(size_t)i * n + j]; have = true; }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (rowc[i]) C[(size_t)i * n + j] += mn;
                if (!colc[j]) C[(size_t)i * n + j] -= mn;
            }
        return 4;
    }
    void solve(int32_t* row_to_col)
    {
        int step = 1;
        while (step != 7) {
            switch (step) {
                case 1: step = reduce_rows(); break;
                case 2: step = star_initial(); break;
                case 3: step = cover_starred(); break;
                case 4: step = prime_zeros(); break;
                case 5: step = augment(); break;
                default: step = shift_costs(); break;
            }
        }
        for (int i = 0; i < n; ++i) {
            row_to_col[i] = -1;
            for (int j = 0; j < n; ++j) if (mark[(size_t)i * n + j] == 1) { row_to_col[i] = j; break; }
        }
    }
};
} // namespace

void munkres_host(const double* cost, int n, int32_t* row_to_col)
{
    Hungarian hg(cost, n);
    hg.solve(row_to_col);
}

extern "C" int32_t pvf_overlap_matrix(const double* a, int32_t na, const double* b, int32_t nb, double ratio, double* out)
{
    API_BEGIN
    overlap_matrix_host(a, na, b, nb, ratio, out);
    API_END
}
// _associate of the reference in one call (tracking.py:136-182): gated overlap matrix, padded to square, cost = max - overlap,
// Munkres, and the pairs whose overlap is positive.  det_of_tracker[t] = matched detection index or -1.
extern "C" int32_t pvf_associate(const double* trackers, int32_t n_trackers, const double* detections, int32_t n_detections, double ratio,
                                 int32_t* det_of_tracker)
{
    API_BEGIN
    PVF_REQUIRE(n_trackers >= 0 && n_detections >= 0 && det_of_tracker, "pvf_associate: bad arguments");
    for (int t = 0; t < n_trackers; ++t) det_of_tracker[t] = -1;
    if (n_trackers < 1 || n_detections < 1) return 0;
    const int n = std::max(n_trackers, n_detections);
    std::vector<double> ov((size_t)n_trackers * n_detections), area((size_t)n * n, 0.0), cost((size_t)n * n);
    overlap_matrix_host(trackers, n_trackers, detections, n_detections, ratio, ov.data());
    double mx = 0.0;                                           // np.max over the padded matrix (it contains zeros)
    for (int t = 0; t < n_trackers; ++t)
        for (int d = 0; d < n_detections; ++d) {
            const double v = ov[(size_t)t * n_detections + d];
            area[(size_t)t * n + d] = v;
            if (v > mx) mx = v;
        }
    for (size_t i = 0; i < cost.size(); ++i) cost[i] = mx - area[i];
    std::vector<int32_t> r2c(n);
    munkres_host(cost.data(), n, r2c.data());
    for (int t = 0; t < n_trackers; ++t) {
        const int d = r2c[t];
        if (d < n_detections && area[(size_t)t * n + d] > 0.0) det_of_tracker[t] = d;
    }
    API_END
}

extern "C" int32_t pvf_munkres(const double* cost, int32_t n, int32_t* row_to_col)
{
    API_BEGIN
    PVF_REQUIRE(n > 0 && cost && row_to_col, "pvf_munkres: bad arguments");
    munkres_host(cost, n, row_to_col);
    API_END
}

// ---- S4 -------------------------------------------------------------------------------------------
extern "C" int32_t pvf_landmarks(pvf_handle h, const pvf_handle* frames, const pvf_rect_i32* boxes, int32_t n, int32_t* pts)
{
    API_BEGIN
    ENTER(c, h);
    if (n == 0) return 0;
    std::vector<Frame> f(n);
    for (int i = 0; i < n; ++i) f[i] = c->frame(frames[i]);
    ert_run(c, f, boxes, n, pts);
    API_END
}

static uint8_t* make_face_chips(Ctx* c, const pvf_handle* frames, const int32_t* pts, int n)
{
    const EmbedModel& e = c->emb;
    PVF_REQUIRE(e.loaded, "embedder not loaded");
    std::vector<ChipJob> jobs(n);
    for (int i = 0; i < n; ++i) {
        ChipDetails d;
        face_chip_details(e, pts + (size_t)i * 68 * 2, &d);
        jobs[i] = chip_plan(c->frame(frames[i]), d);
    }
    const size_t bytes = (size_t)n * e.chip_size * e.chip_size * 3;
    c->s_trk0.ensure(bytes);
    chip_extract_batch(c, jobs, c->s_trk0.as<uint8_t>());
    return c->s_trk0.as<uint8_t>();
}

extern "C" int32_t pvf_face_chips(pvf_handle h, const pvf_handle* frames, const int32_t* pts, int32_t n, uint8_t* chips)
{
    API_BEGIN
    ENTER(c, h);
    if (n == 0) return 0;
    uint8_t* d = make_face_chips(c, frames, pts, n);
    HIP_CHECK(hipMemcpyAsync(chips, d, (size_t)n * 150 * 150 * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    API_END
}

extern "C" int32_t pvf_embed(pvf_handle h, const pvf_handle* frames, const int32_t* pts, int32_t n, float* out)
{
    API_BEGIN
    ENTER(c, h);
    if (n == 0) return 0;
    const int CH = 4096; // faces per chip-extraction round = the network's largest forward (resnet_forward: MAXB)
    for (int i0 = 0; i0 < n; i0 += CH) {
        const int m = std::min(CH, n - i0);
        uint8_t* d = make_face_chips(c, frames + i0, pts + (size_t)i0 * 136, m);
        resnet_forward(c, d, m, out + (size_t)i0 * 128);
    }
    API_END
}

// landmarks and descriptors of n faces in one call: what `extract` does per face (pyannote-face.py:296-297: get_landmarks, then
// get_embedding on the result), for a whole batch, without a trip through the caller between the two (the caller's thread may be busy:
// the tracking state machine of the next shot runs in the same interpreter)
extern "C" int32_t pvf_landmarks_embed(pvf_handle h, const pvf_handle* frames, const pvf_rect_i32* boxes, int32_t n, int32_t* pts, float* out)
{
    API_BEGIN
    ENTER(c, h);
    if (n == 0) return 0;
    PVF_REQUIRE(frames && boxes && pts && out, "pvf_landmarks_embed: bad arguments");
    {
        std::vector<Frame> f(n);
        for (int i = 0; i < n; ++i) f[i] = c->frame(frames[i]);
        ert_run(c, f, boxes, n, pts);
    }
    const int CH = 4096; // faces per chip-extraction round = the network's largest forward (resnet_forward: MAXB)
    for (int i0 = 0; i0 < n; i0 += CH) {
        const int m = std::min(CH, n - i0);
        uint8_t* d = make_face_chips(c, frames + i0, pts + (size_t)i0 * 136, m);
        resnet_forward(c, d, m, out + (size_t)i0 * 128);
    }
    API_END
}

extern "C" int32_t pvf_embed_chips(pvf_handle h, const uint8_t* chips, int32_t n, float* out)
{
    API_BEGIN
    ENTER(c, h);
    if (n == 0) return 0;
    const size_t bytes = (size_t)n * 150 * 150 * 3;
    c->s_trk0.ensure(bytes);
    HIP_CHECK(hipMemcpyAsync(c->s_trk0.p, chips, bytes, hipMemcpyHostToDevice, c->stream));
    resnet_forward(c, c->s_trk0.as<uint8_t>(), n, out);
    API_END
}

extern "C" int32_t pvf_debug_extract_chip(pvf_handle h, pvf_handle frame, const double rect[4], double cs, double sn, int32_t rows,
                                          int32_t cols, uint8_t* out)
{
    API_BEGIN
    ENTER(c, h);
    ChipDetails d{rect[0], rect[1], rect[2], rect[3], cs, sn, rows, cols};
    std::vector<ChipJob> jobs{chip_plan(c->frame(frame), d)};
    const size_t bytes = (size_t)rows * cols * 3;
    c->s_trk0.ensure(bytes);
    chip_extract_batch(c, jobs, c->s_trk0.as<uint8_t>());
    HIP_CHECK(hipMemcpyAsync(out, c->s_trk0.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    API_END
}

// ---- S5 -------------------------------------------------------------------------------------------
static PairInput table_f64(const double* X) { PairInput in; in.X = X; return in; }
static PairOutput host_full(double* D) { PairOutput o; o.out = D; return o; }

extern "C" int32_t pvf_pair_mean_dist(pvf_handle h, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T, double* D)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(X && row_start && D, "pvf_pair_mean_dist: bad arguments");
    pair_mean_dist_dev(c, table_f64(X), N, dim, row_start, T, host_full(D), nullptr, 0, T, 0, true);
    API_END
}

extern "C" int32_t pvf_pair_mean_dist_metric(pvf_handle h, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T, int32_t metric,
                                              double* D)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(X && row_start && D, "pvf_pair_mean_dist_metric: bad arguments");
    pair_mean_dist_dev(c, table_f64(X), N, dim, row_start, T, host_full(D), nullptr, 0, T, metric, true);
    API_END
}

extern "C" int32_t pvf_pair_upper_rows(pvf_handle h, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T,
                                       int32_t track0, int32_t track1, double* D)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(X && row_start && D, "pvf_pair_upper_rows: bad arguments");
    pair_mean_dist_dev(c, table_f64(X), N, dim, row_start, T, host_full(D), nullptr, track0, track1, 0, false);
    API_END
}

// complete rows: the entries below the diagonal of rows [track0, track1) are D[j][i] of the rows j < i, so the upper-triangle rows
// [0, track1) are computed on the device, mirrored there, and the asked-for rows copied out.  That is O(T^2) work for the last share of
// a split (include/pvface.h says so; a split job uses pvf_pair_upper_rows).  The mirror runs over the whole T x T scratch matrix: the
// rows >= track1 were not computed in this call and hold whatever the scratch held, so only the rows copied out below mean anything
// -- the device matrix (dD) is not kept past this call.
extern "C" int32_t pvf_pair_mean_dist_rows(pvf_handle h, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T,
                                           int32_t track0, int32_t track1, double* D)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(X && row_start && D, "pvf_pair_mean_dist_rows: bad arguments");
    if (track1 < 0) track1 = T;
    PVF_REQUIRE(0 <= track0 && track0 <= track1 && track1 <= T, "pvf_pair_mean_dist_rows: bad track range");
    double* dD = nullptr;
    PairOutput none; none.out = nullptr;
    pair_mean_dist_dev(c, table_f64(X), N, dim, row_start, T, none, &dD, 0, track1, 0, false);
    if (track1 > track0) {
        mirror_upper_dev(c, dD, T);
        HIP_CHECK(hipMemcpyAsync(D + (size_t)track0 * T, dD + (size_t)track0 * T, (size_t)(track1 - track0) * T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    API_END
}

static PairInput table_f32(const float* emb, int64_t stride, int32_t n_src, int32_t on_device, const int32_t* order, int32_t decimals)
{
    PairInput in;
    in.emb = emb; in.emb_stride = stride; in.n_src = n_src; in.emb_on_device = on_device != 0; in.order = order; in.decimals = decimals;
    return in;
}

extern "C" int32_t pvf_pair_upper_rows_f32(pvf_handle h, const float* emb, int64_t row_stride_bytes, int32_t n_src, int32_t emb_on_device,
                                           const int32_t* order, int32_t N, int32_t decimals, const int32_t* row_start, int32_t T,
                                           int32_t track0, int32_t track1, double* rows_out, int32_t out_on_device)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(emb && row_start && (rows_out || track1 <= track0), "pvf_pair_upper_rows_f32: bad arguments");
    PairOutput o; o.out = rows_out; o.on_device = out_on_device != 0; o.compact = true;
    pair_mean_dist_dev(c, table_f32(emb, row_stride_bytes, n_src, emb_on_device, order, decimals), N, 128, row_start, T, o, nullptr, track0, track1, 0, false);
    API_END
}

extern "C" int32_t pvf_cluster_dist(pvf_handle h, const double* D, const int32_t* row_start, int32_t T, double threshold, int32_t* labels,
                                    double* merge_log, int32_t* n_merges)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(T > 0 && D && row_start && labels, "pvf_cluster_dist: bad arguments");
    c->s_clu1.ensure((size_t)T * T * sizeof(double) + (size_t)T * 64 + 4096);
    double* dD = c->s_clu1.as<double>();
    HIP_CHECK(hipMemcpyAsync(dD, D, (size_t)T * T * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int n = hac_dev(c, dD, row_start, T, threshold, labels, merge_log);
    if (n_merges) *n_merges = n;
    API_END
}

extern "C" int32_t pvf_cluster_upper(pvf_handle h, const double* U, int32_t on_device, const int32_t* row_start, int32_t T, double threshold,
                                     int32_t* labels, double* merge_log, int32_t* n_merges)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(T > 0 && U && row_start && labels, "pvf_cluster_upper: bad arguments");
    c->s_clu1.ensure((size_t)T * T * sizeof(double) + (size_t)T * 64 + 4096);
    double* dD = c->s_clu1.as<double>();
    HIP_CHECK(hipMemcpyAsync(dD, U, (size_t)T * T * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    mirror_upper_dev(c, dD, T);
    const int n = hac_dev(c, dD, row_start, T, threshold, labels, merge_log);
    if (n_merges) *n_merges = n;
    API_END
}

extern "C" int32_t pvf_cluster_tracks(pvf_handle h, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T,
                                      double threshold, int32_t* labels, double* merge_log, int32_t* n_merges)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(X && row_start && labels, "pvf_cluster_tracks: bad arguments");
    double* dD = nullptr;
    pair_mean_dist_dev(c, table_f64(X), N, dim, row_start, T, PairOutput(), &dD, 0, T, 0, true);
    const int n = hac_dev(c, dD, row_start, T, threshold, labels, merge_log);
    if (n_merges) *n_merges = n;
    API_END
}

extern "C" int32_t pvf_cluster_tracks_f32(pvf_handle h, const float* emb, int64_t row_stride_bytes, int32_t n_src, int32_t emb_on_device,
                                          const int32_t* order, int32_t N, int32_t decimals, const int32_t* row_start, int32_t T, int32_t metric,
                                          double threshold, int32_t* labels, double* merge_log, int32_t* n_merges)
{
    API_BEGIN
    ENTER(c, h);
    PVF_REQUIRE(emb && row_start && labels, "pvf_cluster_tracks_f32: bad arguments");
    double* dD = nullptr;
    pair_mean_dist_dev(c, table_f32(emb, row_stride_bytes, n_src, emb_on_device, order, decimals), N, 128, row_start, T, PairOutput(), &dD, 0, T, metric, true);
    const int n = hac_dev(c, dD, row_start, T, threshold, labels, merge_log);
    if (n_merges) *n_merges = n;
    API_END
}


// ---- text rows of landmarks.txt / embedding.txt (host) ---------------------------------------------------------------------------------
// ref: scripts/pyannote-face.py:299-311  "{t:.3f} {identifier:d}" followed by " {v:.5f}" per value, one line per face.  Python's format
// and C's printf both write the correctly rounded decimal of the double, so the bytes are the same; what this adds is speed (2 million
// values per 1000 frames): a value goes through scaled-integer printing unless it sits within 1e-6 of a rounding tie or is too large
// for that to be decided in double arithmetic, in which case snprintf decides.
static int put_fixed(char* o, double v, int decimals, double scale)
{
    const double s = std::fabs(v) * scale;
    if (!(s < 4.0e9)) return snprintf(o, 40, "%.*f", decimals, v);
    const double fl = std::floor(s);
    const double fr = s - fl;
    if (std::fabs(fr - 0.5) < 1e-6) return snprintf(o, 40, "%.*f", decimals, v);
    uint64_t q = (uint64_t)fl + (fr > 0.5 ? 1u : 0u);
    char tmp[32];
    int n = 0;
    for (int d = 0; d < decimals; ++d) { tmp[n++] = (char)('0' + q % 10); q /= 10; }
    if (decimals > 0) tmp[n++] = '.';
    do { tmp[n++] = (char)('0' + q % 10); q /= 10; } while (q);
    int k = 0;
    if (std::signbit(v)) o[k++] = '-';
    while (n) o[k++] = tmp[--n];
    return k;
}

extern "C" int32_t pvf_format_rows(const double* t, const int64_t* identifier, const double* values, int64_t n_rows, int32_t n_cols,
                                   int32_t decimals, char* out, int64_t cap, int64_t* written)
{
    API_BEGIN
    PVF_REQUIRE(n_rows >= 0 && n_cols >= 0 && decimals >= 0 && decimals <= 9 && written && (n_rows == 0 || (t && identifier && out)) &&
                (values || n_cols == 0 || n_rows == 0), "pvf_format_rows: bad arguments");
    const int64_t per_row = 64 + (int64_t)n_cols * 42;
    PVF_REQUIRE(cap >= n_rows * per_row, "pvf_format_rows: buffer too small (64 + 42 bytes per value and row)");
    double scale = 1.0;
    for (int d = 0; d < decimals; ++d) scale *= 10.0;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, n_rows / 256));
    std::vector<int64_t> len(n_rows, 0);
    // a row's slot holds 42 bytes per value: finite values of 1e25 and more (27+ digits before the point) do not fit and are refused
    // BEFORE anything is written (NaN and +-inf print as 3-4 characters) -- descriptors and timestamps are many orders below that
    auto fits = [](double v) { return !(std::fabs(v) >= 1e25) || std::isinf(v); };
    for (int64_t r = 0; r < n_rows; ++r) {
        PVF_REQUIRE(fits(t[r]), "pvf_format_rows: a timestamp of 1e25 or more does not fit the fixed row layout");
        const double* v = values + (size_t)r * n_cols;
        for (int c = 0; c < n_cols; ++c) PVF_REQUIRE(fits(v[c]), "pvf_format_rows: a value of magnitude 1e25 or more does not fit the fixed row layout");
    }
    auto work = [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            char* o = out + r * per_row;
            int k = put_fixed(o, t[r], 3, 1000.0);
            o[k++] = ' ';
            k += snprintf(o + k, 24, "%lld", (long long)identifier[r]);
            const double* v = values + (size_t)r * n_cols;
            for (int c = 0; c < n_cols; ++c) { o[k++] = ' '; k += put_fixed(o + k, v[c], decimals, scale); }
            o[k++] = '\n';
            len[r] = k;
        }
    };
    if (nt == 1) work(0, n_rows);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(work, n_rows * i / nt, n_rows * (i + 1) / nt);
        for (auto& x : th) x.join();
    }
    int64_t w = 0;                                   // close the gaps between the rows (each was written at its worst-case offset)
    for (int64_t r = 0; r < n_rows; ++r) {
        if (w != r * per_row) memmove(out + w, out + r * per_row, (size_t)len[r]);
        w += len[r];
    }
    *written = w;
    API_END
}


// ref: scripts/pyannote-face.py:307-311 + face/clustering.py:70-75: a descriptor reaches the clustering through embedding.txt, i.e. as the
// float64 value of its 5-decimal text.  In memory: x (float32) -> double -> np.round(x, 5), which numpy evaluates as rint(x * 1e5) / 1e5
// (three correctly rounded double operations; the same three here, on a few threads -- numpy's own np.round spends 10-28 ms on 8000 x 128
// values, at the very end of a run, when the GPU has nothing left to do).
extern "C" int32_t pvf_round_rows(const float* x, int64_t n, int32_t decimals, double* out)
{
    API_BEGIN
    PVF_REQUIRE(n >= 0 && decimals >= 0 && decimals <= 15 && (n == 0 || (x && out)), "pvf_round_rows: bad arguments");
    double scale = 1.0;
    for (int d = 0; d < decimals; ++d) scale *= 10.0;
    auto work = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            double v = (double)x[i];
            v = v * scale;
            // round half to even, like np.rint: below 2^51 adding and subtracting 1.5 * 2^52 IS that rounding (one IEEE addition in the
            // default mode), and it vectorises; larger magnitudes are integers already or go through nearbyint
            const double big = 6755399441055744.0;
            v = (std::fabs(v) < 2251799813685248.0) ? std::copysign((v + big) - big, v) : std::nearbyint(v);      // (copysign: rint keeps -0.0)
            out[i] = v / scale;
        }
    };
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, n / 65536));
    if (nt == 1) work(0, n);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(work, n * i / nt, n * (i + 1) / nt);
        for (auto& t : th) t.join();
    }
    API_END
}

// ---- text rows back into float64 (what np.loadtxt / pandas.read_table return for the files the writers above produce) --------------------
// A token "[-]digits[.digits]" with at most 15 significant digits is mantissa / 10^k with both operands exact in binary64, so ONE IEEE
// division gives the correctly rounded value of the decimal -- the value strtod returns; anything else (exponents, inf, nan, longer digit
// strings) goes through strtod itself.  Rows end at '\n'; every row must hold the same number of values.
extern "C" int32_t pvf_parse_rows(const char* text, int64_t len, double* out, int64_t cap, int64_t* n_rows, int32_t* n_cols)
{
    API_BEGIN
    PVF_REQUIRE(len >= 0 && (len == 0 || text) && cap >= 0 && (cap == 0 || out) && n_rows && n_cols, "pvf_parse_rows: bad arguments");
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    // rows are independent: cut the text at line ends into a few pieces, parse them on threads, then check the shapes and pack
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, len / (1 << 20)));
    std::vector<int64_t> cut(nt + 1, len);
    cut[0] = 0;
    for (int i = 1; i < nt; ++i) {
        int64_t q = len * i / nt;
        while (q < len && text[q] != '\n') ++q;
        cut[i] = std::min(len, q + 1);
    }
    struct Piece { std::vector<double> v; int64_t rows = 0; int cols = -1; bool ragged = false; bool bad = false; };
    std::vector<Piece> pc(nt);
    auto work = [&](int pi) {
        Piece& P = pc[pi];
        const char* s = text + cut[pi];
        const char* const e = text + cut[pi + 1];
        P.v.reserve((size_t)(e - s) / 8 + 16);
        int in_row = 0;
        auto end_row = [&]() {
            if (in_row == 0) return;                               // blank line
            if (P.cols < 0) P.cols = in_row; else if (P.cols != in_row) P.ragged = true;
            ++P.rows; in_row = 0;
        };
        while (s < e) {
            const char c = *s;
            if (c == '\n') { end_row(); ++s; continue; }
            if (c == ' ' || c == '\t' || c == '\r') { ++s; continue; }
            const char* tok = s;
            bool neg = false;
            if (*s == '-' || *s == '+') { neg = (*s == '-'); ++s; }
            uint64_t m = 0; int nd = 0, frac = 0; bool simple = (s < e) && ((*s >= '0' && *s <= '9') || *s == '.'), seen_digit = false;
            while (s < e && *s >= '0' && *s <= '9') { m = m * 10 + (uint64_t)(*s - '0'); nd += (m != 0); ++s; seen_digit = true; }
            if (s < e && *s == '.') {
                ++s;
                while (s < e && *s >= '0' && *s <= '9') { m = m * 10 + (uint64_t)(*s - '0'); nd += (m != 0); ++frac; ++s; seen_digit = true; }
            }
            const bool ends = (s == e) || *s == ' ' || *s == '\n' || *s == '\t' || *s == '\r';
            double val;
            if (simple && seen_digit && ends && nd <= 15 && frac <= 15) {
                val = (double)m / p10[frac];
                if (neg) val = -val;
            } else {
                // the general case: let strtod read a bounded copy of the token
                const char* t2 = tok;
                while (t2 < e && !(*t2 == ' ' || *t2 == '\n' || *t2 == '\t' || *t2 == '\r')) ++t2;
                std::string tmp(tok, t2);
                char* endp = nullptr;
                val = std::strtod(tmp.c_str(), &endp);
                if (tmp.empty() || endp != tmp.c_str() + tmp.size()) { P.bad = true; return; }
                s = t2;
            }
            P.v.push_back(val);
            ++in_row;
        }
        end_row();
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(work, i);
        for (auto& t : th) t.join();
    }
    int64_t rows = 0, total = 0; int cols = -1;
    for (auto& P : pc) {
        PVF_REQUIRE(!P.bad, "pvf_parse_rows: not a number");
        PVF_REQUIRE(!P.ragged, "pvf_parse_rows: rows of different lengths");
        if (P.rows == 0) continue;
        if (cols < 0) cols = P.cols;
        PVF_REQUIRE(cols == P.cols, "pvf_parse_rows: rows of different lengths");
        rows += P.rows; total += (int64_t)P.v.size();
    }
    PVF_REQUIRE(total <= cap, "pvf_parse_rows: output buffer too small");
    int64_t o = 0;
    for (auto& P : pc) { if (!P.v.empty()) memcpy(out + o, P.v.data(), P.v.size() * sizeof(double)); o += (int64_t)P.v.size(); }
    *n_rows = rows; *n_cols = cols < 0 ? 0 : cols;
    API_END
}

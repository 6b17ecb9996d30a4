// pvf_internal.h -- shared declarations of libpvface.so (gfx950 only; no CPU fallback, no oracle code).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/pvface.h"

#define PVF_FHOG_STRIDE 32

struct PvfError : std::runtime_error { using std::runtime_error::runtime_error; };
// a frame produced more raw candidates than the context's candidate slots hold: the entry point enlarges them and runs the call again
struct CandOverflow : PvfError { int needed; explicit CandOverflow(int n) : PvfError("detector: candidate buffer overflow"), needed(n) {} };
// the screening pass (screen.hip) listed more windows than its list holds, or met a feature above the bound its error analysis assumes:
// the call runs again on the dense exact kernel
struct ScreenRetry : PvfError { ScreenRetry() : PvfError("detector: screening pass gave up") {} };

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            char _b[512];                                                                            \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw PvfError(_b);                                                                      \
        }                                                                                            \
    } while (0)

#define PVF_REQUIRE(cond, msg)                     \
    do {                                           \
        if (!(cond)) throw PvfError(std::string(msg)); \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 256;
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// pinned host buffer
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr; cap = 0;
        HIP_CHECK(hipHostMalloc(&p, n + 256, hipHostMallocDefault));
        cap = n + 256;
    }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
    ~HostBuf() { if (p) (void)hipHostFree(p); }
};

// Pinned staging buffers for the small descriptor tables a batched call uploads (tracker jobs, chip plans): a call takes the next
// buffer of the ring, fills it, queues the copy and moves on -- no stream synchronisation just to keep a host vector alive.  A buffer
// is handed out again only after the copy that read it has run (its event).
struct StageRing {
    static constexpr int N = 8;
    HostBuf buf[N];
    hipEvent_t ev[N] = {};
    bool busy[N] = {};
    int next = 0, cur = 0;
    void* take(size_t bytes)
    {
        cur = next;
        next = (next + 1) % N;
        if (busy[cur]) { HIP_CHECK(hipEventSynchronize(ev[cur])); busy[cur] = false; }
        buf[cur].ensure(bytes);
        return buf[cur].p;
    }
    void sent(hipStream_t st)      // the copy out of the buffer taken last has been queued on `st`
    {
        if (!ev[cur]) HIP_CHECK(hipEventCreateWithFlags(&ev[cur], hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev[cur], st));
        busy[cur] = true;
    }
    ~StageRing() { for (auto e : ev) if (e) (void)hipEventDestroy(e); }
};

struct Tensor {
    std::string name;
    int dtype = 0; // 0 f32, 1 i32, 2 f64, 3 u8
    std::vector<int64_t> dims;
    std::vector<uint8_t> data;
    size_t numel() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
    const float* f32() const { return reinterpret_cast<const float*>(data.data()); }
    const int32_t* i32() const { return reinterpret_cast<const int32_t*>(data.data()); }
    const double* f64() const { return reinterpret_cast<const double*>(data.data()); }
};
std::map<std::string, Tensor> pvf_read_container(const char* path);
// `.pvfm` container or dlib `.dat` stream (dlibdat.hip); kind 1 = shape predictor, 2 = embedder
std::map<std::string, Tensor> pvf_read_model(const char* path, int kind);

struct Frame {
    const uint8_t* d = nullptr; // device, HWC RGB contiguous
    int h = 0, w = 0;
    bool owned = false;
    bool pooled = false;        // owned memory that goes back to Ctx::frame_pool on release (every frame the library allocated itself)
    hipEvent_t ready = nullptr; // asynchronous upload still in flight: the compute stream waits for it on first use (Ctx::frame)
};

struct ProfFamily {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0;
    int64_t launches = 0;
};

struct DetectorModel {
    bool loaded = false;
    int n_filters = 0, frows = 0, fcols = 0, cell = 0, padding = 0, win_w = 0, win_h = 0, min_w = 0, min_h = 0, max_levels = 0;
    double nms_iou = 0, nms_covered = 0;
    std::vector<float> thresh;
    float* d_w = nullptr;    // [nf][frows][fcols][32]
    float* d_bmfma4 = nullptr; // B fragments of score_roll_k: [10][12][2][64 lanes][4 k-steps] (3 shifts x 5 filters per 16-column MFMA tile)
    uint16_t* d_bscreen = nullptr;     // f16 B fragments of score_screen_k: [10][12][64 lanes][8] (weights x screen_scale)
    double screen_scale = 1.0;         // the power of two the weights were multiplied by before their conversion to f16
    double screen_bound[8] = {0, 0, 0, 0, 0, 0, 0, 0};        // |screening score / screen_scale - exact chain| <= screen_bound[filter] (screen.hip)
};

struct ShapeModel {
    bool loaded = false;
    int n_cascades = 0, n_trees = 0, n_parts = 0, n_pix = 0, depth = 0;
    float* d_initial = nullptr;
    int32_t* d_anchor = nullptr;
    float* d_deltas = nullptr;
    int32_t* d_idx1 = nullptr;
    int32_t* d_idx2 = nullptr;
    float* d_thresh = nullptr;
    float* d_leaves = nullptr;
};

struct ConvLayer {
    int cin, cout, k, stride, pad;
    float* d_w;     // [cout][K padded to 32], K index = (r*k+s)*cin + c
    float* d_bias;  // [cout]
    float* d_gamma; // [cout]
    float* d_beta;  // [cout]
    float* d_frag = nullptr;   // 32 -> 32 layers: d_w in MFMA fragment order [k-pairs][64 lanes] (resnet.hip: conv_frag_k), made on first use
};

struct EmbedModel {
    bool loaded = false;
    int chip_size = 150;
    double chip_padding = 0.25;
    std::vector<float> mean_shape; // 51*2
    std::vector<ConvLayer> convs;  // 29
    float* d_fc = nullptr;         // [256][128]
    float* d_blob = nullptr;
    float* d_stem = nullptr;       // the first layer's weights in MFMA fragment order [98 k-pairs][64 lanes] (resnet.hip: stem_frag_k)
};

struct TrackerTables {
    bool set = false;
    double* d_mask64 = nullptr;
    double* d_mask_scale = nullptr;
    double* d_tw64 = nullptr;
    double* d_tw32 = nullptr;
    double alpha_pow_m16 = 0, ln_alpha = 0;
};

struct Tracker {
    double* d_state = nullptr; // A[32*64*64*2] B[64*64] As[512*32*2] Bs[32] (+4 unused doubles); allocated by the first start_track
    int* share = nullptr;      // clones: owners of d_state (copy-on-write, dsst.hip own_state); nullptr = sole owner
    double pos[4] = {0, 0, 0, 0};
    double prev_pos[4] = {0, 0, 0, 0};   // position before a deferred update (dsst_update_many mode 1)
    bool started = false;
    bool pending = false;                // deferred update not committed yet
};
constexpr size_t TRK_A = 0;
constexpr size_t TRK_B = TRK_A + (size_t)32 * 64 * 64 * 2;
constexpr size_t TRK_AS = TRK_B + (size_t)64 * 64;
constexpr size_t TRK_BS = TRK_AS + (size_t)512 * 32 * 2;
constexpr size_t TRK_POS = TRK_BS + 32;
constexpr size_t TRK_DOUBLES = TRK_POS + 4;
// translation-window features of one tracker call, compact (fhog.hip fhog1_compact_k): float S[4096], float T[4][4096], uint8 bin[4096]
constexpr size_t TRKF_PLANE = (size_t)64 * 64;
constexpr size_t TRKF_A = 5 * TRKF_PLANE * sizeof(float);
constexpr size_t TRKF_BYTES = TRKF_A + TRKF_PLANE;
#define TRKF_SLOT(q) ((((q) & 511) << 3) | ((q) >> 9))      // where pixel q = 64 y + x sits in each array of the record

// chip extraction plan (host geometry -> device pyramid + bilinear), see chip.hip
struct ChipJob {
    // source frame
    const uint8_t* img; int h, w;
    // bounding box of the needed sub image (inclusive), and number of pyramid_down<2> levels to build (0 = sample the frame)
    int bx0, by0, sw, sh, levels;
    // affine chip -> level image
    double m[4], b[2];
    int rows, cols;
    bool empty;
};
struct ChipDetails { double l, t, r, b, cs, sn; int rows, cols; };

struct Ctx {
    int device = 0;
    // TWO streams (round 4).  `det_stream` runs the detector (pyramid, FHOG, scoring: VALU- and MFMA-bound kernels whose grids fill the
    // chip) and the device resize of --min-size; `stream` runs everything else -- tracker, chips, landmarks, embedding, clustering, shot
    // detection: latency-bound chains the host waits on -- at a higher priority.  The two sides have their own scratch buffers and their
    // own entry-point locks (det_mu / api_mu), so a thread that detects shot k + 1 and a thread that tracks / extracts shot k really
    // run side by side on the device (tools/probes/overlap_probe.py measured what that hides).  Frames are read by both.
    hipStream_t stream = nullptr;
    hipStream_t det_stream = nullptr;
    DetectorModel det;
    ShapeModel shape;
    EmbedModel emb;
    TrackerTables ttab;
    std::unordered_map<uint64_t, Frame> frames;
    std::unordered_map<uint64_t, std::unique_ptr<Tracker>> trackers;
    std::vector<double*> tracker_pool; // freed tracker states for reuse
    std::atomic<uint64_t> next_id{1};
    // Threads.  api_mu serialises the compute entry points of one context (they share the stream and the scratch buffers); frames_mu
    // guards the frame table, the buffer pool and the ingest rings, which a decoder thread fills (pvf_ingest_*, pvf_frame_upload /
    // _release) while another thread runs kernels.  Order: api_mu before frames_mu, never the other way round.
    std::recursive_mutex api_mu;
    std::recursive_mutex det_mu;       // the detector-side entry points (pvf_detect*, pvf_frame_resize); order: det_mu, api_mu, frames_mu
    std::mutex frames_mu;
    std::mutex prof_mu;                // the profiling families and the event pool (ProfScope runs on both sides)
    bool prof_on = false;
    std::map<std::string, ProfFamily> prof;
    std::vector<hipEvent_t> event_pool;
    // scratch (grow only)
    DevBuf s_grad, s_pyr, s_hist, s_norm, s_feat, s_cand, s_misc, s_chip, s_chip_pyr, s_act0, s_act1, s_act2, s_trk0, s_trk1, s_trk2, s_trkfeat, s_clu0, s_clu1;
    HostBuf h_cand, h_misc;
    StageRing stage;
    // det_run_many: alternating candidate buffers / frame-pointer tables / completion events of the two batches in flight
    DevBuf s_cand2[2], s_fptr[2];
    HostBuf h_cand2[2], h_fptr[2];
    hipEvent_t det_ev[2] = {nullptr, nullptr};
    int det_slot = 0;
    int det_cand_cap = 8192;                  // raw candidates per frame the scoring kernel can record (grown on demand, api.hip)
    // screening pass in front of the exact scoring chain (screen.hip; pvf_detector_screening)
    bool det_screen = true, det_screen_suspended = false, screen_attr_set = false;
    int screen_list_cap = 1 << 20;            // (position, filter) pairs a batch may list for exact re-scoring
    DevBuf s_screen;
    int64_t screen_batches = 0, screen_listed = 0, screen_retries = 0;
    double screen_pipe_err = -1;              // measured by screen_probe: worst |pipe - exact| / sum of magnitudes over its K = 3200 accumulations (five cases)
    bool screen_pipe_flushes_subnormals = false;      // screen_probe case 4 (the bound carries the term e_sub either way)
    int n_cu = 256;
    // released frame buffers by size.  A buffer comes back with the event recorded on the compute stream at its release: whoever takes it
    // next orders its first write behind that event (pool_take), so releasing a frame never waits for the kernels that still read it.
    struct PoolBuf { uint8_t* p; hipEvent_t free_after; hipEvent_t free_after_det; };      // (one event per stream that may still read it)
    std::map<size_t, std::deque<PoolBuf>> frame_pool;      // oldest release first: its event is the most likely to have passed
    size_t frame_pool_bytes = 0;
    void* ingest_rings = nullptr;                          // pinned staging rings of this context (ingest.hip)
    struct MlPlanCache* ml_plans = nullptr;   // detector launch plans of this context (detect.hip); freed by ml_plans_free
    std::map<std::vector<int>, std::unique_ptr<DevBuf>> resize_tabs;   // pvf_frame_resize coefficient tables by (in_w, in_h, out_w, out_h)
    const void* feat_ring_owner = nullptr;    // plan whose zero padding ring s_feat currently holds
    uint8_t* d_orient_lut = nullptr; // 511x511 orientation bins (fhog.hip)
    uint8_t* d_wrap_lut = nullptr;       // the same, indexed by the differences mod 512 (2^18 bytes): orientation_lut_wrapped()

    // a copy of the frame record (the table may be re-hashed by another thread as soon as the lock is gone)
    Frame frame(uint64_t id)
    {
        std::lock_guard<std::mutex> lk(frames_mu);
        auto it = frames.find(id);
        if (it == frames.end()) throw PvfError("unknown frame handle");
        Frame& f = it->second;
        if (f.ready) {          // queued by pvf_ingest_submit on the copy stream: order this context's kernels (both streams) behind the copy
            HIP_CHECK(hipStreamWaitEvent(stream, f.ready, 0));
            HIP_CHECK(hipStreamWaitEvent(det_stream, f.ready, 0));
            (void)hipEventDestroy(f.ready);
            f.ready = nullptr;
        }
        return f;
    }
    // A buffer for a new frame: from the pool when one of that size is there.  `writer`: the stream whose work fills the buffer next (its
    // queue waits for the previous readers on the device); nullptr: the host fills it with a blocking copy (the wait happens here,
    // outside frames_mu).
    uint8_t* take_frame_buffer(size_t bytes, hipStream_t writer)
    {
        PoolBuf b{nullptr, nullptr, nullptr};
        {
            std::lock_guard<std::mutex> lk(frames_mu);
            auto& v = frame_pool[bytes];
            if (!v.empty()) { b = v.front(); v.pop_front(); frame_pool_bytes -= bytes; }
        }
        if (!b.p) {
            HIP_CHECK(hipMalloc((void**)&b.p, bytes));
            return b.p;
        }
        for (hipEvent_t e : {b.free_after, b.free_after_det}) {
            if (!e) continue;
            if (writer) HIP_CHECK(hipStreamWaitEvent(writer, e, 0));
            else HIP_CHECK(hipEventSynchronize(e));
            (void)hipEventDestroy(e);
        }
        return b.p;
    }
    // frames_mu held by the caller
    void pool_give(uint8_t* p, size_t bytes)
    {
        hipEvent_t e = nullptr, e2 = nullptr;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(e, stream));
        HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(e2, det_stream));
        frame_pool[bytes].push_back(PoolBuf{p, e, e2});
        frame_pool_bytes += bytes;
    }
    uint64_t add_frame(const Frame& f)
    {
        const uint64_t id = next_id++;
        std::lock_guard<std::mutex> lk(frames_mu);
        frames[id] = f;
        return id;
    }
    Tracker& tracker(uint64_t id)
    {
        auto it = trackers.find(id);
        if (it == trackers.end()) throw PvfError("unknown tracker handle");
        return *it->second;
    }
};

struct ProfScope {
    Ctx* c; ProfFamily* f = nullptr; hipEvent_t a = nullptr, b = nullptr; hipStream_t s = nullptr; bool ranged = false;
    ProfScope(Ctx* ctx, const char* family, hipStream_t on = nullptr);       // on: the stream the family's kernels run on (default: c->stream)
    ~ProfScope();
};

Ctx* pvf_ctx(pvf_handle h);
void pvf_set_error(const char* msg);

// ---- subsystem entry points (C++ side; the extern "C" wrappers live in api.hip) ----
// detector (detect.hip)
struct RawDet { float score; int32_t filter, level, r, c, l, t, rr, b; };
void det_run_batch(Ctx* c, const std::vector<Frame>& frames, int upsample, double adjust,
                   std::vector<std::vector<RawDet>>& raw_sorted);
// raw_sorted[i]: the candidates of frame i in canonical order, or -- with nms -- what survives det_nms (done batch by batch, overlapped)
void det_run_many(Ctx* c, const std::vector<Frame>& frames, int batch, int upsample, double adjust,
                  std::vector<std::vector<RawDet>>& raw_sorted, bool nms = false);
void ml_plans_free(Ctx* c);
void ingest_free_all(Ctx* c);
void det_nms(const DetectorModel& m, const std::vector<RawDet>& sorted, std::vector<RawDet>& out);
void det_pyramid_level(Ctx* c, const Frame& f, int upsample, int level, std::vector<uint8_t>* out, int* h, int* w);
void det_level_features(Ctx* c, const Frame& f, int upsample, int level, std::vector<float>* out, int* fh, int* fw);
void det_pyramid_batch(Ctx* c, const std::vector<Frame>& frames, int upsample);
void fhog_debug(Ctx* c, const uint8_t* himg, int h, int w, int cell, int pad_r, int pad_c, std::vector<float>& out, int* fh, int* fw);
// generic device fhog (cell 4 / 8; stage access): img u8 [n][h][w][3] -> feat [n][fh][fw][32]
void fhog_device(Ctx* c, const uint8_t* d_img, int n, int h, int w, int cell, int pad_r, int pad_c, float* d_feat,
                 DevBuf& hist, DevBuf& norm, size_t img_stride_in = 0);
// the tracker's translation windows: chips u8 [n][64][64][3] -> compact records [n][TRKF_BYTES] (fhog.hip)
void fhog1_compact(Ctx* c, const uint8_t* d_chips, int n, uint8_t* d_rec);
// the tracker's scale samples given as images (stage access): u8 [n][23][23][3] -> feat [n][4][4][32] (dsst.hip scale_fhog_k)
void fhog_scale_chips(Ctx* c, const uint8_t* d_chips, int n, float* d_feat);
void fhog_dims(int ih, int iw, int cell, int pad_r, int pad_c, int* fh, int* fw);
// chips (chip.hip)
ChipJob chip_plan(const Frame& f, const ChipDetails& d);
void chip_extract_batch(Ctx* c, const std::vector<ChipJob>& jobs, uint8_t* d_out /* n*rows*cols*3, same dims */);
void transform_batch(Ctx* c, const std::vector<ChipJob>& jobs, uint8_t* d_out);
// landmarks (ert.hip)
void ert_run(Ctx* c, const std::vector<Frame>& frames, const pvf_rect_i32* boxes, int n, int32_t* pts);
// embedding (resnet.hip)
void face_chip_details(const EmbedModel& m, const int32_t* pts68, ChipDetails* out);
void resnet_forward(Ctx* c, const uint8_t* d_chips, int n, float* h_out);
// tracker (dsst.hip)
void dsst_start_many(Ctx* c, const std::vector<Tracker*>& t, const std::vector<Frame>& f, const double* boxes);
void dsst_clone_many(Ctx* c, const std::vector<Tracker*>& src, const std::vector<Tracker*>& dst);
double* tracker_state_alloc(Ctx* c);
// shot boundary detection (shot.hip)
void shot_dfd(Ctx* c, const std::vector<Frame>& frames, int ow, int oh, const float* tables22, double* dfd, uint8_t* gray_out, float* flow_out);
void dsst_update_many(Ctx* c, const std::vector<Tracker*>& t, const std::vector<Frame>& f, double* psr, double* boxes_out, int mode = 0);
// association (assoc.cpp part of api)
void overlap_matrix_host(const double* a, int na, const double* b, int nb, double ratio, double* out);
void munkres_host(const double* cost, int n, int32_t* row_to_col);
// clustering (cluster.hip)
// the table of a clustering: float64 host rows (the reference's, read from embedding.txt), or float32 rows in host or device memory of
// which row k of the table is np.round(float64(emb[order[k]]), decimals) -- gathered and rounded on the device (decimals < 0: no rounding)
struct PairInput {
    const double* X = nullptr;
    const float* emb = nullptr; int64_t emb_stride = 0; int n_src = 0; const int32_t* order = nullptr; int decimals = -1; bool emb_on_device = false;
};
// where the rows [t0, t1) of D go: nowhere (out == nullptr), into the full T x T matrix `out` addresses, or (compact) into (t1 - t0) x T values
struct PairOutput { double* out = nullptr; bool on_device = false; bool compact = false; };
void pair_mean_dist_dev(Ctx* c, const PairInput& in, int N, int dim, const int32_t* row_start, int T, const PairOutput& out, double** d_D_keep,
                        int t0, int t1, int metric, bool mirror);
void mirror_upper_dev(Ctx* c, double* dD, int T);
int hac_dev(Ctx* c, double* d_D, const int32_t* row_start, int T, double threshold, int32_t* labels, double* merge_log);

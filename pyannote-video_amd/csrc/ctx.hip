// ctx.hip -- context, model containers, frames, HIP-event profiling.
#include "pvf_internal.h"
#include "detect_ml.h"
#include <fstream>
#include <mutex>
#include <dlfcn.h>

static thread_local std::string g_err;
void pvf_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* pvf_last_error(void) { return g_err.c_str(); }

static std::mutex g_ctx_mu;
static std::unordered_map<uint64_t, std::unique_ptr<Ctx>> g_ctxs;
static uint64_t g_next_ctx = 0x1000;

Ctx* pvf_ctx(pvf_handle h)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctxs.find(h);
    if (it == g_ctxs.end()) throw PvfError("unknown context handle");
    return it->second.get();
}

// ---------------------------------------------------------------------------------------------------
std::map<std::string, Tensor> pvf_read_container(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw PvfError(std::string("cannot open model file: ") + path);
    char magic[8];
    f.read(magic, 8);
    if (!f || memcmp(magic, "PVFMODEL", 8) != 0) throw PvfError(std::string("not a PVFMODEL container: ") + path);
    uint32_t ver = 0, n = 0;
    f.read((char*)&ver, 4); f.read((char*)&n, 4);
    std::map<std::string, Tensor> out;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t ln = 0;
        f.read((char*)&ln, 4);
        std::string name(ln, '\0');
        f.read(&name[0], ln);
        uint32_t dt = 0, nd = 0;
        f.read((char*)&dt, 4); f.read((char*)&nd, 4);
        Tensor t;
        t.name = name; t.dtype = (int)dt;
        for (uint32_t d = 0; d < nd; ++d) { uint64_t v; f.read((char*)&v, 8); t.dims.push_back((int64_t)v); }
        uint64_t nbytes = 0;
        f.read((char*)&nbytes, 8);
        std::streamoff pos = f.tellg();
        std::streamoff pad = (8 - (pos % 8)) % 8;
        f.seekg(pad, std::ios::cur);
        t.data.resize(nbytes);
        f.read((char*)t.data.data(), (std::streamsize)nbytes);
        if (!f) throw PvfError(std::string("truncated model file: ") + path);
        out[name] = std::move(t);
    }
    return out;
}

static const Tensor& need(const std::map<std::string, Tensor>& m, const char* k)
{
    auto it = m.find(k);
    if (it == m.end()) throw PvfError(std::string("model file lacks tensor ") + k);
    return it->second;
}

template <class T>
static T* upload(const void* src, size_t count)
{
    T* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, count * sizeof(T)));
    HIP_CHECK(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static void load_detector(Ctx* c, const char* path)
{
    auto m = pvf_read_container(path);
    const Tensor& meta = need(m, "det.meta");
    PVF_REQUIRE(meta.numel() >= 10, "det.meta too short");
    const int32_t* q = meta.i32();
    DetectorModel& d = c->det;
    d.n_filters = q[0]; d.frows = q[1]; d.fcols = q[2]; d.cell = q[3]; d.padding = q[4];
    d.win_w = q[5]; d.win_h = q[6]; d.min_w = q[7]; d.min_h = q[8]; d.max_levels = q[9];
    PVF_REQUIRE(d.n_filters == 5, "detector: dlib's frontal face detector has 5 filters; the scoring kernel packs 3 shifts x 5 filters per MFMA tile");
    PVF_REQUIRE(d.frows == 10 && d.fcols == 10 && d.cell == 8, "detector: 10x10 cells of 8 px supported");
    const Tensor& nms = need(m, "det.nms");
    d.nms_iou = nms.f64()[0]; d.nms_covered = nms.f64()[1];
    const Tensor& w = need(m, "det.w");
    PVF_REQUIRE(w.numel() == (size_t)d.n_filters * d.frows * d.fcols * 32, "det.w shape");
    const Tensor& th = need(m, "det.thresh");
    d.thresh.assign(th.f32(), th.f32() + d.n_filters);
    if (d.d_w) (void)hipFree(d.d_w);
    d.d_w = upload<float>(w.f32(), w.numel());
    {
        // B fragments of score_roll_k.  K of one filter row = the 12 cells x 31 planes a tile of three column shifts spans, taken
        // as ONE run of 372 values (index kk = 31 * cell + plane; no pad plane) = 93 k-steps of 4: lane (column j = lane & 15 = 5 * shift +
        // filter, kq = lane >> 4) holds kk = 4 * step + kq.  Packed four steps per lane: [m][group of 8 steps: 12][half: 2][lane: 64][4];
        // steps 93..95 stay zero and are never issued.
        std::vector<float> b4((size_t)10 * 12 * 2 * 64 * 4, 0.0f);
        for (int mm = 0; mm < 10; ++mm)
            for (int step = 0; step < 93; ++step)
                for (int l = 0; l < 64; ++l) {
                    const int kq = l >> 4, j = l & 15, kk = 4 * step + kq;
                    if (j >= 15) continue;
                    const int np = kk / 31, p = kk % 31;
                    const int sh = j / 5, f = j % 5, n = np - sh;
                    if (n < 0 || n >= 10) continue;
                    const int grp = step >> 3, pq = step & 7;
                    b4[((((size_t)mm * 12 + grp) * 2 + (pq >> 2)) * 64 + l) * 4 + (pq & 3)] = w.f32()[(((size_t)f * 10 + mm) * 10 + n) * 32 + p];
                }
        if (d.d_bmfma4) (void)hipFree(d.d_bmfma4);
        d.d_bmfma4 = upload<float>(b4.data(), b4.size());
    }
    screen_prepare_model(d, w.f32());
    d.loaded = true;
}

static void load_shape(Ctx* c, const char* path)
{
    auto m = pvf_read_model(path, 1);
    const int32_t* q = need(m, "sp.meta").i32();
    ShapeModel& s = c->shape;
    s.n_cascades = q[0]; s.n_trees = q[1]; s.n_parts = q[2]; s.n_pix = q[3]; s.depth = q[4];
    PVF_REQUIRE(s.n_parts == 68, "shape predictor: 68 parts expected");
    PVF_REQUIRE(s.depth >= 1 && s.depth <= 6, "shape predictor: tree depth 1..6");
    PVF_REQUIRE(s.n_pix >= 1 && s.n_pix <= 1024, "shape predictor: 1..1024 feature pixels per cascade");
    PVF_REQUIRE(s.n_cascades >= 1 && s.n_trees >= 1, "shape predictor: no cascades / trees");
    {
        // the kernel indexes with what the file says: sizes and indices are checked here, once (a corrupt or crafted model file must
        // not turn into out-of-range device reads)
        const size_t splits = (size_t)s.n_cascades * s.n_trees * (((size_t)1 << s.depth) - 1), leaves = (size_t)s.n_cascades * s.n_trees * ((size_t)1 << s.depth);
        PVF_REQUIRE(need(m, "sp.initial_shape").numel() == 136 && need(m, "sp.anchor_idx").numel() == (size_t)s.n_cascades * s.n_pix &&
                    need(m, "sp.deltas").numel() == (size_t)s.n_cascades * s.n_pix * 2 && need(m, "sp.split_idx1").numel() == splits &&
                    need(m, "sp.split_idx2").numel() == splits && need(m, "sp.split_thresh").numel() == splits &&
                    need(m, "sp.leaves").numel() == leaves * 136, "shape predictor: tensor sizes do not match sp.meta");
        const int32_t* a = need(m, "sp.anchor_idx").i32();
        for (size_t k = 0; k < (size_t)s.n_cascades * s.n_pix; ++k) PVF_REQUIRE(a[k] >= 0 && a[k] < s.n_parts, "shape predictor: anchor index out of range");
        const int32_t* i1 = need(m, "sp.split_idx1").i32();
        const int32_t* i2 = need(m, "sp.split_idx2").i32();
        for (size_t k = 0; k < splits; ++k)
            PVF_REQUIRE(i1[k] >= 0 && i1[k] < s.n_pix && i2[k] >= 0 && i2[k] < s.n_pix, "shape predictor: split feature index out of range");
    }
    auto up_f = [&](const char* k) { const Tensor& t = need(m, k); return upload<float>(t.f32(), t.numel()); };
    auto up_i = [&](const char* k) { const Tensor& t = need(m, k); return upload<int32_t>(t.i32(), t.numel()); };
    s.d_initial = up_f("sp.initial_shape");
    s.d_anchor = up_i("sp.anchor_idx");
    s.d_deltas = up_f("sp.deltas");
    s.d_idx1 = up_i("sp.split_idx1");
    s.d_idx2 = up_i("sp.split_idx2");
    s.d_thresh = up_f("sp.split_thresh");
    s.d_leaves = up_f("sp.leaves");
    s.loaded = true;
}

static const int UNITS[14][3] = {{32, 32, 0}, {32, 32, 0}, {32, 32, 0}, {32, 64, 1}, {64, 64, 0}, {64, 64, 0}, {64, 64, 0},
                                 {64, 128, 1}, {128, 128, 0}, {128, 128, 0}, {128, 256, 1}, {256, 256, 0}, {256, 256, 0}, {256, 256, 1}};

static void load_embedder(Ctx* c, const char* path)
{
    auto m = pvf_read_model(path, 2);
    EmbedModel& e = c->emb;
    e.chip_size = need(m, "emb.meta").i32()[0];
    e.chip_padding = need(m, "emb.padding").f64()[0];
    PVF_REQUIRE(e.chip_size == 150, "embedder: 150x150 chips expected");
    const Tensor& ms = need(m, "emb.mean_shape");
    PVF_REQUIRE(ms.numel() == 102, "emb.mean_shape must be 51x2");
    e.mean_shape.assign(ms.f32(), ms.f32() + 102);
    const Tensor& blob = need(m, "emb.blob");
    const float* p = blob.f32();
    const float* end = p + blob.numel();
    for (ConvLayer& L : e.convs) if (L.d_frag) { (void)hipFree(L.d_frag); L.d_frag = nullptr; }
    e.convs.clear();
    if (e.d_stem) { (void)hipFree(e.d_stem); e.d_stem = nullptr; }     // (fragment-ordered copy of the first layer's weights: rebuilt on the next forward)
    auto add_conv = [&](int cin, int cout, int k, int stride, int pad) {
        ConvLayer L{cin, cout, k, stride, pad, nullptr, nullptr, nullptr, nullptr, nullptr};
        const size_t nw = (size_t)cout * cin * k * k;
        PVF_REQUIRE(p + nw + 3 * (size_t)cout <= end, "emb.blob too short");
        // [cout][cin][r][s] -> [cout][kk], kk = (r*k+s)*cp + c, rows zero-padded to a multiple of 32 (the conv kernel's K chunk); the
        // 3-channel input layer is stored with a fourth, all-zero channel (cp = 4) so that the kernel stages one pixel tap with one 16-byte
        // load: x + 0 * w is exact, the chain is unchanged
        const int cp = (cin == 3) ? 4 : cin;
        const int K = k * k * cp, Kpad = (K + 31) / 32 * 32;
        std::vector<float> wt((size_t)cout * Kpad, 0.0f);
        for (int o = 0; o < cout; ++o)
            for (int ci = 0; ci < cin; ++ci)
                for (int r = 0; r < k; ++r)
                    for (int s = 0; s < k; ++s) {
                        const int kk = (r * k + s) * cp + ci;
                        wt[(size_t)o * Kpad + kk] = p[(((size_t)o * cin + ci) * k + r) * k + s];
                    }
        L.d_w = upload<float>(wt.data(), wt.size());
        p += nw;
        L.d_bias = upload<float>(p, cout); p += cout;
        L.d_gamma = upload<float>(p, cout); p += cout;
        L.d_beta = upload<float>(p, cout); p += cout;
        e.convs.push_back(L);
    };
    add_conv(3, 32, 7, 2, 0);
    for (int u = 0; u < 14; ++u) {
        const int cin = UNITS[u][0], n = UNITS[u][1], down = UNITS[u][2];
        add_conv(cin, n, 3, down ? 2 : 1, down ? 0 : 1);
        add_conv(n, n, 3, 1, 1);
    }
    PVF_REQUIRE(p + 256 * 128 == end, "emb.blob size mismatch");
    e.d_fc = upload<float>(p, 256 * 128);
    e.loaded = true;
}

// ---------------------------------------------------------------------------------------------------
// roctx ranges around every kernel family (tracing): PVF_ROCTX=1 resolves roctxRangePushA / roctxRangePop from
// librocprofiler-sdk-roctx.so (or roctracer's libroctx64.so) at run time (no link-time dependency) and brackets the host side of each family's
// launches, so that `rocprofv3 --marker-trace` shows "pyramid", "fhog", "score", "dsst", "chip", "ert", "conv", "pdist", "hac", "shot"
// next to the kernels they queue.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("PVF_ROCTX");
        if (!e || strcmp(e, "1") != 0) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);      // (the library rocprofv3 --marker-trace listens to)
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);                  // (roctracer's, for rocprof v1 / v2)
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
}  // namespace

ProfScope::ProfScope(Ctx* ctx, const char* family, hipStream_t on) : c(ctx), s(on ? on : ctx->stream)
{
    if (roctx().push) { roctx().push(family); ranged = true; }
    if (!c->prof_on) return;
    {
        std::lock_guard<std::mutex> lk(c->prof_mu);
        f = &c->prof[family];                          // (std::map: the address survives later insertions)
        auto get = [&]() {
            hipEvent_t e;
            if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
            else HIP_CHECK(hipEventCreate(&e));
            return e;
        };
        a = get(); b = get();
    }
    HIP_CHECK(hipEventRecord(a, s));
}
ProfScope::~ProfScope()
{
    if (ranged) roctx().pop();
    if (!f) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(c->prof_mu);
    f->pending.emplace_back(a, b);
    f->launches += 1;
}

static void prof_drain(Ctx* c)
{
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
    std::lock_guard<std::mutex> lk(c->prof_mu);
    for (auto& kv : c->prof) {
        for (auto& pr : kv.second.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.total_ms += ms;
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

// ---------------------------------------------------------------------------------------------------
#define API_BEGIN try {
#define API_END                                                        \
    return 0;                                                          \
    }                                                                  \
    catch (const std::exception& e) { pvf_set_error(e.what()); return -1; } \
    catch (...) { pvf_set_error("unknown error"); return -2; }

extern "C" int32_t pvf_version(void) { return 500; }      // round 5: pvf_pair_upper_rows (pvf_pair_mean_dist_rows returns complete rows again), pvf_detect_raw_many;      // round 4: two streams per context, upper-triangle pair means, float32 in-memory clustering entries; 410: the detector's screening pass (pvf_detector_screening*)

extern "C" int32_t pvf_device_count(int32_t* n)
{
    API_BEGIN
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) k = 0;
    *n = k;
    API_END
}

extern "C" int32_t pvf_ctx_create_prio(int32_t device, int32_t priority_class, pvf_handle* out)
{
    API_BEGIN
    int k = 0;
    if (hipGetDeviceCount(&k) != hipSuccess || k <= 0) throw PvfError("no HIP device visible: libpvface has no CPU fallback");
    PVF_REQUIRE(device >= 0 && device < k, "device index out of range");
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        throw PvfError(std::string("libpvface is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
    std::unique_ptr<Ctx> c(new Ctx());
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    {
        int lo = 0, hi = 0;   // numerically lower = higher priority
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // the detector's stream at the lowest priority, the latency-bound side above it (class < 0: both at the lowest)
        int prio = priority_class < 0 ? lo : hi;
        if (const char* e = getenv("PVF_MAIN_STREAM_PRIO")) prio = (strcmp(e, "low") == 0) ? lo : hi;     // (measurement switch)
        // (class 2, measurements only: the other way round -- the detector's stream above the main one; tools/probes/coissue_probe.py)
        HIP_CHECK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, priority_class == 2 ? lo : prio));
        HIP_CHECK(hipStreamCreateWithPriority(&c->det_stream, hipStreamNonBlocking, priority_class == 2 ? hi : lo));
    }
    if (const char* e = getenv("PVF_DETECTOR_SCREENING")) c->det_screen = atoi(e) != 0;     // (what pvf_detector_screening sets: 0 = dense scoring only)
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    uint64_t id = g_next_ctx++;
    g_ctxs[id] = std::move(c);
    *out = id;
    API_END
}

extern "C" int32_t pvf_ctx_create(int32_t device, pvf_handle* out) { return pvf_ctx_create_prio(device, 0, out); }

extern "C" int32_t pvf_ctx_destroy(pvf_handle h)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->det_stream);
    for (auto& kv : c->frames) if (kv.second.owned) (void)hipFree((void*)kv.second.d);
    for (auto& kv : c->trackers) {
        Tracker& t = *kv.second;
        if (t.share) { if (--*t.share > 0) continue; delete t.share; t.share = nullptr; }      // clones: the last owner frees
        if (t.d_state) (void)hipFree(t.d_state);
    }
    for (auto p : c->tracker_pool) (void)hipFree(p);
    for (int k = 0; k < 2; ++k) if (c->det_ev[k]) (void)hipEventDestroy(c->det_ev[k]);
    ml_plans_free(c);
    ingest_free_all(c);
    for (auto& kv : c->frame_pool) for (auto q : kv.second) {
        if (q.free_after) (void)hipEventDestroy(q.free_after);
        if (q.free_after_det) (void)hipEventDestroy(q.free_after_det);
        (void)hipFree(q.p);
    }
    if (c->d_orient_lut) (void)hipFree(c->d_orient_lut);
    if (c->d_wrap_lut) (void)hipFree(c->d_wrap_lut);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->det_stream);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctxs.erase(h);
    API_END
}

extern "C" int32_t pvf_sync(pvf_handle h)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> det_lock(c->det_mu);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    HIP_CHECK(hipStreamSynchronize(c->det_stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    API_END
}

extern "C" int32_t pvf_load_detector(pvf_handle h, const char* path)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> det_lock(c->det_mu);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(path != nullptr, "pvf_load_detector: path is NULL (the Python layer passes the packaged default)");
    load_detector(c, path);
    API_END
}
extern "C" int32_t pvf_load_shape_predictor(pvf_handle h, const char* path)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(path != nullptr, "path is NULL");
    load_shape(c, path);
    API_END
}
extern "C" int32_t pvf_load_embedder(pvf_handle h, const char* path)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(path != nullptr, "path is NULL");
    load_embedder(c, path);
    API_END
}

// host only: one tensor of a model file as the loaders see it (`.pvfm` container or dlib `.dat`); out == NULL: size only
extern "C" int32_t pvf_model_tensor(const char* path, int32_t kind, const char* name, void* out, int64_t cap_bytes, int64_t* nbytes)
{
    API_BEGIN
    PVF_REQUIRE(path && name && nbytes, "pvf_model_tensor: bad arguments");
    auto m = pvf_read_model(path, kind);
    const Tensor& t = need(m, name);
    *nbytes = (int64_t)t.data.size();
    if (out) {
        PVF_REQUIRE(cap_bytes >= (int64_t)t.data.size(), "pvf_model_tensor: buffer too small");
        memcpy(out, t.data.data(), t.data.size());
    }
    API_END
}

extern "C" int32_t pvf_set_tracker_tables(pvf_handle h, const double* mask64, const double* mask_scale, const double* tw64,
                                          const double* tw32, double alpha_pow_m16, double ln_alpha)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    HIP_CHECK(hipSetDevice(c->device));
    TrackerTables& t = c->ttab;
    if (!t.d_mask64) {
        HIP_CHECK(hipMalloc((void**)&t.d_mask64, 64 * 64 * 8));
        HIP_CHECK(hipMalloc((void**)&t.d_mask_scale, 32 * 8));
        HIP_CHECK(hipMalloc((void**)&t.d_tw64, 64 * 8));
        HIP_CHECK(hipMalloc((void**)&t.d_tw32, 32 * 8));
    }
    HIP_CHECK(hipMemcpy(t.d_mask64, mask64, 64 * 64 * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t.d_mask_scale, mask_scale, 32 * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t.d_tw64, tw64, 64 * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t.d_tw32, tw32, 32 * 8, hipMemcpyHostToDevice));
    t.alpha_pow_m16 = alpha_pow_m16; t.ln_alpha = ln_alpha;
    t.set = true;
    API_END
}

extern "C" int32_t pvf_frame_upload(pvf_handle h, const uint8_t* rgb, int32_t fh, int32_t fw, int64_t stride, pvf_handle* out)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(rgb && fh > 0 && fw > 0, "pvf_frame_upload: bad frame");
    if (stride == 0) stride = (int64_t)fw * 3;
    PVF_REQUIRE(stride >= (int64_t)fw * 3, "pvf_frame_upload: row stride smaller than a row");
    uint8_t* d = c->take_frame_buffer((size_t)fh * fw * 3, nullptr);
    // a blocking copy outside the compute stream: the frame is complete when the call returns, whatever the context is running
    HIP_CHECK(hipMemcpy2D(d, (size_t)fw * 3, rgb, (size_t)stride, (size_t)fw * 3, fh, hipMemcpyDefault));    // host or device source
    Frame f; f.d = d; f.h = fh; f.w = fw; f.owned = true; f.pooled = true;
    *out = c->add_frame(f);
    API_END
}

extern "C" int32_t pvf_frame_wrap_device(pvf_handle h, const void* dev, int32_t fh, int32_t fw, pvf_handle* out)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    PVF_REQUIRE(dev && fh > 0 && fw > 0, "pvf_frame_wrap_device: bad frame");
    Frame f; f.d = (const uint8_t*)dev; f.h = fh; f.w = fw; f.owned = false;
    *out = c->add_frame(f);
    API_END
}

extern "C" int32_t pvf_frame_device_ptr(pvf_handle h, pvf_handle frame, const void** dev)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    *dev = c->frame(frame).d;
    API_END
}

// frames_mu held.  The buffer goes back to the pool behind an event on the compute stream, so the call never waits for the kernels
// that were queued on this frame; the next user of the buffer does (Ctx::pool_take).  Returns true for memory the caller owns
// (pvf_frame_wrap_device): compute calls no longer end with a stream synchronisation, so kernels that read the frame may still be
// queued, and the caller is free to reuse the memory as soon as the release returns -- the release must wait for both streams, which
// it does AFTER dropping frames_mu (wait_streams_unlocked): the detector thread's frame look-ups, the decoder's uploads and the pool
// are not held up behind a queued detector batch.
static bool release_frame_locked(Ctx* c, pvf_handle frame)
{
    auto it = c->frames.find(frame);
    PVF_REQUIRE(it != c->frames.end(), "unknown frame handle");
    Frame f = it->second;
    c->frames.erase(it);
    if (f.ready) { (void)hipEventSynchronize(f.ready); (void)hipEventDestroy(f.ready); }     // released before any kernel read it: let the upload finish
    if (f.owned) { c->pool_give((uint8_t*)f.d, (size_t)f.h * f.w * 3); return false; }
    return true;
}

// what was queued on either stream up to now (an event each; the frame's table entry is gone, so nothing newer can read it); frames_mu NOT held
static void wait_streams_unlocked(Ctx* c)
{
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st[2] = {c->stream, c->det_stream};
    for (int i = 0; i < 2; ++i) {
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[i], st[i]) != hipSuccess) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            ev[i] = nullptr;
            (void)hipStreamSynchronize(st[i]);
        }
    }
    for (int i = 0; i < 2; ++i)
        if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
}

extern "C" int32_t pvf_frame_release(pvf_handle h, pvf_handle frame)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    bool wait;
    {
        std::lock_guard<std::mutex> lk(c->frames_mu);
        wait = release_frame_locked(c, frame);
    }
    if (wait) wait_streams_unlocked(c);
    API_END
}

extern "C" int32_t pvf_frame_release_many(pvf_handle h, const pvf_handle* frames, int32_t n)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(n >= 0 && (frames || n == 0), "pvf_frame_release_many: bad arguments");
    bool wait = false;
    {
        std::lock_guard<std::mutex> lk(c->frames_mu);
        for (int i = 0; i < n; ++i) wait = release_frame_locked(c, frames[i]) || wait;
    }
    if (wait) wait_streams_unlocked(c);
    API_END
}

extern "C" int32_t pvf_frame_pool_trim(pvf_handle h, int64_t keep_bytes, int64_t* pooled_bytes)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->frames_mu);
    for (auto& kv : c->frame_pool) {
        auto& v = kv.second;
        while (!v.empty() && (int64_t)c->frame_pool_bytes > keep_bytes) {
            Ctx::PoolBuf b = v.back();
            v.pop_back();
            c->frame_pool_bytes -= kv.first;
            if (b.free_after) { (void)hipEventSynchronize(b.free_after); (void)hipEventDestroy(b.free_after); }
            if (b.free_after_det) { (void)hipEventSynchronize(b.free_after_det); (void)hipEventDestroy(b.free_after_det); }
            HIP_CHECK(hipFree(b.p));
        }
    }
    if (pooled_bytes) *pooled_bytes = (int64_t)c->frame_pool_bytes;
    API_END
}

// free / total device memory as the driver sees it (every allocation of the process and of its neighbours on this GPU)
extern "C" int32_t pvf_mem_info(pvf_handle h, int64_t* free_bytes, int64_t* total_bytes)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    API_END
}

extern "C" int32_t pvf_prof_enable(pvf_handle h, int32_t on)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    if (!on && c->prof_on) prof_drain(c);
    c->prof_on = on != 0;
    API_END
}
extern "C" int32_t pvf_prof_reset(pvf_handle h)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    prof_drain(c);
    for (auto& kv : c->prof) { kv.second.total_ms = 0; kv.second.launches = 0; }
    API_END
}
extern "C" int32_t pvf_prof_get(pvf_handle h, const char* family, double* total_ms, int64_t* launches)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> api_lock(c->api_mu);
    prof_drain(c);
    auto it = c->prof.find(family);
    if (it == c->prof.end()) { *total_ms = 0; *launches = 0; }
    else { *total_ms = it->second.total_ms; *launches = it->second.launches; }
    API_END
}

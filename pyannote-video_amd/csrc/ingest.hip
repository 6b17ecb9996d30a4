// ingest.hip -- frame staging: what the reference does per frame in Python -- `proc.stdout.read` into a fresh
// numpy array (video.py:368-401), an optional `cv2.resize` to the detection size (video.py:402-403, tracking.py:389-400), a second
// decode for `extract` (pyannote-face.py:261 vs :287) -- becomes:
//   - a ring of pinned host slots the decoder writes into (pvf_ingest_*); a slot goes to HBM with ONE asynchronous copy on a copy
//     stream of its own, so uploads run ahead of / beside the detector instead of in front of it;
//   - frames that carry a "ready" event: the compute stream waits for a frame's copy the first time a kernel is about to read it
//     (Ctx::frame), not when the copy is queued, so later frames keep streaming in while earlier ones are processed;
//   - the down-scaled detection frames made on the device from the staged frame (pvf_frame_resize: OpenCV's 8-bit bilinear,
//     restated), the full-size frame staying resident for `extract`: one decode, one upload, no host resize.
#include "pvf_internal.h"
#include <cmath>

struct IngestRing {
    int h = 0, w = 0, depth = 0;
    uint8_t* host = nullptr;                 // depth * h * w * 3 bytes, pinned
    hipStream_t copy = nullptr;
    std::vector<hipEvent_t> done;            // last upload of each slot
    std::vector<char> busy;
    int next = 0;
};

static std::unordered_map<uint64_t, std::unique_ptr<IngestRing>>& rings(Ctx* c)
{
    return *reinterpret_cast<std::unordered_map<uint64_t, std::unique_ptr<IngestRing>>*>(c->ingest_rings);
}

void ingest_free_all(Ctx* c)
{
    if (!c->ingest_rings) return;
    auto* m = reinterpret_cast<std::unordered_map<uint64_t, std::unique_ptr<IngestRing>>*>(c->ingest_rings);
    for (auto& kv : *m) {
        IngestRing& r = *kv.second;
        if (r.copy) { (void)hipStreamSynchronize(r.copy); (void)hipStreamDestroy(r.copy); }
        for (auto e : r.done) if (e) (void)hipEventDestroy(e);
        if (r.host) (void)hipHostFree(r.host);
    }
    delete m;
    c->ingest_rings = nullptr;
}

#define API_BEGIN try {
#define API_END                                                        \
    return 0;                                                          \
    }                                                                  \
    catch (const std::exception& e) { pvf_set_error(e.what()); return -1; } \
    catch (...) { pvf_set_error("unknown error"); return -2; }

extern "C" int32_t pvf_ingest_create(pvf_handle h, int32_t fh, int32_t fw, int32_t depth, pvf_handle* ring)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(fh > 0 && fw > 0 && depth > 0 && ring, "pvf_ingest_create: bad arguments");
    std::unique_ptr<IngestRing> r(new IngestRing());
    r->h = fh; r->w = fw; r->depth = depth;
    HIP_CHECK(hipHostMalloc((void**)&r->host, (size_t)depth * fh * fw * 3, hipHostMallocDefault));
    HIP_CHECK(hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking));
    r->done.assign(depth, nullptr);
    r->busy.assign(depth, 0);
    for (int i = 0; i < depth; ++i) HIP_CHECK(hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming));
    const uint64_t id = c->next_id++;
    std::lock_guard<std::mutex> lk(c->frames_mu);
    if (!c->ingest_rings) c->ingest_rings = new std::unordered_map<uint64_t, std::unique_ptr<IngestRing>>();
    rings(c)[id] = std::move(r);
    *ring = id;
    API_END
}

extern "C" int32_t pvf_ingest_destroy(pvf_handle h, pvf_handle ring)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    std::unique_ptr<IngestRing> own;
    {
        std::lock_guard<std::mutex> lk(c->frames_mu);
        PVF_REQUIRE(c->ingest_rings && rings(c).count(ring), "unknown ingest ring");
        own = std::move(rings(c)[ring]);
        rings(c).erase(ring);
    }
    IngestRing& r = *own;
    HIP_CHECK(hipStreamSynchronize(r.copy));
    (void)hipStreamDestroy(r.copy);
    for (auto e : r.done) (void)hipEventDestroy(e);
    (void)hipHostFree(r.host);
    API_END
}

// next slot in ring order; returns once the previous upload from that slot has left the host buffer
extern "C" int32_t pvf_ingest_acquire(pvf_handle h, pvf_handle ring, int32_t* slot, uint8_t** host_rgb)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    IngestRing* rp = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->frames_mu);
        PVF_REQUIRE(c->ingest_rings && rings(c).count(ring) && slot && host_rgb, "pvf_ingest_acquire: bad arguments");
        rp = rings(c)[ring].get();
    }
    IngestRing& r = *rp;                        // one producer per ring: its slot bookkeeping needs no lock
    const int s = r.next;
    r.next = (r.next + 1) % r.depth;
    if (r.busy[s]) { HIP_CHECK(hipEventSynchronize(r.done[s])); r.busy[s] = 0; }
    *slot = s;
    *host_rgb = r.host + (size_t)s * r.h * r.w * 3;
    API_END
}

// block until every upload queued on the ring's copy stream has finished (measurement / shutdown)
extern "C" int32_t pvf_ingest_wait(pvf_handle h, pvf_handle ring)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    hipStream_t copy = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->frames_mu);
        PVF_REQUIRE(c->ingest_rings && rings(c).count(ring), "unknown ingest ring");
        copy = rings(c)[ring]->copy;
    }
    HIP_CHECK(hipStreamSynchronize(copy));
    API_END
}

// queue the upload of a filled slot; the frame handle is valid at once (kernels that read it wait for the copy on the device)
extern "C" int32_t pvf_ingest_submit(pvf_handle h, pvf_handle ring, int32_t slot, pvf_handle* frame)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    HIP_CHECK(hipSetDevice(c->device));
    IngestRing* rp = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->frames_mu);
        PVF_REQUIRE(c->ingest_rings && rings(c).count(ring) && frame, "pvf_ingest_submit: bad arguments");
        rp = rings(c)[ring].get();
    }
    IngestRing& r = *rp;
    PVF_REQUIRE(slot >= 0 && slot < r.depth, "pvf_ingest_submit: slot out of range");
    const size_t bytes = (size_t)r.h * r.w * 3;
    uint8_t* d = c->take_frame_buffer(bytes, r.copy);   // a recycled buffer: the copy waits (on the device) for the kernels that still read it
    HIP_CHECK(hipMemcpyAsync(d, r.host + (size_t)slot * bytes, bytes, hipMemcpyHostToDevice, r.copy));
    HIP_CHECK(hipEventRecord(r.done[slot], r.copy));
    r.busy[slot] = 1;
    Frame f; f.d = d; f.h = r.h; f.w = r.w; f.owned = true; f.pooled = true;
    HIP_CHECK(hipEventCreateWithFlags(&f.ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(f.ready, r.copy));
    *frame = c->add_frame(f);
    API_END
}

// ---------------------------------------------------------------------------------------------------
// cv2.resize(frame, (out_w, out_h)) with the default INTER_LINEAR on 8-bit images (reference video.py:402-403), [EXT] restated from
// OpenCV's resize.cpp: pixel-centre mapping fx = (dx + 0.5) * scale - 0.5, source index clamped to the image, 11-bit coefficients
// (cvRound(f * 2048) as int16), horizontal pass in int32, vertical pass
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
// The coefficient tables are built on the host (a few thousand entries), one lane = one output pixel (3 channels).
struct ResizeTab { std::vector<int32_t> idx; std::vector<int16_t> coef; };
static ResizeTab linear_table(int in, int out)
{
    ResizeTab t;
    t.idx.resize(out); t.coef.resize((size_t)out * 2);
    const double scale = (double)in / out;
    for (int d = 0; d < out; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= in - 1) { f = 0; s = in - 1; }
        t.idx[d] = s;
        auto cv_round = [](float v) { return (int)std::nearbyint(v); };     // round half to even, like cvRound
        t.coef[2 * d] = (int16_t)cv_round((1.f - f) * 2048.f);
        t.coef[2 * d + 1] = (int16_t)cv_round(f * 2048.f);
    }
    return t;
}

__global__ void __launch_bounds__(256) cv_resize_linear_k(const uint8_t* __restrict__ in, int ih, int iw, uint8_t* __restrict__ out, int oh, int ow,
                                                          const int32_t* __restrict__ xi, const int16_t* __restrict__ xc,
                                                          const int32_t* __restrict__ yi, const int16_t* __restrict__ yc)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const int sx = xi[x], sy = yi[y];
    const int sx1 = min(sx + 1, iw - 1), sy1 = min(sy + 1, ih - 1);
    const int a0 = xc[2 * x], a1 = xc[2 * x + 1], b0 = yc[2 * y], b1 = yc[2 * y + 1];
    const uint8_t* r0 = in + (size_t)sy * iw * 3;
    const uint8_t* r1 = in + (size_t)sy1 * iw * 3;
    uint8_t* o = out + ((size_t)y * ow + x) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int S0 = r0[3 * sx + k] * a0 + r0[3 * sx1 + k] * a1;
        const int S1 = r1[3 * sx + k] * a0 + r1[3 * sx1 + k] * a1;
        o[k] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
    }
}

extern "C" int32_t pvf_frame_resize(pvf_handle h, pvf_handle frame, int32_t out_w, int32_t out_h, pvf_handle* out)
{
    API_BEGIN
    Ctx* c = pvf_ctx(h);
    std::lock_guard<std::recursive_mutex> det_lock(c->det_mu);       // the --min-size copies are made where the detector runs
    HIP_CHECK(hipSetDevice(c->device));
    PVF_REQUIRE(out && out_w > 0 && out_h > 0, "pvf_frame_resize: bad arguments");
    const Frame f = c->frame(frame);
    // coefficient tables: built and uploaded once per (source size, target size), in a buffer of their own (--min-size asks for the same
    // resize for every frame of a video)
    const std::vector<int> key{f.w, f.h, out_w, out_h};
    auto it = c->resize_tabs.find(key);
    if (it == c->resize_tabs.end()) {
        const ResizeTab tx = linear_table(f.w, out_w), ty = linear_table(f.h, out_h);
        std::unique_ptr<DevBuf> tab(new DevBuf());
        tab->ensure((size_t)(out_w + out_h) * (4 + 4));
        uint8_t* q = tab->as<uint8_t>();
        HIP_CHECK(hipMemcpy(q, tx.idx.data(), (size_t)out_w * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(q + (size_t)out_w * 4, ty.idx.data(), (size_t)out_h * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(q + (size_t)(out_w + out_h) * 4, tx.coef.data(), (size_t)out_w * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(q + (size_t)(out_w + out_h) * 4 + (size_t)out_w * 4, ty.coef.data(), (size_t)out_h * 4, hipMemcpyHostToDevice));
        it = c->resize_tabs.emplace(key, std::move(tab)).first;
    }
    uint8_t* p = it->second->as<uint8_t>();
    const int32_t* dxi = (const int32_t*)p; const int32_t* dyi = dxi + out_w;
    const int16_t* dxc = (const int16_t*)(dyi + out_h); const int16_t* dyc = dxc + 2 * out_w;
    const size_t bytes = (size_t)out_h * out_w * 3;
    uint8_t* d = c->take_frame_buffer(bytes, c->det_stream);
    hipLaunchKernelGGL(cv_resize_linear_k, dim3((out_w + 255) / 256, out_h), dim3(256), 0, c->det_stream, f.d, f.h, f.w, d, out_h, out_w, dxi, dxc, dyi, dyc);
    HIP_CHECK(hipGetLastError());
    Frame g; g.d = d; g.h = out_h; g.w = out_w; g.owned = true; g.pooled = true;
    // the copy is written on the detector stream; a consumer on the main stream (tracker start / update, landmarks, shot) must come
    // after it whoever calls in whatever order: Ctx::frame() makes both streams wait for `ready` on first use and destroys it
    HIP_CHECK(hipEventCreateWithFlags(&g.ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(g.ready, c->det_stream));
    *out = c->add_frame(g);
    API_END
}

// dist.hip -- libpvface_dist.so: the RCCL all-gather of the multi-GPU path behind a C ABI (include/pvface_dist.h).
// One process per GPU; each communicator owns a HIP stream.  Payloads live in device memory on both sides (the caller's buffers: the
// gathered rows feed the clustering kernels where they land); two collectives per gather: the counts, then the payload in exact sizes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/pvface_dist.h"

namespace {
struct Err : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw Err(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCLC(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw Err(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

struct Comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void* d_send = nullptr; size_t send_cap = 0;
    void* d_recv = nullptr; size_t recv_cap = 0;
    int64_t* h_counts = nullptr;          // pinned, world + 1 entries: the counts land HERE and reach the caller's array only on success
    void grow(void** p, size_t* cap, size_t need)
    {
        if (need <= *cap) return;
        if (*p) HIPC(hipFree(*p));
        *p = nullptr; *cap = 0;
        HIPC(hipMalloc(p, need + need / 4 + 256));
        *cap = need + need / 4 + 256;
    }
};
thread_local std::string g_err;
std::mutex g_mu;
std::unordered_map<uint64_t, std::unique_ptr<Comm>> g_comms;
uint64_t g_next = 0x2000;

Comm* get(pvfd_handle h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_comms.find(h);
    if (it == g_comms.end()) throw Err("unknown communicator handle");
    return it->second.get();
}

// A collective that never completes (a rank that died, went a different way through its shots, or posted other sizes) must not hang the
// job silently: the stream is POLLED -- the communicator's asynchronous error state beside it -- and after PVF_DIST_TIMEOUT_S seconds
// (default 120) the communicator is aborted and the call fails with an error that names this rank and what it was waiting for.
// After an abort the collective's kernel ends, but what was queued BEHIND it on the stream (the copy of the counts, the next call's use of
// d_send / d_recv) still runs: the stream is drained -- bounded, the device may be beyond help -- before the error leaves this library, so
// nothing of this call touches memory after the caller has been told it failed.
void abort_and_drain(Comm* c)
{
    (void)ncclCommAbort(c->comm); c->comm = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(c->stream) == hipErrorNotReady && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 10.0)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
}

void wait_collective(Comm* c, const char* what, const std::string& detail)
{
    double limit = 120.0;
    if (const char* e = getenv("PVF_DIST_TIMEOUT_S")) { const double v = atof(e); if (v > 0) limit = v; }
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) throw Err(std::string(what) + ": " + hipGetErrorString(q));
        ncclResult_t async = ncclSuccess;
        if (ncclCommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            abort_and_drain(c);
            throw Err(std::string(what) + ": rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) + ": communicator error: " + ncclGetErrorString(async) + " (" + detail + ")");
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > limit) {
            abort_and_drain(c);
            throw Err(std::string(what) + ": rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) + " gave up after " + std::to_string((int)waited) +
                      " s (PVF_DIST_TIMEOUT_S): a peer never joined this collective (" + detail + "); the communicator was aborted");
        }
        if (spin < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

void gather_counts(Comm* c, int64_t n_rows, int64_t* counts)
{
    c->grow(&c->d_send, &c->send_cap, sizeof(int64_t));
    c->grow(&c->d_recv, &c->recv_cap, sizeof(int64_t) * c->world);
    HIPC(hipMemcpyAsync(c->d_send, &n_rows, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    NCCLC(ncclAllGather(c->d_send, c->d_recv, 1, ncclInt64, c->comm, c->stream));
    // into the communicator's own pinned buffer: should the collective fail, no copy into the CALLER's array is left on the stream
    if (!c->h_counts) HIPC(hipHostMalloc((void**)&c->h_counts, sizeof(int64_t) * (c->world + 1), hipHostMallocDefault));
    HIPC(hipMemcpyAsync(c->h_counts, c->d_recv, sizeof(int64_t) * c->world, hipMemcpyDeviceToHost, c->stream));
    wait_collective(c, "pvfd_allgather_counts", "all-gather of one int64 per rank: mine = " + std::to_string(n_rows));
    memcpy(counts, c->h_counts, sizeof(int64_t) * c->world);
}
// spins for about `ms` milliseconds (s_memrealtime: the constant 100 MHz reference clock -- s_memtime follows the shader clock): stands
// for a collective whose peer never arrives (pvfd_debug_stall)
__global__ void stall_k(long long ticks)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
} // namespace

#define API_BEGIN try {
#define API_END return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown error"; return -2; }

extern "C" const char* pvfd_last_error(void) { return g_err.c_str(); }

extern "C" int32_t pvfd_unique_id(uint8_t id[PVFD_ID_BYTES])
{
    API_BEGIN
    static_assert(sizeof(ncclUniqueId) <= PVFD_ID_BYTES, "communicator id does not fit");
    ncclUniqueId u;
    NCCLC(ncclGetUniqueId(&u));
    memset(id, 0, PVFD_ID_BYTES);
    memcpy(id, &u, sizeof u);
    API_END
}

extern "C" int32_t pvfd_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t id[PVFD_ID_BYTES], pvfd_handle* out)
{
    API_BEGIN
    if (!out || !id || world < 1 || rank < 0 || rank >= world) throw Err("pvfd_comm_create: bad arguments");
    HIPC(hipSetDevice(device));
    std::unique_ptr<Comm> c(new Comm());
    c->device = device; c->rank = rank; c->world = world;
    HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCLC(ncclCommInitRank(&c->comm, world, u, rank));
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t h = g_next++;
    g_comms[h] = std::move(c);
    *out = h;
    API_END
}

extern "C" int32_t pvfd_comm_destroy(pvfd_handle h)
{
    API_BEGIN
    Comm* c = get(h);
    HIPC(hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    (void)hipStreamDestroy(c->stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_comms.erase(h);
    API_END
}

extern "C" int32_t pvfd_allgather_counts(pvfd_handle h, int64_t n, int64_t* counts)
{
    API_BEGIN
    Comm* c = get(h);
    HIPC(hipSetDevice(c->device));
    if (!counts) throw Err("pvfd_allgather_counts: bad arguments");
    if (!c->comm) throw Err("pvfd_allgather_counts: the communicator was aborted by an earlier failure");
    gather_counts(c, n, counts);
    API_END
}

// All-gather of a different number of bytes per rank, device memory to device memory: one group of ncclBroadcast calls, rank r the root of
// the r-th one, each receiving straight into its place in d_recv -- exact sizes (a padded ncclAllGather would move world x the largest
// share: with the triangle split of the distance matrix the last rank's share is 5x the first one's), no staging copy, no host bounce.
// xGMI is point to point: every rank sends its share to world - 1 peers, so the step costs about (total bytes) / (per-link rate).
extern "C" int32_t pvfd_allgatherv_dev(pvfd_handle h, const void* d_send, const int64_t* nbytes, void* d_recv)
{
    API_BEGIN
    Comm* c = get(h);
    HIPC(hipSetDevice(c->device));
    if (!c->comm) throw Err("pvfd_allgatherv_dev: the communicator was aborted by an earlier failure");
    if (!nbytes || !d_recv) throw Err("pvfd_allgatherv_dev: bad arguments");
    for (int r = 0; r < c->world; ++r) if (nbytes[r] < 0) throw Err("pvfd_allgatherv_dev: negative count");
    if (nbytes[c->rank] > 0 && !d_send) throw Err("pvfd_allgatherv_dev: no send buffer");
    NCCLC(ncclGroupStart());
    size_t off = 0;
    for (int r = 0; r < c->world; ++r) {
        char* dst = (char*)d_recv + off;
        if (nbytes[r] > 0)          // (every rank knows every count: empty shares are skipped by all of them alike)
            NCCLC(ncclBroadcast(r == c->rank ? d_send : (const void*)dst, dst, (size_t)nbytes[r], ncclInt8, r, c->comm, c->stream));
        off += (size_t)nbytes[r];
    }
    NCCLC(ncclGroupEnd());
    std::string detail = "grouped broadcasts, bytes per rank =";
    for (int r = 0; r < c->world; ++r) detail += " " + std::to_string((long long)nbytes[r]);
    wait_collective(c, "pvfd_allgatherv_dev", detail);
    API_END
}

extern "C" int32_t pvfd_debug_stall(pvfd_handle h, int32_t milliseconds)
{
    API_BEGIN
    Comm* c = get(h);
    HIPC(hipSetDevice(c->device));
    if (!c->comm) throw Err("pvfd_debug_stall: the communicator was aborted by an earlier failure");
    if (milliseconds < 0 || milliseconds > 60000) throw Err("pvfd_debug_stall: 0 .. 60000 ms");
    hipLaunchKernelGGL(stall_k, dim3(1), dim3(64), 0, c->stream, (long long)milliseconds * 100000LL);
    HIPC(hipGetLastError());
    wait_collective(c, "pvfd_debug_stall", "a kernel that spins for " + std::to_string(milliseconds) + " ms in place of a collective");
    API_END
}

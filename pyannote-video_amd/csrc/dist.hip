// dist.hip -- libpvface_dist.so: the RCCL all-gather of the multi-GPU path behind a C ABI (include/pvface_dist.h).
// One process per GPU; each communicator owns a HIP stream and grow-only device buffers.  xGMI is point to point, the payload is small
// (a few MB per rank), so the exchange is latency bound: two collectives per gather (row counts, then one padded payload), never per-row sends.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdint>
#include <cstring>
#include <memory>
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

void gather_counts(Comm* c, int64_t n_rows, int64_t* counts)
{
    c->grow(&c->d_send, &c->send_cap, sizeof(int64_t));
    c->grow(&c->d_recv, &c->recv_cap, sizeof(int64_t) * c->world);
    HIPC(hipMemcpyAsync(c->d_send, &n_rows, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    NCCLC(ncclAllGather(c->d_send, c->d_recv, 1, ncclInt64, c->comm, c->stream));
    HIPC(hipMemcpyAsync(counts, c->d_recv, sizeof(int64_t) * c->world, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
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
    (void)hipStreamDestroy(c->stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_comms.erase(h);
    API_END
}

extern "C" int32_t pvfd_max_rows(pvfd_handle h, int64_t n_rows, int64_t* counts, int64_t* total_rows)
{
    API_BEGIN
    Comm* c = get(h);
    HIPC(hipSetDevice(c->device));
    if (!counts || n_rows < 0) throw Err("pvfd_max_rows: bad arguments");
    gather_counts(c, n_rows, counts);
    int64_t tot = 0;
    for (int r = 0; r < c->world; ++r) tot += counts[r];
    if (total_rows) *total_rows = tot;
    API_END
}

extern "C" int32_t pvfd_allgather_rows(pvfd_handle h, const double* rows, int64_t n_rows, int32_t row_doubles, int64_t* counts, double* out,
                                       int64_t out_cap_rows, int64_t* total_rows)
{
    API_BEGIN
    Comm* c = get(h);
    HIPC(hipSetDevice(c->device));
    if (!counts || !out || n_rows < 0 || row_doubles <= 0 || (n_rows > 0 && !rows)) throw Err("pvfd_allgather_rows: bad arguments");
    gather_counts(c, n_rows, counts);
    int64_t tot = 0, cap = 1;
    for (int r = 0; r < c->world; ++r) { tot += counts[r]; if (counts[r] > cap) cap = counts[r]; }
    if (total_rows) *total_rows = tot;
    if (tot > out_cap_rows) throw Err("pvfd_allgather_rows: output buffer too small for the gathered rows");
    const size_t slot = (size_t)cap * row_doubles;          // doubles per rank in the padded payload
    c->grow(&c->d_send, &c->send_cap, slot * sizeof(double));
    c->grow(&c->d_recv, &c->recv_cap, slot * sizeof(double) * c->world);
    HIPC(hipMemsetAsync(c->d_send, 0, slot * sizeof(double), c->stream));
    if (n_rows) HIPC(hipMemcpyAsync(c->d_send, rows, (size_t)n_rows * row_doubles * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLC(ncclAllGather(c->d_send, c->d_recv, slot, ncclDouble, c->comm, c->stream));
    size_t o = 0;
    for (int r = 0; r < c->world; ++r) {
        const size_t n = (size_t)counts[r] * row_doubles;
        if (n) HIPC(hipMemcpyAsync(out + o, (const double*)c->d_recv + (size_t)r * slot, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        o += n;
    }
    HIPC(hipStreamSynchronize(c->stream));
    API_END
}

// agz_comm.hip — the ONE exchange step of a sharded self-play generation behind the C ABI (include/agz.h agz_comm_*):
// an RCCL all-gather, over xGMI, of the packed sample records every rank's engine produced (game-id shards, one engine per GPU, nothing
// exchanged while the games are played).  The reference has no counterpart (single device); its caller is selfplay.jl:34, which pushes
// the generation's samples into ONE PoolSample — here: the gathered records of all ranks, unpacked by agz_unpack_records.
//
// RCCL is bound at run time (dlopen / dlsym): libagz.so has no link-time dependency on it, a single-GPU host never loads it, and a host
// process that already carries an RCCL (PyTorch's bundled librccl.so) is not handed a second copy's symbols by the dynamic linker.
// Search order: AGZ_RCCL_LIB, symbols already global in the process, librccl.so.1 / librccl.so on the loader path, /opt/rocm/lib.
//
// Memory: allocated ONCE, at agz_comm_create — two slots (the exchange of call k overlaps call k + 1) of a send buffer of
// HDR + capacity x rec_bytes bytes and a receive buffer of world x that.  Per rank: 2 (1 + world) (HDR + capacity x rec_bytes) bytes.
// A block starts with a 32-byte header {int64 record count, int64 records per block of this collective, int64 status, 0}: the status is
// the return code of the rank's self-play call (agz_comm_post_status) — a rank whose call failed still enters the collective, with no
// records, and every rank learns it from the gathered headers (nobody is left waiting inside ncclAllGather).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include "../../include/agz.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;                // rccl.h: NCCL_UNIQUE_ID_BYTES = 128
enum { ncclSuccess = 0, ncclUint8 = 1 };                             // rccl.h: ncclResult_t / ncclDataType_t values used here
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool ok = false;
};
thread_local std::string g_comm_error;

Rccl load_rccl() {
    Rccl R;
    const char* names[] = {getenv("AGZ_RCCL_LIB"), nullptr /* the process itself */, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; i < 5 && !R.ok; ++i) {
        if (i == 0 && !names[0]) continue;
        void* lib = i == 1 ? dlopen(nullptr, RTLD_NOW) : dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!lib) continue;
        void* s = dlsym(lib, "ncclAllGather");
        if (!s) { if (i != 1) dlclose(lib); continue; }
        R.lib = lib;
        R.AllGather = (decltype(R.AllGather))s;
        R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))dlsym(lib, "ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(lib, "ncclCommDestroy");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(lib, "ncclGetErrorString");
        R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.GetErrorString;
    }
    if (!R.ok) R.err = "RCCL not found (AGZ_RCCL_LIB, librccl.so.1, /opt/rocm/lib/librccl.so.1)";
    return R;
}
Rccl& rccl() {
    static Rccl R = load_rccl();                                     // (a function-local static: initialised once, also when two threads get here together)
    return R;
}

constexpr size_t HDR = 32;                                           // bytes in front of a rank's records: {count, records per block, status, 0}

}  // namespace

struct agz_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int32_t rec_bytes = 0;
    int64_t capacity = 0;           // records per rank
    size_t block = 0;               // bytes of one rank's block: HDR + capacity * rec_bytes
    uint8_t* send[2] = {nullptr, nullptr};
    uint8_t* recv[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipEvent_t packed[2] = {nullptr, nullptr};   // recorded on the ENGINE's stream behind the pack kernels: the exchange's stream waits for it
    int64_t* hdr_host = nullptr;    // pinned: two slots x 4 words — the header of the slot's block on its way to the device
    int64_t post_status = 0;        // what the next collective carries (agz_comm_post_status)
    int32_t statuses[64];           // of the exchange last waited for
    int64_t sent[2] = {0, 0};       // records per rank of the slot's collective (-1: none in flight)
    size_t stride[2] = {0, 0};      // bytes between two ranks' blocks in the slot's receive buffer
    int64_t tail[2] = {0, 0};       // records per rank of the slot's SECOND collective (a rank produced more than `sent`): behind the blocks, stride tail x rec_bytes
    uint64_t started = 0, waited = 0;
    int last = -1;                  // slot of the last collective waited for (agz_comm_fetch_records reads it)
    int64_t counts[64];
    std::string err;
    int fail(const char* fmt, ...) {
        char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf; return 0;
    }
};

#define CHIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (c)->fail("%s: %s", #call, hipGetErrorString(e_)); return AGZ_ERR_HIP; } } while (0)

extern "C" {

const char* agz_comm_last_error(const agz_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int agz_comm_unique_id(void* id) {
    if (!id) return AGZ_ERR_ARG;
    Rccl& R = rccl();
    if (!R.ok) { g_comm_error = R.err; return AGZ_ERR_UNSUPPORTED; }
    ncclUniqueId u;
    const int rc = R.GetUniqueId(&u);
    if (rc != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + R.GetErrorString(rc); return AGZ_ERR_HIP; }
    memcpy(id, u.internal, AGZ_COMM_ID_BYTES);
    return AGZ_OK;
}

void agz_comm_destroy(agz_comm* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    for (int i = 0; i < 2; ++i) { hipFree(c->send[i]); hipFree(c->recv[i]); if (c->done[i]) hipEventDestroy(c->done[i]); if (c->packed[i]) hipEventDestroy(c->packed[i]); }
    if (c->hdr_host) hipHostFree(c->hdr_host);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int agz_comm_create(agz_engine* h, int rank, int world, const void* id, int64_t capacity_records, agz_comm** out) {
    if (!h || !id || !out || world < 1 || world > 64 || rank < 0 || rank >= world || capacity_records < 1) { g_comm_error = "agz_comm_create: bad arguments"; return AGZ_ERR_ARG; }
    Rccl& R = rccl();
    if (!R.ok) { g_comm_error = R.err; return AGZ_ERR_UNSUPPORTED; }
    agz_game_info info;
    int rc = agz_get_info(h, &info); if (rc) { g_comm_error = "agz_comm_create: no engine info"; return rc; }
    int64_t n0 = 0;
    rc = agz_get_samples_packed(h, nullptr, 0, &n0);               // (a size query: makes the engine's device current)
    if (rc) { g_comm_error = agz_last_error(h); return rc; }
    agz_comm* c = new agz_comm;
    c->rank = rank; c->world = world; c->rec_bytes = info.rec_bytes; c->capacity = capacity_records;
    c->block = HDR + (size_t)capacity_records * (size_t)info.rec_bytes;
    c->sent[0] = c->sent[1] = -1;
    memset(c->statuses, 0, sizeof c->statuses);
    auto bail = [&](int code) { g_comm_error = c->err; agz_comm_destroy(c); return code; };
    if (hipGetDevice(&c->device) != hipSuccess) { c->fail("hipGetDevice failed"); return bail(AGZ_ERR_HIP); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->fail("hipStreamCreate failed"); return bail(AGZ_ERR_HIP); }
    if (hipHostMalloc((void**)&c->hdr_host, 2 * HDR, hipHostMallocDefault) != hipSuccess) { c->hdr_host = nullptr; c->fail("hipHostMalloc failed"); return bail(AGZ_ERR_NOMEM); }
    memset(c->hdr_host, 0, 2 * HDR);
    for (int i = 0; i < 2; ++i) {
        if (hipMalloc((void**)&c->send[i], c->block) != hipSuccess || hipMalloc((void**)&c->recv[i], c->block * (size_t)world) != hipSuccess ||
            hipEventCreateWithFlags(&c->done[i], hipEventBlockingSync | hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->packed[i], hipEventDisableTiming) != hipSuccess) {
            c->fail("agz_comm_create: %zu bytes of exchange buffers per rank (2 slots x (1 + %d ranks) x (32 + %lld records x %d bytes)) do not fit",
                    2 * c->block * (size_t)(1 + world), world, (long long)capacity_records, info.rec_bytes);
            (void)hipGetLastError();
            return bail(AGZ_ERR_NOMEM);
        }
        if (hipMemset(c->send[i], 0, HDR) != hipSuccess) { c->fail("hipMemset failed"); return bail(AGZ_ERR_HIP); }
    }
    ncclUniqueId u; memcpy(u.internal, id, AGZ_COMM_ID_BYTES);
    const int nrc = R.CommInitRank(&c->comm, world, u, rank);
    if (nrc != ncclSuccess) { c->fail("ncclCommInitRank(rank %d of %d): %s", rank, world, R.GetErrorString(nrc)); c->comm = nullptr; return bail(AGZ_ERR_HIP); }
    *out = c;
    return AGZ_OK;
}

int agz_comm_post_status(agz_comm* c, int status) {
    if (!c) return AGZ_ERR_ARG;
    c->post_status = status;
    return AGZ_OK;
}
int agz_comm_get_statuses(agz_comm* c, int32_t* statuses) {
    if (!c || !statuses) return AGZ_ERR_ARG;
    if (c->last < 0) { c->fail("agz_comm_get_statuses: no completed exchange"); return AGZ_ERR_STATE; }
    memcpy(statuses, c->statuses, sizeof(int32_t) * (size_t)c->world);
    return AGZ_OK;
}

// Nothing here waits for the device: the pack kernels are queued on the engine's stream, the exchange's stream waits for them through an
// event, the header travels from pinned memory (the slot's words are not rewritten before the slot's collective has been waited for).
int agz_allgather_samples_start(agz_engine* h, agz_comm* c, int64_t send_records) {
    if (!h || !c) return AGZ_ERR_ARG;
    if (c->started - c->waited >= 2) { c->fail("agz_allgather_samples_start: two collectives are in flight (wait for the older one first)"); return AGZ_ERR_STATE; }
    const int slot = (int)(c->started & 1);
    const int64_t status = c->post_status;
    c->post_status = 0;
    int64_t n = 0;
    int rc = AGZ_OK;
    if (status == 0) {
        rc = agz_get_samples_packed(h, nullptr, 0, &n);             // (size query)
        // a rank whose records do not fit still takes part in the collective, with its true count in the header: every rank's _wait then
        // fails the same way (AGZ_ERR_ARG, "more than the exchange capacity") instead of the others waiting for a rank that has bailed out
        if (!rc && n <= c->capacity) rc = agz_get_samples_packed_async(h, c->send[slot] + HDR, c->capacity, &n);   // queued on the engine's stream
        if (rc) { c->fail("agz_get_samples_packed: %s", agz_last_error(h)); return rc; }
        CHIP(c, hipEventRecord(c->packed[slot], (hipStream_t)agz_stream(h)));
        CHIP(c, hipStreamWaitEvent(c->stream, c->packed[slot], 0));
    }
    int64_t sent = send_records > 0 ? send_records : c->capacity;
    if (sent > c->capacity) sent = c->capacity;
    int64_t* hdr = c->hdr_host + (size_t)slot * (HDR / 8);
    hdr[0] = n; hdr[1] = sent; hdr[2] = status; hdr[3] = 0;
    CHIP(c, hipMemcpyAsync(c->send[slot], hdr, HDR, hipMemcpyHostToDevice, c->stream));
    const size_t bytes = HDR + (size_t)sent * (size_t)c->rec_bytes;
    const int nrc = rccl().AllGather(c->send[slot], c->recv[slot], bytes, ncclUint8, c->comm, c->stream);
    if (nrc != ncclSuccess) { c->fail("ncclAllGather: %s", rccl().GetErrorString(nrc)); return AGZ_ERR_HIP; }
    CHIP(c, hipEventRecord(c->done[slot], c->stream));
    c->sent[slot] = sent; c->stride[slot] = bytes;
    ++c->started;
    return AGZ_OK;
}

int agz_allgather_samples_wait(agz_comm* c, int64_t* counts, int64_t* max_count) {
    if (!c) return AGZ_ERR_ARG;
    if (c->started == c->waited) { c->fail("agz_allgather_samples_wait: no collective in flight"); return AGZ_ERR_STATE; }
    const int slot = (int)(c->waited & 1);
    CHIP(c, hipSetDevice(c->device));
    CHIP(c, hipEventSynchronize(c->done[slot]));
    // the collective is over: the slot is RETIRED whatever the headers say, so that an error below leaves the pipeline in a state the
    // caller can go on from (the next _wait sees the next slot, _start does not report a collective that no longer exists)
    ++c->waited; c->last = slot; c->tail[slot] = 0;
    int64_t mx = 0;
    int rc = AGZ_OK;
    for (int r = 0; r < c->world; ++r) {
        int64_t hdr[4] = {0, 0, 0, 0};
        const hipError_t e = hipMemcpy(hdr, c->recv[slot] + (size_t)r * c->stride[slot], HDR, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->fail("hipMemcpy of rank %d's header: %s", r, hipGetErrorString(e)); c->counts[r] = 0; c->statuses[r] = AGZ_ERR_HIP; rc = AGZ_ERR_HIP; continue; }
        c->counts[r] = hdr[0]; c->statuses[r] = (int32_t)hdr[2];
        if (hdr[1] != c->sent[slot] && rc == AGZ_OK) { c->fail("rank %d sent blocks of %lld records, this rank of %lld: the ranks must agree on send_records", r, (long long)hdr[1], (long long)c->sent[slot]); rc = AGZ_ERR_STATE; }
        if (hdr[0] > mx) mx = hdr[0];
    }
    if (counts) memcpy(counts, c->counts, sizeof(int64_t) * (size_t)c->world);
    if (max_count) *max_count = mx;
    if (rc) return rc;
    if (mx > c->capacity) { c->fail("a rank produced %lld records, more than the exchange capacity %lld (agz_comm_create)", (long long)mx, (long long)c->capacity); return AGZ_ERR_ARG; }
    if (mx > c->sent[slot]) {
        // some rank produced more records than every rank agreed to send (the host's prediction was too small).  Every rank sees the
        // same counts, so all of them take this branch together and gather the rest — blocking, from the send buffer, which still holds
        // the call's records — into the room behind the blocks (world x capacity records fit the receive buffer).  Rare by construction.
        const int64_t tn = mx - c->sent[slot];
        const int nrc = rccl().AllGather(c->send[slot] + HDR + (size_t)c->sent[slot] * (size_t)c->rec_bytes, c->recv[slot] + (size_t)c->world * c->stride[slot],
                                         (size_t)tn * (size_t)c->rec_bytes, ncclUint8, c->comm, c->stream);
        if (nrc != ncclSuccess) { c->fail("ncclAllGather (second step): %s", rccl().GetErrorString(nrc)); return AGZ_ERR_HIP; }
        CHIP(c, hipStreamSynchronize(c->stream));
        c->tail[slot] = tn;
    }
    return AGZ_OK;
}

int agz_comm_fetch_records(agz_comm* c, int rank, void* host_dst, int64_t first, int64_t n) {
    if (!c || rank < 0 || rank >= c->world || (!host_dst && n > 0) || first < 0 || n < 0) return AGZ_ERR_ARG;
    if (c->last < 0) { c->fail("agz_comm_fetch_records: no completed exchange"); return AGZ_ERR_STATE; }
    if (first + n > c->counts[rank]) { c->fail("agz_comm_fetch_records: records [%lld, %lld) of rank %d, which sent %lld", (long long)first, (long long)(first + n), rank, (long long)c->counts[rank]); return AGZ_ERR_ARG; }
    if (n == 0) return AGZ_OK;
    CHIP(c, hipSetDevice(c->device));
    const int slot = c->last;
    const size_t rb = (size_t)c->rec_bytes;
    const int64_t head_n = first < c->sent[slot] ? std::min(n, c->sent[slot] - first) : 0;    // records that travelled in the rank's block
    if (head_n > 0)
        CHIP(c, hipMemcpy(host_dst, c->recv[slot] + (size_t)rank * c->stride[slot] + HDR + (size_t)first * rb, (size_t)head_n * rb, hipMemcpyDeviceToHost));
    if (n > head_n) {                                                // ... and those of the second collective
        const int64_t t0 = first + head_n - c->sent[slot];
        CHIP(c, hipMemcpy((uint8_t*)host_dst + (size_t)head_n * rb,
                          c->recv[slot] + (size_t)c->world * c->stride[slot] + ((size_t)rank * (size_t)c->tail[slot] + (size_t)t0) * rb, (size_t)(n - head_n) * rb, hipMemcpyDeviceToHost));
    }
    return AGZ_OK;
}

const void* agz_comm_records_device(agz_comm* c, int rank) {
    if (!c || rank < 0 || rank >= c->world || c->last < 0 || c->tail[c->last]) return nullptr;   // (contiguous only when one collective carried everything)
    return c->recv[c->last] + (size_t)rank * c->stride[c->last] + HDR;
}

int agz_allgather_samples(agz_engine* h, agz_comm* c, int64_t* counts) {
    // the counts first, then the records padded to the largest count — two collectives, blocking
    if (!h || !c) return AGZ_ERR_ARG;
    if (c->started != c->waited) { c->fail("agz_allgather_samples: a pipelined collective is in flight"); return AGZ_ERR_STATE; }
    const int64_t status = c->post_status;                           // (sent again, with the records, by _start below)
    int64_t n = 0;
    if (status == 0) {
        const int rc = agz_get_samples_packed(h, nullptr, 0, &n);
        if (rc) { c->fail("agz_get_samples_packed: %s", agz_last_error(h)); return rc; }
    }
    CHIP(c, hipSetDevice(c->device));
    const int slot = (int)(c->started & 1);
    // (a) counts: 8 bytes per rank through the head of the buffers (a rank that posted a status sends no records)
    CHIP(c, hipMemcpy(c->send[slot], &n, 8, hipMemcpyHostToDevice));
    int nrc = rccl().AllGather(c->send[slot], c->recv[slot], 8, ncclUint8, c->comm, c->stream);
    if (nrc != ncclSuccess) { c->fail("ncclAllGather (counts): %s", rccl().GetErrorString(nrc)); return AGZ_ERR_HIP; }
    CHIP(c, hipStreamSynchronize(c->stream));
    int64_t all[64], mx = 0;
    CHIP(c, hipMemcpy(all, c->recv[slot], 8 * (size_t)c->world, hipMemcpyDeviceToHost));
    for (int r = 0; r < c->world; ++r) if (all[r] > mx) mx = all[r];
    // (b) the records, padded to the largest count (a count beyond the capacity is reported by _wait, on every rank alike)
    const int rc = agz_allgather_samples_start(h, c, mx > 0 ? std::min(mx, c->capacity) : 1); if (rc) return rc;
    return agz_allgather_samples_wait(c, counts, nullptr);
}

int agz_allgather_samples_status(agz_engine* h, agz_comm* c, int status, int64_t* counts, int32_t* statuses) {
    if (!h || !c) return AGZ_ERR_ARG;
    c->post_status = status;
    const int rc = agz_allgather_samples(h, c, counts);
    if (rc) return rc;
    if (statuses) memcpy(statuses, c->statuses, sizeof(int32_t) * (size_t)c->world);
    return AGZ_OK;
}

}  // extern "C"

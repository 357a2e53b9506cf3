// libzkmi355_rccl.so: the zk_allgather_fn of a sharded proof on RCCL (include/zkmi355_rccl.h).  Links librccl only: the core library never sees it, a host
// that has no torch gets its collective from here.  One communicator per rank; one process per GPU (ncclCommInitRank) or one process driving N devices
// (ncclCommInitAll).  The all-gather carries bytes (ncclChar): 128-byte XYZZ partial points per commitment and the quotient's numerators — what
// north_star calls the "final RCCL all-reduce over xGMI" (EC addition is not an RCCL reduction: every rank sums the gathered points itself).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <new>
#include <thread>
#include <vector>
#include "../../include/zkmi355.h"
#include "../../include/zkmi355_rccl.h"

static_assert(ZK_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the header's id size is RCCL's");

struct zk_rccl_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
    uint32_t world = 1, rank = 0, timeout_ms = 30000;
    uint64_t calls = 0;
    bool broken = false;                                              // a collective failed or timed out: the communicator was aborted, every later call fails at once
    char err[384] = {0};
};

namespace {
thread_local char g_err[384] = "";                                   // failures that have no communicator yet
int fail(zk_rccl_comm* c, int code, const char* fmt, ...) {
    char* dst = c ? c->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 384, fmt, ap);
    va_end(ap);
    return code;
}
// the exception barrier of this library's own extern "C" surface (the idiom of abi_guard.h, without the core library's context type)
int rccl_exception(zk_rccl_comm* c, const char* fn) noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(c, ZK_ERR_LIMIT, "%s: out of host memory (std::bad_alloc)", fn); }
    catch (const std::exception& e) { return fail(c, ZK_ERR_HIP, "%s: unexpected C++ exception: %s", fn, e.what()); }
    catch (...) { return fail(c, ZK_ERR_HIP, "%s: unexpected C++ exception", fn); }
}
#define ZK_ABI_TRY try
#define ZK_ABI_CATCH(comm_) catch (...) { return rccl_exception((comm_), __func__); }
#define ZK_ABI_CATCH_VALUE(comm_, value) catch (...) { (void)rccl_exception((comm_), __func__); return value; }
#define ZK_ABI_CATCH_VOID(comm_) catch (...) { (void)rccl_exception((comm_), __func__); }

int wrap(zk_rccl_comm* c, ncclComm_t comm, int device, uint32_t world, uint32_t rank, uint32_t timeout_ms) {
    c->comm = comm; c->device = device; c->world = world; c->rank = rank; c->timeout_ms = timeout_ms ? timeout_ms : 30000;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        return fail(nullptr, ZK_ERR_HIP, "zk_rccl: no stream on device %d", device);
    return ZK_OK;
}
}  // namespace

extern "C" int zk_rccl_unique_id(void* out) ZK_ABI_TRY {
    if (!out) return fail(nullptr, ZK_ERR_ARG, "zk_rccl_unique_id: null");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, ZK_ERR_COMM, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ZK_OK;
} ZK_ABI_CATCH(nullptr)

extern "C" int zk_rccl_comm_create(uint32_t world, uint32_t rank, const void* unique_id, int device, uint32_t timeout_ms, zk_rccl_comm** comm) ZK_ABI_TRY {
    if (!comm) return fail(nullptr, ZK_ERR_ARG, "zk_rccl_comm_create: null");
    *comm = nullptr;
    if (!unique_id || world == 0 || rank >= world) return fail(nullptr, ZK_ERR_ARG, "zk_rccl_comm_create: rank %u of %u", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess)
        return fail(nullptr, ZK_ERR_NODEV, "zk_rccl_comm_create: device %d of %d", device, ndev);
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t nc = nullptr;
    const ncclResult_t r = ncclCommInitRank(&nc, (int)world, id, (int)rank);
    if (r != ncclSuccess) return fail(nullptr, ZK_ERR_COMM, "ncclCommInitRank(rank %u of %u, device %d): %s", rank, world, device, ncclGetErrorString(r));
    zk_rccl_comm* c = new (std::nothrow) zk_rccl_comm();
    if (!c) { (void)ncclCommAbort(nc); return fail(nullptr, ZK_ERR_LIMIT, "zk_rccl_comm_create: out of host memory"); }
    const int rc = wrap(c, nc, device, world, rank, timeout_ms);
    if (rc) { (void)ncclCommAbort(nc); delete c; return rc; }
    *comm = c;
    return ZK_OK;
} ZK_ABI_CATCH(nullptr)

extern "C" int zk_rccl_comm_init_all(uint32_t ndev, const int* devices, uint32_t timeout_ms, zk_rccl_comm** comms) ZK_ABI_TRY {
    if (!comms || ndev == 0 || ndev > 64) return fail(nullptr, ZK_ERR_ARG, "zk_rccl_comm_init_all: %u devices", ndev);
    for (uint32_t r = 0; r < ndev; r++) comms[r] = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return fail(nullptr, ZK_ERR_NODEV, "zk_rccl_comm_init_all: no GPU");
    std::vector<int> devs(ndev);
    for (uint32_t r = 0; r < ndev; r++) {
        devs[r] = devices ? devices[r] : (int)r;
        if (devs[r] < 0 || devs[r] >= have) return fail(nullptr, ZK_ERR_NODEV, "zk_rccl_comm_init_all: device %d of %d", devs[r], have);
    }
    std::vector<ncclComm_t> nc(ndev, nullptr);
    const ncclResult_t res = ncclCommInitAll(nc.data(), (int)ndev, devs.data());
    if (res != ncclSuccess) return fail(nullptr, ZK_ERR_COMM, "ncclCommInitAll(%u devices): %s", ndev, ncclGetErrorString(res));
    int rc = ZK_OK;
    for (uint32_t r = 0; r < ndev && !rc; r++) {
        comms[r] = new (std::nothrow) zk_rccl_comm();
        rc = comms[r] ? wrap(comms[r], nc[r], devs[r], ndev, r, timeout_ms) : fail(nullptr, ZK_ERR_LIMIT, "zk_rccl_comm_init_all: out of host memory");
        if (!rc) nc[r] = nullptr;                                     // (owned by comms[r] now)
    }
    if (rc) {
        for (uint32_t r = 0; r < ndev; r++) {
            if (nc[r]) (void)ncclCommAbort(nc[r]);
            if (comms[r]) { if (!nc[r] && comms[r]->comm) (void)ncclCommAbort(comms[r]->comm); if (comms[r]->stream) (void)hipStreamDestroy(comms[r]->stream); delete comms[r]; comms[r] = nullptr; }
        }
    }
    return rc;
} ZK_ABI_CATCH(nullptr)

extern "C" int zk_rccl_allgather(void* user, const void* send_dev, void* recv_dev, size_t bytes) ZK_ABI_TRY {
    zk_rccl_comm* c = (zk_rccl_comm*)user;
    if (!c) return 1;
    if (c->broken) return 1;                                          // (err keeps the first failure's text)
    if ((!send_dev || !recv_dev) && bytes) { (void)fail(c, ZK_ERR_ARG, "zk_rccl_allgather: null buffer"); return 1; }
    if (hipSetDevice(c->device) != hipSuccess) { (void)fail(c, ZK_ERR_HIP, "zk_rccl_allgather: hipSetDevice(%d)", c->device); return 1; }
    const ncclResult_t r = ncclAllGather(send_dev, recv_dev, bytes, ncclChar, c->comm, c->stream);
    const char* what = nullptr;
    char text[256];
    if (r != ncclSuccess) { snprintf(text, sizeof text, "ncclAllGather: %s", ncclGetErrorString(r)); what = text; }
    // Wait for the collective, but never for ever: a peer that died leaves this rank's kernel spinning on a flag that nobody will set.  hipStreamQuery polls without
    // blocking; RCCL's asynchronous errors (a failed transport) surface through ncclCommGetAsyncError.
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!what) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { snprintf(text, sizeof text, "hipStreamQuery: %s", hipGetErrorString(q)); what = text; break; }
        ncclResult_t async = ncclSuccess;
        if (ncclCommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            snprintf(text, sizeof text, "asynchronous RCCL error: %s", ncclGetErrorString(async)); what = text; break;
        }
        const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (waited > (long long)c->timeout_ms) { snprintf(text, sizeof text, "all-gather of %zu bytes not complete after %u ms (a rank died or a link hangs)", bytes, c->timeout_ms); what = text; break; }
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));      // (a proof's exchanges finish in tens of microseconds: spin first, then yield the core)
    }
    if (what) {
        (void)fail(c, ZK_ERR_COMM, "zk_rccl_allgather (rank %u of %u, call %llu): %s", c->rank, c->world, (unsigned long long)c->calls, what);
        c->broken = true;
        (void)ncclCommAbort(c->comm);                                 // frees the device side too: nothing of the hung collective keeps running
        c->comm = nullptr;
        return 1;
    }
    c->calls++;
    return 0;
} ZK_ABI_CATCH_VALUE((zk_rccl_comm*)user, 1)

extern "C" uint32_t zk_rccl_comm_world(const zk_rccl_comm* comm) ZK_ABI_TRY { return comm ? comm->world : 0; } ZK_ABI_CATCH_VALUE(nullptr, 0u)
extern "C" uint32_t zk_rccl_comm_rank(const zk_rccl_comm* comm) ZK_ABI_TRY { return comm ? comm->rank : 0; } ZK_ABI_CATCH_VALUE(nullptr, 0u)
extern "C" uint64_t zk_rccl_comm_calls(const zk_rccl_comm* comm) ZK_ABI_TRY { return comm ? comm->calls : 0; } ZK_ABI_CATCH_VALUE(nullptr, 0u)
extern "C" const char* zk_rccl_last_error(const zk_rccl_comm* comm) ZK_ABI_TRY { return comm ? comm->err : g_err; } ZK_ABI_CATCH_VALUE(nullptr, "zkmi355_rccl: internal error")

extern "C" void zk_rccl_comm_destroy(zk_rccl_comm* comm) ZK_ABI_TRY {
    if (!comm) return;
    (void)hipSetDevice(comm->device);
    if (comm->comm) {
        if (comm->stream) (void)hipStreamSynchronize(comm->stream);
        (void)ncclCommDestroy(comm->comm);
    }
    if (comm->stream) (void)hipStreamDestroy(comm->stream);
    delete comm;
} ZK_ABI_CATCH_VOID(nullptr)

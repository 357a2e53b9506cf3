// TEST-ONLY kernel emulator: just enough of the HIP surface to run the product's kernel sources
// on CPU threads (one pthread per work-item of a workgroup, workgroups run one after another).
// Used by tests/csrc/libzkmi355_emu.so for `pytest -m "not gpu"`; never part of the product.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define ZK_KERNEL

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
inline thread_local pthread_barrier_t* emu_barrier = nullptr;
inline thread_local unsigned char* emu_smem = nullptr;
static inline void __syncthreads() { pthread_barrier_wait(emu_barrier); }
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }

template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicMax(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }

// ---- host API shims -----------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipStreamNonBlocking = 1 };
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 256) ? 2 : 0; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return 0; }
static inline hipError_t hipHostUnregister(void*) { return 0; }

// ---- launcher ------------------------------------------------------------------------------
template <class F>
struct EmuJob {
    F* f;
    dim3 grid, block;
    unsigned tid;
    pthread_barrier_t* bar;
    unsigned char* smem;
};
template <class F>
static void* emu_thread(void* arg) {
    EmuJob<F>* j = static_cast<EmuJob<F>*>(arg);
    blockDim = j->block; gridDim = j->grid; emu_barrier = j->bar; emu_smem = j->smem;
    threadIdx = dim3(j->tid % j->block.x, j->tid / j->block.x, 0);
    for (unsigned by = 0; by < j->grid.y; by++)
        for (unsigned bx = 0; bx < j->grid.x; bx++) {
            blockIdx = dim3(bx, by, 0);
            (*j->f)();
            pthread_barrier_wait(j->bar);  // workgroups run one after another (static __shared__ reuse)
        }
    return nullptr;
}
inline std::mutex& emu_launch_mutex() { static std::mutex m; return m; }
template <class F>
static inline void emu_launch(dim3 grid, dim3 block, size_t smem_bytes, F f) {
    unsigned nt = block.x * block.y;
    if (nt == 0 || grid.x == 0 || grid.y == 0) return;
    // `__shared__` is a process-wide static here, so kernels of different contexts must not overlap
    std::lock_guard<std::mutex> serialise(emu_launch_mutex());
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, nt);
    void* smem = nullptr;
    if (posix_memalign(&smem, 256, smem_bytes ? smem_bytes : 256)) abort();
    std::vector<pthread_t> th(nt);
    std::vector<EmuJob<F>> jobs(nt);
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 256 * 1024);
    for (unsigned t = 0; t < nt; t++) {
        jobs[t] = EmuJob<F>{&f, grid, block, t, &bar, static_cast<unsigned char*>(smem)};
        if (pthread_create(&th[t], &at, emu_thread<F>, &jobs[t])) abort();
    }
    for (unsigned t = 0; t < nt; t++) pthread_join(th[t], nullptr);
    pthread_attr_destroy(&at);
    pthread_barrier_destroy(&bar);
    free(smem);
}
#define ZK_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu_launch(dim3(grid), dim3(block), (smem), [&]() { kern(__VA_ARGS__); })
#define ZK_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu_smem)

// TEST-ONLY kernel emulator: just enough of the HIP surface to run the product's kernel sources
// on the CPU: every work-item of a workgroup is a user-level context (ucontext fiber) on the calling OS
// thread, __syncthreads() yields to a round-robin scheduler, workgroups run one after another.
// (One pthread per work-item with pthread barriers spent the whole test time in futex calls.)
// Used by tests/csrc/libzkmi355_emu.so for `pytest -m "not gpu"`; never part of the product.
#pragma once
#include <pthread.h>
#include <ucontext.h>
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

// (the TSan build watches the library's HOST threads — prover.hip's lanes, pools and shared keys: the emulated kernels run one launch at a time under
//  emu_launch_mutex, and instrumenting every load of the unrolled field arithmetic took the compile of msm.hip past half an hour)
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_DEVICE_ATTR __attribute__((no_sanitize("thread")))
#endif
#endif
#ifndef EMU_DEVICE_ATTR
#define EMU_DEVICE_ATTR
#endif
#define __global__ EMU_DEVICE_ATTR
#define __device__ EMU_DEVICE_ATTR
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define ZK_KERNEL EMU_DEVICE_ATTR
#define ZK_WAVES_PER_EU(n)
#define ZK_LAUNCH_BOUNDS(n)

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
inline thread_local unsigned char* emu_smem = nullptr;
// Work-item contexts.  glibc's swapcontext saves and restores the signal mask — two system calls per switch, and a barrier-heavy kernel switches millions of
// times: the test suite spent most of its time in the kernel.  On x86-64 the switch is done here instead (callee-saved registers + stack pointer, System V
// ABI); elsewhere ucontext remains.
// Under the sanitizer builds (`make emu-asan / emu-tsan`) every switch is announced: ASan keeps a shadow ("fake") stack per context and must know the bounds of
// the stack it lands on, TSan keeps a clock per fiber (switches synchronise: work-items of one OS thread never race; what TSan watches is the library's real threads).
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
extern "C" void* __tsan_get_current_fiber(void);
extern "C" void* __tsan_create_fiber(unsigned flags);
extern "C" void __tsan_destroy_fiber(void* fiber);
extern "C" void __tsan_switch_to_fiber(void* fiber, unsigned flags);
#endif
#endif
struct EmuSan {                                    // per context; empty in the plain build
#ifdef EMU_ASAN
    void* fake = nullptr; const void* bottom = nullptr; size_t size = 0;
#endif
#ifdef EMU_TSAN
    void* fiber = nullptr;
#endif
};
#if defined(__x86_64__)
struct EmuCtx { void* sp; EmuSan san; };
extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_ctx_switch,.-emu_ctx_switch
)");
static inline void emu_swap_raw(EmuCtx* from, EmuCtx* to) { emu_ctx_switch(&from->sp, to->sp); }
static inline void emu_make_raw(EmuCtx* c, void* stack, size_t size, void (*entry)()) {
    void** sp = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(stack) + size) & ~(uintptr_t)15);
    *--sp = nullptr;                               // the entry function's (never used) return address: rsp = 8 mod 16 at its first instruction
    *--sp = reinterpret_cast<void*>(entry);        // where the first switch into this context returns to
    for (int i = 0; i < 6; i++) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    c->sp = sp;
}
#else
struct EmuCtx { ucontext_t uc; EmuSan san; };
static inline void emu_swap_raw(EmuCtx* from, EmuCtx* to) { swapcontext(&from->uc, &to->uc); }
static inline void emu_make_raw(EmuCtx* c, void* stack, size_t size, void (*entry)()) {
    getcontext(&c->uc);
    c->uc.uc_stack.ss_sp = stack;
    c->uc.uc_stack.ss_size = size;
    c->uc.uc_link = nullptr;
    makecontext(&c->uc, entry, 0);
}
#endif
// `leaving_for_good`: the context that switches away never runs again (ASan drops its fake stack)
static inline void emu_swap(EmuCtx* from, EmuCtx* to, bool leaving_for_good = false) {
#ifdef EMU_ASAN
    __sanitizer_start_switch_fiber(leaving_for_good ? nullptr : &from->san.fake, to->san.bottom, to->san.size);
#endif
#ifdef EMU_TSAN
    __tsan_switch_to_fiber(to->san.fiber, 0);
#endif
    emu_swap_raw(from, to);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(from->san.fake, nullptr, nullptr);      // back in `from`, resumed by somebody's switch
#endif
    (void)leaving_for_good;
}
static inline void emu_make(EmuCtx* c, void* stack, size_t size, void (*entry)()) {
    emu_make_raw(c, stack, size, entry);
#ifdef EMU_ASAN
    c->san.fake = nullptr; c->san.bottom = stack; c->san.size = size;
#endif
#ifdef EMU_TSAN
    c->san.fiber = __tsan_create_fiber(0);
#endif
}
struct EmuSched {
    EmuCtx main;
    std::vector<EmuCtx> ctx;
    std::vector<char> done;
    std::vector<char> stack_mem;     // nt stacks, grow-only
    unsigned cur = 0;
    void (*run)(void*) = nullptr;
    void* arg = nullptr;
    dim3 grid;
};
inline thread_local EmuSched* emu_sched = nullptr;
// barrier = hand the OS thread back to the scheduler; it resumes this work-item after every other one has run up to its own barrier
static inline void __syncthreads() { EmuSched* s = emu_sched; emu_swap(&s->ctx[s->cur], &s->main); }
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
inline std::atomic<long> emu_live_device_allocs{0};      // blocks handed out by hipMalloc / hipHostMalloc and not yet freed: the leak check of tests/test_abi_no_throw.py
static inline hipError_t hipMalloc(void** p, size_t n) { if (posix_memalign(p, 256, n ? n : 256)) return 2; emu_live_device_allocs++; return 0; }
static inline hipError_t hipFree(void* p) { if (p) emu_live_device_allocs--; free(p); return 0; }
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
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { if (posix_memalign(p, 4096, n ? n : 4096)) return 2; emu_live_device_allocs++; return 0; }
static inline hipError_t hipHostFree(void* p) { if (p) emu_live_device_allocs--; free(p); return 0; }

// ---- launcher ------------------------------------------------------------------------------
static void emu_fiber_main() {
    EmuSched* s = emu_sched;
    const unsigned me = s->cur;
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &s->main.san.bottom, &s->main.san.size);      // first time on this stack; the stack we came from is the launching thread's
#endif
    for (unsigned by = 0; by < s->grid.y; by++)
        for (unsigned bx = 0; bx < s->grid.x; bx++) {
            blockIdx = dim3(bx, by, 0);
            s->run(s->arg);
            __syncthreads();                      // workgroups run one after another (static __shared__ reuse)
        }
    s->done[me] = 1;
    emu_swap(&s->ctx[me], &s->main, true);        // never resumed
}
inline std::mutex& emu_launch_mutex() { static std::mutex m; return m; }
template <class F>
static inline void emu_launch(dim3 grid, dim3 block, size_t smem_bytes, F f) {
    const unsigned nt = block.x * block.y;
    if (nt == 0 || grid.x == 0 || grid.y == 0) return;
    // `__shared__` is a process-wide static here, so kernels of different contexts must not overlap
    std::lock_guard<std::mutex> serialise(emu_launch_mutex());
    static thread_local EmuSched sched;
    EmuSched* s = &sched;
    constexpr size_t STACK = 256 * 1024;
    if (s->stack_mem.size() < (size_t)nt * STACK) s->stack_mem.resize((size_t)nt * STACK);
    s->ctx.resize(nt);
    s->done.assign(nt, 0);
    s->run = [](void* c) { (*static_cast<F*>(c))(); };
    s->arg = &f;
    s->grid = grid;
    static thread_local void* smem = nullptr;      // dynamic LDS: one grow-only buffer per launching thread (an allocation per launch was an mmap / munmap pair)
    static thread_local size_t smem_cap = 0;
    if (smem_bytes > smem_cap || !smem) {
        free(smem);
        smem_cap = smem_bytes > 4096 ? smem_bytes : 4096;
        if (posix_memalign(&smem, 256, smem_cap)) abort();
    }
    emu_sched = s;
    emu_smem = static_cast<unsigned char*>(smem);
    blockDim = block; gridDim = grid;
    for (unsigned t = 0; t < nt; t++) emu_make(&s->ctx[t], s->stack_mem.data() + (size_t)t * STACK, STACK, emu_fiber_main);
#ifdef EMU_TSAN
    s->main.san.fiber = __tsan_get_current_fiber();
#endif
    unsigned remaining = nt;
    while (remaining) {
        for (unsigned t = 0; t < nt; t++) {
            if (s->done[t]) continue;
            s->cur = t;
            threadIdx = dim3(t % block.x, t / block.x, 0);
            emu_swap(&s->main, &s->ctx[t]);
            if (s->done[t]) remaining--;
        }
    }
#ifdef EMU_TSAN
    for (unsigned t = 0; t < nt; t++) __tsan_destroy_fiber(s->ctx[t].san.fiber);
#endif
    emu_sched = nullptr;
}
#define ZK_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu_launch(dim3(grid), dim3(block), (smem), [&]() { kern(__VA_ARGS__); })
#define ZK_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu_smem)

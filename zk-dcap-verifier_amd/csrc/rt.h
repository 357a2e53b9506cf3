// Runtime glue.  The product build (default) is plain HIP for gfx950.
//
// With -DZK_EMU the SAME kernel sources are compiled as host C++ and every workgroup is run on
// the CPU, one user-level context per work-item (emu_rt.h: a header only the TEST build has on its include path — `make emu`).  That build exists only so the kernels' index logic
// (counting sort, sub-bucket splitting, tree rounds, NTT addressing ...) can be exercised by
// `pytest -m "not gpu"` in a container without a GPU.  It is test infrastructure: the product
// library libzkmi355.so is never built with ZK_EMU and has no CPU path.
#pragma once
#ifdef ZK_EMU
#include "emu_rt.h"
#else
#include <hip/hip_runtime.h>
#define ZK_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (smem), (stream), __VA_ARGS__)
// dynamic LDS, 16-byte aligned base (guide G17)
#define ZK_DYN_SHARED(type, name)                                              \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#define ZK_KERNEL __global__
#define ZK_LAUNCH_BOUNDS(n) __launch_bounds__(n)
// ask the register allocator for at least n waves per SIMD (VGPR budget 512 / n): a hint that costs spills when it cannot be met — check ScratchSize
#define ZK_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif

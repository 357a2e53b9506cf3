// Instruction-rate and primitive microbenchmarks for gfx950 (run on the GPU box):
//   integer multiply flavours, f64 FMA, the Montgomery product and the XYZZ mixed addition.
// These set the INTEGER roofline quoted in DESIGN.md (the MSM/NTT kernels are bound by
// v_mad_u64_u32 issue, not by HBM).   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../zk-dcap-verifier_amd/csrc/ec.cuh"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 8192;

__global__ void k_mad64(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = blockIdx.x * 3 + 7;
    uint64_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b;
    for (int i = 0; i < ITER; i++) {
        x0 = (uint64_t)(uint32_t)x0 * a + x0; x1 = (uint64_t)(uint32_t)x1 * b + x1;
        x2 = (uint64_t)(uint32_t)x2 * a + x2; x3 = (uint64_t)(uint32_t)x3 * b + x3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 ^ x1 ^ x2 ^ x3) ^ (uint32_t)((x0 ^ x1 ^ x2 ^ x3) >> 32);
}
__global__ void k_mullo(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed | 1, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = x0 * a + 1; x1 = x1 * a + 1; x2 = x2 * a + 1; x3 = x3 * a + 1; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ void k_mulhi(uint32_t* out, uint32_t seed) {
    uint32_t a = (threadIdx.x + seed) | 0x80000001u, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = __umulhi(x0, a) + a; x1 = __umulhi(x1, a) + a; x2 = __umulhi(x2, a) + a; x3 = __umulhi(x3, a) + a; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ void k_mul24(uint32_t* out, uint32_t seed) {
    uint32_t a = (threadIdx.x + seed) & 0xffffff, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) {
        x0 = __umul24(x0, a) + 1; x1 = __umul24(x1, a) + 1; x2 = __umul24(x2, a) + 1; x3 = __umul24(x3, a) + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ void k_fma64(uint32_t* out, uint32_t seed) {
    double a = 1.0 + 1e-9 * (threadIdx.x + seed), x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = fma(x0, a, 0.5); x1 = fma(x1, a, 0.5); x2 = fma(x2, a, 0.5); x3 = fma(x3, a, 0.5); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3);
}
__global__ void k_fma32(uint32_t* out, uint32_t seed) {
    float a = 1.0f + 1e-6f * (threadIdx.x + seed), x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = fmaf(x0, a, 0.5f); x1 = fmaf(x1, a, 0.5f); x2 = fmaf(x2, a, 0.5f); x3 = fmaf(x3, a, 0.5f); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3);
}
__global__ void k_add32(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = (x0 + a) ^ x1; x1 = (x1 + a) ^ x2; x2 = (x2 + a) ^ x3; x3 = (x3 + a) ^ x0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ void k_fqmul(uint32_t* out, uint32_t seed) {
    u256 x = Fq::one(), y = Fq::R2();
    x.v[0] ^= threadIdx.x + seed; y.v[1] ^= blockIdx.x;
    x = Fq::reduce_once(x);
    for (int i = 0; i < ITER / 4; i++) { x = Fq::mul(x, y); y = Fq::mul(y, x); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.v[0] ^ y.v[3];
}
__global__ void k_fqsqr(uint32_t* out, uint32_t seed) {
    u256 x = Fq::one(), y = Fq::R2();
    x.v[0] ^= threadIdx.x + seed; y.v[1] ^= blockIdx.x;
    x = Fq::reduce_once(x);
    for (int i = 0; i < ITER / 4; i++) { x = Fq::sqr(x); y = Fq::sqr(y); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.v[0] ^ y.v[3];
}
__global__ void k_fqmul_cios(uint32_t* out, uint32_t seed) {
    u256 x = Fq::one(), y = Fq::R2();
    x.v[0] ^= threadIdx.x + seed; y.v[1] ^= blockIdx.x;
    x = Fq::reduce_once(x);
    for (int i = 0; i < ITER / 4; i++) { x = Fq::mul_cios(x, y); y = Fq::mul_cios(y, x); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.v[0] ^ y.v[3];
}
__global__ void k_fqadd(uint32_t* out, uint32_t seed) {
    u256 x = Fq::one(), y = Fq::R2();
    x.v[0] ^= threadIdx.x + seed; y.v[1] ^= blockIdx.x;
    for (int i = 0; i < ITER; i++) { x = Fq::add(x, y); y = Fq::sub(y, x); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.v[0] ^ y.v[3];
}
__global__ void k_madd(uint32_t* out, uint32_t seed) {
    Affine g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
    XYZZ acc = xyzz_mdbl(g.x, g.y);
    u256 t = Fq::one(); t.v[0] ^= (threadIdx.x + seed) & 0xff;  // perturb accumulators so lanes differ
    acc.x = Fq::mul(acc.x, t);
    for (int i = 0; i < ITER / 8; i++) xyzz_madd(acc, g.x, g.y);   // not on-curve after the perturbation; timing only
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x.v[0] ^ acc.zzz.v[2];
}

template <class K>
static double run(const char* name, K kern, double ops_per_thread, int block, int blocks_per_cu) {
    int grid = 256 * blocks_per_cu;
    uint32_t* d; CK(hipMalloc(&d, (size_t)grid * block * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d, 1u); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(a)); hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d, (uint32_t)r); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    double rate = ops_per_thread * grid * block / (best * 1e-3);
    printf("%-10s block=%4d x%2d/CU  %8.3f ms  %10.3f Gop/s   (%.2f op/clk/CU @2.4GHz)\n", name, block, blocks_per_cu, best, rate / 1e9, rate / 256 / 2.4e9);
    CK(hipFree(d));
    return rate;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    for (int bpc : {4, 8}) {
        run("mad_u64_u32", k_mad64, 4.0 * ITER, 256, bpc);
        run("mul_lo_u32", k_mullo, 4.0 * ITER, 256, bpc);
        run("mul_hi_u32", k_mulhi, 4.0 * ITER, 256, bpc);
        run("mul_u24", k_mul24, 4.0 * ITER, 256, bpc);
        run("fma_f64", k_fma64, 4.0 * ITER, 256, bpc);
        run("fma_f32", k_fma32, 4.0 * ITER, 256, bpc);
        run("add_xor32", k_add32, 8.0 * ITER, 256, bpc);
    }
    for (int bpc : {2, 4, 8}) {
        run("fq_mul", k_fqmul, 2.0 * (ITER / 4), 256, bpc);
        run("fq_sqr", k_fqsqr, 2.0 * (ITER / 4), 256, bpc);
        run("fq_mul_cios", k_fqmul_cios, 2.0 * (ITER / 4), 256, bpc);
        run("fq_addsub", k_fqadd, 2.0 * ITER, 256, bpc);
    }
    for (int bpc : {1, 2, 4}) run("xyzz_madd", k_madd, 1.0 * (ITER / 8), 256, bpc);
    return 0;
}

// Instruction-rate and primitive microbenchmarks for gfx950 (run on the GPU box):
//   integer multiply flavours, f64 FMA, the Montgomery product and the XYZZ mixed addition.
// These set the INTEGER roofline quoted in DESIGN.md (the MSM/NTT kernels are bound by
// v_mad_u64_u32 issue, not by HBM).   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
// the same instruction with 12 independent accumulators per lane and no carry-out consumer: its ISSUE rate (the 4-chain loop above is latency-limited at low occupancy)
__global__ void k_mad64_x12(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = blockIdx.x * 3 + 7;
    uint64_t x[12];
    for (int j = 0; j < 12; j++) x[j] = a + j * b;
    for (int i = 0; i < ITER / 3; i++) {
#pragma unroll
        for (int j = 0; j < 12; j++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[j]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t r = 0;
    for (int j = 0; j < 12; j++) r ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
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
// the plain-C coarsely integrated operand scanning form of the Montgomery product (round 1's first version; lives here since round 3 — the library has one product)
__host__ __device__ __forceinline__ u256 fq_mul_cios(const u256& a, const u256& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        uint32_t t8 = t[8] + (uint32_t)c;
        uint32_t m = t[0] * FqParams::INV;
        c = ((uint64_t)m * Fq::p(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            uint64_t s = (uint64_t)m * Fq::p(j) + t[j] + c;
            t[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        uint64_t s = (uint64_t)t8 + c;
        t[7] = (uint32_t)s;
        t[8] = (uint32_t)(s >> 32);
    }
    u256 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = t[i];
    return Fq::reduce_once(o);
}

// ---- round 3 experiment: carry-free limbs.  9 limbs of 29 bits, Montgomery radix 2^261, product scanning with ONE 64-bit accumulator per column:
// `v_mad_u64_u32` accumulates with no carry-out and no `v_addc_co_u32`: 81 + 81 mads + 9 mul_lo + 16 64-bit shifts + 17 masks, against 128 mads + 128 addc + 8 mul_lo of
// the 8 x 32-bit form.  GO (profiles/r03/run84, run85: 179 vs 142 G products/s): the arithmetic now lives in csrc/field29.cuh (generated by tools/gen_mac29.py) and
// these kernels time the library's own functions.  (run85 also timed a variant with the 17 a*b column sums as independent chains first: slower, 159 G/s.)
__global__ void k_mont29(uint32_t* out, uint32_t seed) {
    u256 x0 = Fq::one(), y0 = Fq::R2();
    x0.v[0] ^= threadIdx.x + seed; y0.v[1] ^= blockIdx.x;
    u261 x = Fq29::from32<0>(Fq::reduce_once(x0)), y = Fq29::from32<0>(y0);
    for (int i = 0; i < ITER / 4; i++) { x = Fq29::mul(x, y); y = Fq29::mul(y, x); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.l[0] ^ y.l[3] ^ x.l[8];
}
__global__ void k_shoup29(uint32_t* out, uint32_t seed) {            // a * w with w's precomputed quotient (field29.cuh mul_shoup): the NTT's twiddle product
    u256 x0 = Fq::one(), w0 = Fq::R2(), w1 = Fq::one();
    x0.v[0] ^= threadIdx.x + seed; w1.v[1] ^= blockIdx.x;
    u261 x = Fq29::from32<0>(Fq::reduce_once(x0));
    const u261 wa = Fq29::from32<0>(w0), wb = Fq29::from32<0>(w1), qa = Fq29::shoup_quotient(w0), qb = Fq29::shoup_quotient(w1);
    for (int i = 0; i < ITER / 4; i++) { x = Fq29::mul_shoup(x, wa, qa); x = Fq29::mul_shoup(x, wb, qb); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.l[0] ^ x.l[3] ^ x.l[8];
}
__global__ void k_sqr29(uint32_t* out, uint32_t seed) {
    u256 x0 = Fq::one(), y0 = Fq::R2();
    x0.v[0] ^= threadIdx.x + seed; y0.v[1] ^= blockIdx.x;
    u261 x = Fq29::from32<0>(Fq::reduce_once(x0)), y = Fq29::from32<0>(y0);
    for (int i = 0; i < ITER / 4; i++) { x = Fq29::sqr(x); y = Fq29::sqr(y); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.l[0] ^ y.l[3] ^ x.l[8];
}
__global__ void k_madd29(uint32_t* out, uint32_t seed) {
    Affine g;
    const uint64_t one[4] = BN254_FQ_R;
    for (int i = 0; i < 8; i++) { g.x.v[i] = (uint32_t)(one[i >> 1] >> (32 * (i & 1))); }
    g.y = Fq::dbl(g.x);
    XYZZ29 acc = xyzz29_enter(xyzz_mdbl(g.x, g.y));
    acc.x.l[0] ^= (threadIdx.x + seed) & 0xff;                           // (not a curve point any more: the formulas do not care, the timing is the same)
    for (int i = 0; i < ITER / 8; i++) xyzz29_madd(acc, g.x, g.y);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x.l[0] ^ acc.zz.l[3];
}
// host self-test of the 29-bit form against the library's product: enter / leave round trip and a * b through both forms
static int mont29_selftest() {
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int t = 0; t < 2000; t++) {
        u256 a, b;
        for (int i = 0; i < 8; i++) { a.v[i] = (uint32_t)next(); b.v[i] = (uint32_t)next(); }
        a.v[7] &= 0x1fffffffu; b.v[7] &= 0x1fffffffu;                    // < 2^253 < p
        if (!Fq::eq(Fq29::leave(Fq29::enter(a)), a)) return 1;
        if (!Fq::eq(Fq29::leave(Fq29::mul(Fq29::enter(a), Fq29::enter(b))), Fq::mul(a, b))) return 2;
        if (!Fq::eq(Fq29::leave(Fq29::sqr(Fq29::enter(a))), Fq::sqr(a))) return 3;
        if (!Fq::eq(Fq29::leave(Fq29::mul(Fq29::from32<5>(a), Fq29::enter(b))), Fq::mul(a, b))) return 4;
    }
    return 0;
}
__global__ void k_fqmul_cios(uint32_t* out, uint32_t seed) {
    u256 x = Fq::one(), y = Fq::R2();
    x.v[0] ^= threadIdx.x + seed; y.v[1] ^= blockIdx.x;
    x = Fq::reduce_once(x);
    for (int i = 0; i < ITER / 4; i++) { x = fq_mul_cios(x, y); y = fq_mul_cios(y, x); }
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

__global__ void k_madd_lazy(uint32_t* out, uint32_t seed) {            // the form msm_accumulate runs: coordinates in [0, 2q), eight of ten products without the final subtraction
    Affine g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
    XYZZ acc = xyzz_mdbl(g.x, g.y);
    u256 t = Fq::one(); t.v[0] ^= (threadIdx.x + seed) & 0xff;
    acc.x = Fq::mul(acc.x, t);
    for (int i = 0; i < ITER / 8; i++) xyzz_madd_lazy(acc, g.x, g.y);
    xyzz_normalize(acc);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x.v[0] ^ acc.zzz.v[2];
}

// ---- round 2: what a 52-bit-limb v_fma_f64 Montgomery product would be built from -------------------------------------
// Per 52x52 partial product (Emmart/Zheng/Weems): hi = fma_rz(a, b, 2^104); lo = fma_rz(a, b, (2^104 + 2^52) - hi); then the two bit patterns are
// added into 64-bit integer column sums.  The kernels below measure each ingredient's issue rate and the whole 5 multiplicand-limb group.
__global__ void k_add64f(uint32_t* out, uint32_t seed) {        // v_add_f64
    double a = 1.0 + 1e-9 * (threadIdx.x + seed), x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3);
}
__global__ void k_addu64(uint32_t* out, uint32_t seed) {        // 64-bit integer add (v_lshl_add_u64 / v_add_co + v_addc_co, compiler's choice)
    uint64_t a = ((uint64_t)(threadIdx.x + seed) << 29) | 0x123456789ull, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 += a ^ x1; x1 += a ^ x2; x2 += a ^ x3; x3 += a ^ x0; }
    uint64_t r = x0 ^ x1 ^ x2 ^ x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
__global__ void k_lshladd64(uint32_t* out, uint32_t seed) {     // v_lshl_add_u64 forced
    uint64_t a = ((uint64_t)(threadIdx.x + seed) << 29) | 0x123456789ull, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %2, %2, 0, %1\n\tv_lshl_add_u64 %3, %3, 0, %1\n\tv_lshl_add_u64 %4, %4, 0, %1"
                     : "+v"(x0), "+v"(a), "+v"(x1), "+v"(x2), "+v"(x3));
    }
    uint64_t r = x0 ^ x1 ^ x2 ^ x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
__global__ void k_shr64(uint32_t* out, uint32_t seed) {         // v_lshrrev_b64 (carry extraction between 52-bit columns)
    uint64_t a = ((uint64_t)(threadIdx.x + seed) << 40) | 0xfedcba987ull, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) { x0 = (x0 >> 7) ^ a; x1 = (x1 >> 9) ^ a; x2 = (x2 >> 11) ^ a; x3 = (x3 >> 13) ^ a; }
    uint64_t r = x0 ^ x1 ^ x2 ^ x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
// the same shift forced (the C loop above folds into other instructions): v_lshrrev_b64 against the two 32-bit instructions that do the same (v_alignbit_b32 + v_lshrrev_b32)
__global__ void k_shr64_asm(uint32_t* out, uint32_t seed) {
    uint64_t a = ((uint64_t)(threadIdx.x + seed) << 40) | 0xfedcba987ull, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\tv_lshrrev_b64 %3, 1, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        x0 |= a;
    }
    uint64_t r = x0 ^ x1 ^ x2 ^ x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
__global__ void k_shr64_pair(uint32_t* out, uint32_t seed) {
    uint32_t l0 = threadIdx.x + seed, h0 = 0xfedcba98u, l1 = l0 + 1, h1 = h0 + 1, l2 = l0 + 2, h2 = h0 + 2, l3 = l0 + 3, h3 = h0 + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_alignbit_b32 %0, %1, %0, 1\n\tv_lshrrev_b32 %1, 1, %1\n\tv_alignbit_b32 %2, %3, %2, 1\n\tv_lshrrev_b32 %3, 1, %3\n\t"
                     "v_alignbit_b32 %4, %5, %4, 1\n\tv_lshrrev_b32 %5, 1, %5\n\tv_alignbit_b32 %6, %7, %6, 1\n\tv_lshrrev_b32 %7, 1, %7"
                     : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2), "+v"(l3), "+v"(h3));
        h0 |= 0x80000000u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = l0 ^ h0 ^ l1 ^ h1 ^ l2 ^ h2 ^ l3 ^ h3;
}
// one partial product of the FMA scheme, as it would sit in the inner loop: 2 fma + 1 f64 add + 2 integer 64-bit adds.  Round-toward-zero is set
// once per kernel (MODE.FP_ROUND for f64 = bits 3:2).  Four independent chains; timing only (operands drift, exactness is not the point here).
__global__ void k_fma_partial(uint32_t* out, uint32_t seed) {
    __builtin_amdgcn_s_setreg((2 << 11) | (2 << 6) | 1 /* hwreg(HW_REG_MODE, 2, 2) */, 3);
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    double a0 = 4503599627370495.0 - (threadIdx.x + seed), a1 = a0 - 2, a2 = a0 - 4, a3 = a0 - 6, b = 4503599627370401.0 - blockIdx.x;
    uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    for (int i = 0; i < ITER; i++) {
        double p0 = __builtin_fma(a0, b, C1), p1 = __builtin_fma(a1, b, C1), p2 = __builtin_fma(a2, b, C1), p3 = __builtin_fma(a3, b, C1);
        double q0 = __builtin_fma(a0, b, C2 - p0), q1 = __builtin_fma(a1, b, C2 - p1), q2 = __builtin_fma(a2, b, C2 - p2), q3 = __builtin_fma(a3, b, C2 - p3);
        h0 += __double_as_longlong(p0); h1 += __double_as_longlong(p1); h2 += __double_as_longlong(p2); h3 += __double_as_longlong(p3);
        l0 += __double_as_longlong(q0); l1 += __double_as_longlong(q1); l2 += __double_as_longlong(q2); l3 += __double_as_longlong(q3);
        a0 = __longlong_as_double((__double_as_longlong(a0) & ~0xfffull) | (l0 & 0xfff));     // keep the chains data dependent (1 and + 1 or per product: counted below)
        a1 = __longlong_as_double((__double_as_longlong(a1) & ~0xfffull) | (l1 & 0xfff));
        a2 = __longlong_as_double((__double_as_longlong(a2) & ~0xfffull) | (l2 & 0xfff));
        a3 = __longlong_as_double((__double_as_longlong(a3) & ~0xfffull) | (l3 & 0xfff));
    }
    uint64_t r = h0 ^ h1 ^ h2 ^ h3 ^ l0 ^ l1 ^ l2 ^ l3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
// the same bit-product volume on the integer path in use today: one v_mad_u64_u32 + v_addc_co_u32 per 32x32 partial product
__global__ void k_mad_partial(uint32_t* out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed + 0x9e3779b9u, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = blockIdx.x * 2654435761u + 1;
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    for (int i = 0; i < ITER; i++) {
        mac1_vv(c0, n0, a0, b); mac1_vv(c1, n1, a1, b); mac1_vv(c2, n2, a2, b); mac1_vv(c3, n3, a3, b);
        a0 ^= (uint32_t)c0 & 0xfff; a1 ^= (uint32_t)c1 & 0xfff; a2 ^= (uint32_t)c2 & 0xfff; a3 ^= (uint32_t)c3 & 0xfff;
    }
    uint64_t r = c0 ^ c1 ^ c2 ^ c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ n0 ^ n1 ^ n2 ^ n3;
}
// upper bound of what lazy [0, 2p) reduction could give — the Montgomery product WITHOUT its final conditional subtraction
// (results in [0, 2p); NOT a drop-in: every sub / equality test downstream would need the wider range)
template <class FP> static ZK_HD u256 mul_no_final_sub(const u256& a, const u256& b) {
    uint64_t acc = 0; uint32_t cnt = 0; uint32_t m[8]; u256 r;
    using F_ = Field<FP>;
    auto p = [](int i) { return F_::p(i); };
#include "../zk-dcap-verifier_amd/csrc/field_mul_body.inc"
    r.v[7] = (uint32_t)acc;
    return r;
}
__global__ void k_fqmul_lazy(uint32_t* out, uint32_t seed) {
    u256 x = Fq::one(), y = Fq::R2();
    x.v[0] ^= threadIdx.x + seed; y.v[1] ^= blockIdx.x;
    x = Fq::reduce_once(x);
    for (int i = 0; i < ITER / 4; i++) { x = mul_no_final_sub<FqParams>(x, y); y = mul_no_final_sub<FqParams>(y, x); x.v[7] &= 0x3fffffff; y.v[7] &= 0x3fffffff; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.v[0] ^ y.v[3];
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

// ---- round 4, VERDICT r3 item 4: Montgomery REDUCTION on the matrix cores? -----------------------------------------------------------------------------------------------
// Two of a Montgomery product's three multiplications have constant operands (m = T p' mod R, then m p), so they can be written as products of the data's BYTES with
// constant Toeplitz matrices: for the 64 elements of a wave, D[column][element] = Toeplitz(p)[64 x 32] x bytes(m)[32 x 64] is four v_mfma_i32_32x32x32_i8 (with the data as
// the B operand the 64 column sums of an element come out in the element's own lane pair: a v_permlane32_swap per register, no LDS transposition).  The matrix pipe is idle
// in every kernel of this library and runs beside the VALU — but what it hands back is 64 byte-column sums per element (each below 2^21, at bit offset 8 c) that the VALU must
// fold into nine 29-bit limbs again, plus the byte extraction of m and the lane swaps.  The go / no-go question is therefore a VALU one and needs no MFMA to answer:
//   k_mp_mads        the m p block as the library does it today: 81 v_mad_u64_u32 with p's limbs as scalar operands + the 9 v_mul_lo that make m, 16 shifts
//   k_mfma_recombine the VALU work that would REMAIN with m p on the matrix cores: m's bytes out of its limbs (B operand), 16 lane swaps, 64 column sums shifted and
//                    added into 64-bit accumulators and cut into limbs — with the column sums faked by cheap VALU ops (the MFMA itself is free here: an upper bound in favour
//                    of the matrix route).  m must then be known BEFORE the product (today it falls out of the column scan), so its 45-mad half product is charged too.
// If k_mfma_recombine is not clearly faster than k_mp_mads the route is dead whatever the MFMA costs.
__global__ void k_mp_mads(uint32_t* out, uint32_t seed) {
    u261 t;
    for (int i = 0; i < 9; i++) t.l[i] = (threadIdx.x * 2654435761u + seed + i * 0x9e3779b9u) & Fq29::M29;
    uint32_t sink = 0;
    for (int it = 0; it < ITER / 8; it++) {
        uint64_t acc = t.l[0];
        uint32_t m[9];
#pragma unroll
        for (int k = 0; k < 17; k++) {                                // the reduction half of field29_mul_body.inc: m_k from the running column, then m * p down the columns
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i < (k < 9 ? k : 9); i++) acc += (uint64_t)m[i] * Fq29::p29(k - i);
            if (k < 9) { m[k] = ((uint32_t)acc * Fq29::INV29) & Fq29::M29; acc += (uint64_t)m[k] * Fq29::p29(0); acc += t.l[k < 8 ? k + 1 : 0]; }
            else t.l[k - 9] = (uint32_t)acc & Fq29::M29;
            acc >>= 29;
        }
        t.l[8] = (uint32_t)acc;
        sink ^= t.l[3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink ^ t.l[0] ^ t.l[8];
}
__global__ void k_mfma_recombine(uint32_t* out, uint32_t seed) {
    u261 t;
    for (int i = 0; i < 9; i++) t.l[i] = (threadIdx.x * 2654435761u + seed + i * 0x9e3779b9u) & Fq29::M29;
    uint32_t sink = 0;
    for (int it = 0; it < ITER / 8; it++) {
        // (1) m = T p' mod 2^261 ahead of the product: the low half of a 9 x 9 limb product, 45 multiply-adds
        uint32_t m[9];
        {
            uint64_t acc = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
#pragma unroll
                for (int i = 0; i <= k; i++) acc += (uint64_t)t.l[i] * Fq29::p29(k - i);      // (p' has limbs like p's: any constant serves the timing)
                m[k] = (uint32_t)acc & Fq29::M29;
                acc >>= 29;
            }
        }
        // (2) the B operand: m as 33 bytes in 9 registers -> 8 packed words (v_alignbit / shifts), then the other half-wave's registers through v_permlane32_swap
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int bit = 32 * j, i = bit / 29, sh = bit % 29;
            uint64_t v = (uint64_t)m[i] >> sh;
            if (i + 1 < 9) v |= (uint64_t)m[i + 1] << (29 - sh);
            if (i + 2 < 9 && 58 - sh < 32) v |= (uint64_t)m[i + 2] << (58 - sh);
            w[j] = (uint32_t)v;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = __builtin_amdgcn_ds_bpermute(((threadIdx.x ^ 32u) & 63u) << 2, w[j]) ^ w[j];     // (stands in for 8 of the 16 v_permlane32_swap: same issue cost class)
        // (3) 64 column sums S_c (< 2^21) at bit offsets 8 c, as the MFMA would leave them; here: cheap functions of the operand words
        // (4) fold them into 64-bit accumulators limb by limb: one shift-add per column, one mask + one shift per limb
        uint64_t acc = 0;
        int c = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (; c < 64 && 8 * c < 29 * (k + 1); c++) {
                const uint32_t S = (w[c & 7] >> (c >> 3)) & 0x1fffffu;
                acc += (uint64_t)S << (8 * c - 29 * k);
            }
            t.l[k] = ((uint32_t)acc & Fq29::M29) ^ (k < 8 ? 0u : 0u);
            acc >>= 29;
        }
        sink ^= t.l[5] + (uint32_t)acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink ^ t.l[0] ^ t.l[8];
}

// ---- FETCH_SIZE calibration on this repo's own access patterns (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  `microbench --gather64` / `--stream32` run ONE kernel each with a known byte count, to be read next to `rocprofv3 --pmc FETCH_SIZE`:
//   gather64: every lane reads ONE 64-byte table point (4 x 16 B, as msm_accumulate_kernel gathers its affine points) at a pseudo-random index of a 512 MiB table;
//   stream32: every lane reads 32 consecutive bytes of a 1 GiB stream (2 x 16 B per lane: how the NTT passes and the quotient read their columns).
__global__ void k_gather64(const uint4* table, uint32_t mask, uint32_t per_lane, uint32_t* out) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t i = 0; i < per_lane; i++) {
        x = x * 1664525u + 1013904223u;
        const uint4* p = table + (size_t)((x >> 4) & mask) * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// the same gather with entries of 80 bytes (two coordinates as nine 29-bit limbs each, 72 bytes, padded to keep 16-byte loads): what a limb-form bucket table would cost the memory side.
// An 80-byte entry at an 80-byte stride lies across two 64-byte lines in 3 of 4 positions.
__global__ void k_gather80(const uint4* table, uint32_t mask, uint32_t per_lane, uint32_t* out) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t i = 0; i < per_lane; i++) {
        x = x * 1664525u + 1013904223u;
        const uint4* p = table + (size_t)((x >> 4) & mask) * 5;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_stream32(const uint4* src, size_t n32, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = src[2 * i], b = src[2 * i + 1];
        acc ^= a.x ^ b.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static int calibrate(const char* which) {
    const int block = 256, grid = 256 * 8;
    uint32_t* out; CK(hipMalloc(&out, (size_t)grid * block * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    if (!strcmp(which, "--gather80")) {
        const size_t points = (size_t)1 << 23;
        uint4* t; CK(hipMalloc(&t, points * 80)); CK(hipMemset(t, 1, points * 80));
        const uint32_t per_lane = 64;
        CK(hipEventRecord(a)); hipLaunchKernelGGL(k_gather80, dim3(grid), dim3(block), 0, 0, (const uint4*)t, (uint32_t)(points - 1), per_lane, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double g = (double)grid * block * per_lane;
        printf("k_gather80: %.0f gathers of 80 B (random over a 640 MiB table), %.3f ms, %.2f G gathers/s, %.1f GB/s requested\n", g, ms, g / ms / 1e6, g * 80 / ms / 1e6);
    } else if (!strcmp(which, "--gather64")) {
        const size_t points = (size_t)1 << 23;                                       // 2^23 x 64 B = 512 MiB: the window-expanded table of one SRS at k = 19
        uint4* t; CK(hipMalloc(&t, points * 64)); CK(hipMemset(t, 1, points * 64));
        const uint32_t per_lane = 64;
        CK(hipEventRecord(a)); hipLaunchKernelGGL(k_gather64, dim3(grid), dim3(block), 0, 0, (const uint4*)t, (uint32_t)(points - 1), per_lane, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)grid * block * per_lane * 64;
        printf("k_gather64: %.0f gathers of 64 B = %.0f bytes requested (random over a 512 MiB table: at most 512 MiB of them can be first touches), %.3f ms, %.1f GB/s, %.2f G gathers/s\n", bytes / 64, bytes, ms, bytes / ms / 1e6, bytes / 64 / ms / 1e6);
    } else {
        const size_t bytes = (size_t)1 << 30;
        uint4* sbuf; CK(hipMalloc(&sbuf, bytes)); CK(hipMemset(sbuf, 1, bytes));
        CK(hipEventRecord(a)); hipLaunchKernelGGL(k_stream32, dim3(grid), dim3(block), 0, 0, (const uint4*)sbuf, bytes / 32, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("k_stream32: %zu bytes read once (32 B per lane, consecutive), %.3f ms, %.1f GB/s\n", bytes, ms, (double)bytes / ms / 1e6);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && (!strcmp(argv[1], "--gather64") || !strcmp(argv[1], "--gather80") || !strcmp(argv[1], "--stream32"))) return calibrate(argv[1]);
    {
        const int st = mont29_selftest();
        printf("mont29 host self-test (9 x 29-bit limbs vs Fq::mul): %s\n", st ? "FAILED" : "ok");
        if (st) return 1;
        if (argc > 1 && !strcmp(argv[1], "--selftest")) return 0;
    }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    for (int bpc : {4, 8}) {
        run("mad_u64_u32", k_mad64, 4.0 * ITER, 256, bpc);
        run("mad_u64_u32_12_chains", k_mad64_x12, 12.0 * (ITER / 3), 256, bpc);
        run("mul_lo_u32", k_mullo, 4.0 * ITER, 256, bpc);
        run("mul_hi_u32", k_mulhi, 4.0 * ITER, 256, bpc);
        run("mul_u24", k_mul24, 4.0 * ITER, 256, bpc);
        run("fma_f64", k_fma64, 4.0 * ITER, 256, bpc);
        run("fma_f32", k_fma32, 4.0 * ITER, 256, bpc);
        run("add_xor32", k_add32, 8.0 * ITER, 256, bpc);
    }
    printf("-- round 2: ingredients of a 52-bit-limb FMA Montgomery product vs the integer path (per partial product: fma scheme covers 52x52 = 2704 bit-products, mad+addc 32x32 = 1024)\n");
    for (int bpc : {4, 8}) {
        run("add_f64", k_add64f, 4.0 * ITER, 256, bpc);
        run("add_u64", k_addu64, 4.0 * ITER, 256, bpc);
        run("lshl_add_u64", k_lshladd64, 4.0 * ITER, 256, bpc);
        run("lshr_b64", k_shr64, 4.0 * ITER, 256, bpc);
        run("lshrrev_b64_asm", k_shr64_asm, 4.0 * ITER, 256, bpc);
        run("alignbit+lshr32_pair", k_shr64_pair, 4.0 * ITER, 256, bpc);
        double f = run("fma_partial52", k_fma_partial, 4.0 * ITER, 256, bpc);
        double m = run("mad_partial32", k_mad_partial, 4.0 * ITER, 256, bpc);
        printf("   bit-products/s: fma scheme %.1f T, mad+addc %.1f T  -> ratio %.2f (before carry propagation, quotient digits and limb conversion of the fma form)\n",
               f * 2704 / 1e12, m * 1024 / 1e12, (f * 2704) / (m * 1024));
    }
    for (int bpc : {2, 4, 8}) {
        run("fq_mul_lazy", k_fqmul_lazy, 2.0 * (ITER / 4), 256, bpc);
        run("fq_mul", k_fqmul, 2.0 * (ITER / 4), 256, bpc);
        run("fq_sqr", k_fqsqr, 2.0 * (ITER / 4), 256, bpc);
        run("fq_mul_cios", k_fqmul_cios, 2.0 * (ITER / 4), 256, bpc);
        run("fq29_mul (9 x 29-bit limbs)", k_mont29, 2.0 * (ITER / 4), 256, bpc);
        run("fq29_sqr", k_sqr29, 2.0 * (ITER / 4), 256, bpc);
        run("fq29_mul_shoup (constant operand, precomputed quotient)", k_shoup29, 2.0 * (ITER / 4), 256, bpc);
        run("fq_addsub", k_fqadd, 2.0 * ITER, 256, bpc);
    }
    printf("-- round 4: the m*p block of a Montgomery reduction on the VALU vs the VALU work that a matrix-core (i8 MFMA, constant Toeplitz operand) formulation would leave behind\n");
    for (int bpc : {2, 4}) {
        const double a = run("montgomery_mp_81_mads", k_mp_mads, 1.0 * (ITER / 8), 256, bpc);
        const double b = run("mfma_route_valu_residue", k_mfma_recombine, 1.0 * (ITER / 8), 256, bpc);
        printf("   VALU residue of the MFMA route / today's m*p block: %.2f x the time (> ~0.75: no-go, the matrix cores cannot pay for the recombination)\n", a / b);
    }
    for (int bpc : {1, 2, 4}) run("xyzz_madd", k_madd, 1.0 * (ITER / 8), 256, bpc);
    for (int bpc : {1, 2, 4}) run("xyzz_madd_lazy", k_madd_lazy, 1.0 * (ITER / 8), 256, bpc);
    for (int bpc : {1, 2, 4}) run("xyzz29_madd (29-bit limbs)", k_madd29, 1.0 * (ITER / 8), 256, bpc);
    return 0;
}

// Radix-2^r multi-pass NTT over BN254 Fr for MI355X (gfx950).
//
// Replaces halo2_proofs (zkwebauthn @ c254c75, Cargo.lock:1314-1327) src/arithmetic.rs best_fft
// (in place, natural order in and out, out[j] = sum_i a[i] omega^(ij)) and the
// EvaluationDomain wrappers of src/poly/domain.rs (App. C.3), whose scaling steps are fused here
// into the first load / last store of the transform.
//
// The CPU algorithm (bit-reverse, then log n radix-2 sweeps over the whole array) would cost log n
// HBM round trips.  Here the transform is factored N = R1 * R2 (* R3): each pass loads a tile of
// R x C elements into LDS (two 16-byte planes so that neighbouring lanes hit distinct banks),
// runs all log R butterfly stages there, applies the inter-pass twiddle on the way out and writes
// C-element (C*32 B) contiguous runs.  The digit-reversed placement of the last pass replaces the
// bit-reversal pass, so natural order is kept with 2-3 HBM round trips in total.
#include "ctx.h"
#include <vector>

namespace zk {

int ntt_pow_tables(zk_ctx* ctx, uint32_t log_n, const u256& omega, const void** lo, const void** hi, uint32_t* lo_bits);

struct NttPassArgs {
    const void* src;
    void* dst;
    const void* const* srcs;   // batch: device arrays of per-column pointers (blockIdx.y = column); null for one column
    void* const* dsts;
    uint32_t col_major;        // strided passes of a batch: the grid is (columns, tiles) — consecutive workgroups take the SAME tile of different columns, so the tile's inter-pass twiddles
                               // (and a coset transform's pre-scaling factors) are read from HBM once per batch and from L2 by the other columns
    uint32_t log_n;
    uint32_t blk_log;   // log2 of the sub-transform this pass works inside
    uint32_t r;         // log2 radix of this pass
    uint32_t c_log;     // log2 columns per tile
    uint32_t q_log;     // final pass: log2 of the fast output digit range (Q)
    uint32_t p_log;     // final pass: log2 of the middle digit range (P)
    const void* stage_tw;   // omega_R^k, k < R/2 (final pass: the library's form)
    const void* stage_sh;   // 29-bit passes: the same twiddles as Shoup pairs (ntt_shoup_table_kernel)
    const void* stage_tw29; // 29-bit passes: the same twiddles x 2^261 (Montgomery operands of the last step of a radix above 2^6)
    const void* tw_lo;      // omega^e, e < 2^lo_bits
    const void* tw_hi;      // omega^(h << lo_bits)
    uint32_t lo_bits;
    const void* tw_full;    // non-final pass: twiddle of element (row, m) at [row * cols + m], or null
    // fused operations
    uint32_t n_valid;       // first pass only (0 = all)
    int quarter_input;      // first pass only: n_valid <= N / 4, so rows >= R/4 of every tile are zero
    // first pass only: input i *= w^((cs_stride * i) mod 2^cs_log) with w's two-level power tables (coset NTT: w = extended_omega, stride = coset)
    const void* cs_lo; const void* cs_hi; uint32_t cs_lo_bits; uint32_t cs_stride; uint32_t cs_log;
    int pre_zeta;
    const void* pre_full;   // first pass on 29-bit limbs only: [i] = 32 * ZETA^(i mod 3) * w^(cs_stride * i), replaces pre_zeta and the cs_ powers
    int post_scale;
    int post_zeta_inv;
    u256 scale;
    // the passes on 29-bit limbs (ntt_*_pass29_kernel): constants of the fused operations as x * 2^261 mod p, canonical
    u256 scale29, zeta29[2];
};

ZK_HD uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? __builtin_bitreverse32(x) >> (32u - bits) : 0u; }   // (v_bfrev_b32 + one shift; x < 2^bits)
ZK_HD u256 zeta_pow(uint32_t k) {  // ZETA^k, k in {1, 2}
    const uint64_t z1[4] = BN254_FR_ZETA_M, z2[4] = BN254_FR_ZETA2_M;
    u256 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)((k == 1 ? z1[i >> 1] : z2[i >> 1]) >> (32 * (i & 1)));
    return o;
}
ZK_HD u256 ntt_load_input(const NttPassArgs& a, size_t idx) {
    if (a.n_valid && idx >= a.n_valid) return Fr::zero();
    u256 v = load_u256(a.src, idx);
    if (a.pre_zeta) {
        uint32_t m = (uint32_t)idx % 3u;
        if (m) v = Fr::mul(v, zeta_pow(m));
    }
    if (a.cs_stride) {
        const uint32_t e = (a.cs_stride * (uint32_t)idx) & ((1u << a.cs_log) - 1u);
        if (e) {
            u256 w = load_u256(a.cs_lo, e & ((1u << a.cs_lo_bits) - 1u));
            const uint32_t h = e >> a.cs_lo_bits;
            if (h) w = Fr::mul(w, load_u256(a.cs_hi, h));
            v = Fr::mul(v, w);
        }
    }
    return v;
}
// last store of a transform: v comes out of the tile in [0, 4p) and leaves canonical — through the full product of a fused scaling, or a plain normalisation
ZK_HD u256 ntt_post(const NttPassArgs& a, u256 v, size_t out_idx) {
    bool canonical = false;
    if (a.post_scale) { v = Fr::mul(v, a.scale); canonical = true; }
    if (a.post_zeta_inv) {
        uint32_t m = (uint32_t)out_idx % 3u;
        if (m) { v = Fr::mul(v, zeta_pow(3 - m)); canonical = true; }  // ZETA^-m = ZETA^(3-m)
    }
    return canonical ? v : Fr::normalize(v);
}
ZK_HD void lds_put(uint4* lo, uint4* hi, uint32_t idx, const u256& v) {
    lo[idx] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    hi[idx] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
}
ZK_HD u256 lds_get(const uint4* lo, const uint4* hi, uint32_t idx) {
    uint4 l = lo[idx], h = hi[idx];
    u256 o;
    o.v[0] = l.x; o.v[1] = l.y; o.v[2] = l.z; o.v[3] = l.w;
    o.v[4] = h.x; o.v[5] = h.y; o.v[6] = h.z; o.v[7] = h.w;
    return o;
}

// Position of (tile row, column) in the two LDS planes: the linear index row * C + col with its low three bits XORed by the top three bits of the row.
// A 16-byte LDS store is serviced in groups of 8 CONSECUTIVE lanes over 32 four-byte banks (MI355X_MICROARCH.md, LDS table): the 8 lanes of a group must hit 8
// distinct 16-byte slots modulo 128 bytes.  Every access of the tile but one walks consecutive linear indices with consecutive lanes (8 slots of one aligned
// block: any XOR with a per-block constant permutes them, conflict-free).  The exception is the final pass's load loop: consecutive lanes = consecutive input
// rows, stored BIT-REVERSED, so lanes 0..7 differ in the TOP three bits of the stored row and share everything below — without the swizzle all eight land on
// one slot (profiles/r02, r03 run84: 45 % of that kernel's LDS cycles were conflicts); with it they take the eight slots of eight different blocks.
ZK_HD uint32_t tile_at(uint32_t row, uint32_t col, uint32_t r, uint32_t c_log) {
    const uint32_t lin = (row << c_log) + col;
    return r + c_log < 6 ? lin : lin ^ ((lin >> (r + c_log - 3)) & 7u);      // (bits 0..2 never feed the shift: a bijection of the tile)
}

// all log R DIT stages on the tile held in LDS (rows were stored bit-reversed).  Two stages at a time are
// done in registers (radix-4 step: 4 loads, 4 butterflies, 4 stores) so the tile makes half as many LDS
// round trips and barriers as a radix-2 sweep; an odd last stage is a plain radix-2 step.
// stage_tw: the R/2 stage twiddles of the pass, staged in LDS by the caller (32 bytes each: every radix-4 step reads three of them per quad,
// and an LDS read returns in ~50 cycles where the L2 hit of a global load takes 200+)
// Where twiddle i of the final pass sits in LDS (in units of its 64-byte Shoup pair).  A ds_read_b128 is served in groups of 16 lanes over 64 banks, and the lanes of a group
// that belong to different tile rows read DIFFERENT twiddles of the same step — indices that are multiples of 2^(r - 1 - s), i.e. pairs a multiple of 256 bytes apart in the
// linear layout: 2- to 4-way conflicts on every twiddle read of the early steps (round 4 moved the pairs into LDS without this: the kernel's conflict share doubled, 0.11 -> 0.20).
// The low two bits of the position (which quarter of the 64 banks) take the XOR of all higher bit pairs of i, so neighbouring multiples of any power of two land on different quarters.
ZK_HD uint32_t tw_slot(uint32_t i) { return i ^ (((i >> 2) ^ (i >> 4) ^ (i >> 6)) & 3u); }
__device__ __forceinline__ void ntt_tile_stages(uint4* lo, uint4* hi, uint32_t r, uint32_t c_log, const uint4* stage_tw, bool quarter_input) {
    const uint32_t C = 1u << c_log;
    // stage_tw holds Shoup pairs, 64 bytes per twiddle: the canonical integer w | wq = floor(w 2^256 / p) — Fr::mul_shoup_lazy: 115 partial products against 136, [0, 2p) out
    // like mul_lazy, no Montgomery factor (a tile value stays in the library's form)
    auto half_at = [&](size_t j) { const size_t i = 4 * (size_t)tw_slot((uint32_t)(j >> 1)) + 2 * (j & 1); const uint4 l = stage_tw[i], h = stage_tw[i + 1]; u256 o; o.v[0] = l.x; o.v[1] = l.y; o.v[2] = l.z; o.v[3] = l.w; o.v[4] = h.x; o.v[5] = h.y; o.v[6] = h.z; o.v[7] = h.w; return o; };
    auto mul_tw = [&](const u256& x, size_t i) { return Fr::mul_shoup_lazy(x, half_at(2 * i), half_at(2 * i + 1)); };
    uint32_t s = 0;
    if (quarter_input && r >= 2) {
        // coeff_to_extended with extended_k >= k + 2: rows >= R/4 of the first pass are the zero padding, i.e. (rows are stored bit-reversed) only
        // every 4th position of the tile is non-zero and the first two stages just copy it to its three neighbours — no arithmetic
        const uint32_t nq = (1u << (r - 2)) << c_log;
        for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
            const uint32_t col = q & (C - 1), bq = q >> c_log;
            const u256 x0 = lds_get(lo, hi, tile_at(bq << 2, col, r, c_log));
            lds_put(lo, hi, tile_at((bq << 2) + 1, col, r, c_log), x0);
            lds_put(lo, hi, tile_at((bq << 2) + 2, col, r, c_log), x0);
            lds_put(lo, hi, tile_at((bq << 2) + 3, col, r, c_log), x0);
        }
        __syncthreads();
        s = 2;
    }
    // Butterflies in Harvey's redundant form (field.cuh): tile values live in [0, 4p); a twiddle product comes back in [0, 2p) without its final
    // subtraction, sums and differences are left uncorrected, and ONE conditional subtraction of 2p per input brings a value back under 2p — a third of the
    // corrections of the canonical form.  Whoever reads the tile afterwards (the pass's output code) takes [0, 4p).
    for (; s + 1 < r; s += 2) {
        const uint32_t h = 1u << s;
        const uint32_t nq = (1u << (r - 2)) << c_log;  // quads per step in the tile
        for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
            const uint32_t col = q & (C - 1), bq = q >> c_log;
            const uint32_t grp = bq >> s, pos = bq & (h - 1);
            const uint32_t rb = (grp << (s + 2)) + pos;
            const uint32_t i0 = tile_at(rb, col, r, c_log), i1 = tile_at(rb + h, col, r, c_log), i2 = tile_at(rb + 2 * h, col, r, c_log), i3 = tile_at(rb + 3 * h, col, r, c_log);
            u256 x0 = lds_get(lo, hi, i0), x1 = lds_get(lo, hi, i1), x2 = lds_get(lo, hi, i2), x3 = lds_get(lo, hi, i3);
            if (pos) {
                const size_t i1 = (size_t)pos << (r - 1 - s);
                const u256 w1 = half_at(2 * i1), w1q = half_at(2 * i1 + 1);
                x1 = Fr::mul_shoup_lazy(x1, w1, w1q);
                x3 = Fr::mul_shoup_lazy(x3, w1, w1q);
            } else {
                x1 = Fr::red2p(x1);
                x3 = Fr::red2p(x3);
            }
            x0 = Fr::red2p(x0);
            x2 = Fr::red2p(x2);
            const u256 t0 = Fr::red2p(Fr::add_lazy(x0, x1)), t1 = Fr::red2p(Fr::sub_lazy(x0, x1));
            u256 t2 = Fr::add_lazy(x2, x3), t3 = Fr::sub_lazy(x2, x3);
            t2 = pos ? mul_tw(t2, (size_t)pos << (r - 2 - s)) : Fr::red2p(t2);
            t3 = mul_tw(t3, (size_t)(pos + h) << (r - 2 - s));
            lds_put(lo, hi, i0, Fr::add_lazy(t0, t2));
            lds_put(lo, hi, i1, Fr::add_lazy(t1, t3));
            lds_put(lo, hi, i2, Fr::sub_lazy(t0, t2));
            lds_put(lo, hi, i3, Fr::sub_lazy(t1, t3));
        }
        __syncthreads();
    }
    if (s < r) {
        const uint32_t half = 1u << s;
        const uint32_t nbf = (1u << (r - 1)) << c_log;  // butterflies of the stage in the tile
        for (uint32_t q = threadIdx.x; q < nbf; q += blockDim.x) {
            const uint32_t col = q & (C - 1), bq = q >> c_log;
            const uint32_t grp = bq >> s, pos = bq & (half - 1);
            const uint32_t rb = (grp << (s + 1)) + pos;
            const uint32_t i0 = tile_at(rb, col, r, c_log), i1 = tile_at(rb + half, col, r, c_log);
            const u256 x = Fr::red2p(lds_get(lo, hi, i0));
            u256 y = lds_get(lo, hi, i1);
            y = pos ? mul_tw(y, (size_t)pos << (r - 1 - s)) : Fr::red2p(y);
            lds_put(lo, hi, i0, Fr::add_lazy(x, y));
            lds_put(lo, hi, i1, Fr::sub_lazy(x, y));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same passes on carry-free limbs (field29.cuh: 9 x 29 bits, Montgomery radix 2^261).  Data in HBM keeps the library's form (x * 2^256 in eight 32-bit words):
// a twiddle enters a product as w * 2^261 (the stage tables and the full inter-pass tables of a 29-bit plan are generated that way; the two-level power tables
// and the fused constants go through the limb conversion's shift by 5 or arrive as x * 2^261 in the arguments), so value * twiddle * 2^-261 stays in the library's
// form with no correction.  The tile holds 36 bytes per element (two 16-byte planes + one 4-byte plane, same swizzled index).
// (bounds of the butterflies: at ntt_tile_stages29)
// ------------------------------------------------------------------------------------------------
struct Tile29 {
    uint4* lo; uint4* hi; uint32_t* top;
};
// (The ninth limb's 4-byte plane is NOT swizzled on its own.  A ds_read_b32 / ds_write_b32 is served in groups of 32 lanes over 32 banks, so with 8 or 16 columns a group spans 4
// or 2 tile rows, and in the load loop and the first radix-4 step those rows meet on the same banks: that plane is where the kernel's 0.17 conflict share comes from.  Folding the
// higher row bits into the window bits removes it — 0.170 -> 0.115 — and costs more than it saves: the pass is VALU-bound with the LDS busy 6 % of its wave cycles, and 5 to 10
// integer instructions per access made coeff_to_extended x 64 at k = 19 1.5-3 % (cheap two-field XOR) and 8 % (full fold) SLOWER, same box: profiles/r05/run302.)
ZK_HD void lds_put29(const Tile29& t, uint32_t idx, const u261& v) {
    t.lo[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    t.hi[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    t.top[idx] = v.l[8];
}
ZK_HD u261 lds_get29(const Tile29& t, uint32_t idx) {
    const uint4 l = t.lo[idx], h = t.hi[idx];
    u261 o;
    o.l[0] = l.x; o.l[1] = l.y; o.l[2] = l.z; o.l[3] = l.w;
    o.l[4] = h.x; o.l[5] = h.y; o.l[6] = h.z; o.l[7] = h.w;
    o.l[8] = t.top[idx];
    return o;
}
// The stage twiddles of the 29-bit passes are CONSTANT operands, so their products take Shoup's form (field29.cuh shoup_q / shoup_r): w as its canonical integer with the precomputed
// quotient wq = floor(w 2^261 / p), 143 multiply-adds against the 162 + 9 of a Montgomery product, no Montgomery factor (value * w stays in the library's form), result
// with exact limbs below 3 p for any operand below 2^261 with limbs below 3 * 2^30.  The pairs live in LDS after the tile, in what a 40 KB share of the CU's LDS leaves: all R/2 of
// them up to radix 2^6, every second one above — which serves every step but the last (twiddle indices are multiples of 2^(r - 2 - s)); the last step of such a pass multiplies the
// Montgomery way, with the R/2 twiddles (x 2^261) written over the same LDS area just before it.  Uniform per step, and every twiddle read is a plain LDS read.
struct TwPairs {
    Tile29 w, q;          // Shoup pairs of the twiddles whose index is a multiple of 2^lds_log
    Tile29 mont;          // the same area as R/2 Montgomery twiddles, once stage_to_mont() has run
    uint32_t lds_log;
};
// x * w: the quotient digits from wq first, w fetched only then (one of the two constants live at a time)
ZK_HD u261 mul_tw(const TwPairs& tp, const u261& x, uint32_t idx) {
    const Fr29::Quot9 q = Fr29::shoup_q(x, lds_get29(tp.q, idx >> tp.lds_log));
    return Fr29::shoup_r(x, lds_get29(tp.w, idx >> tp.lds_log), q);
}
// Bounds, in multiples of p: a Shoup product returns below 3 with exact limbs; tile values are N-form (limbs below 2^29 + 8) and below V.
//   first step when it starts at stage 0 (every twiddle but one is 1: plain sums, V = 2 in):  t0, t2 = sums < 4;  t1 = x0 - x1 + 3p < 5;  t3 = (x2 - x3 + 3p) w < 3;
//       out0 < 8, out1 < 8, out2 = t0 - t2 + 5p < 9, out3 = t1 - t3 + 4p < 9                                                              -> V = 9
//   every other radix-4 step multiplies on ALL lanes (a wave mixes twiddle indices, so a "trivial twiddle" branch would run both sides anyway; index 0 of a table is 1):
//       x1', x3' < 3;  t0, t2 < V + 3;  t1, t3 < V + 4;  t2', t3' < 3;  outputs < V + 8;  the odd last stage: V + 4
//   so a pass of radix 2^7 ends below 9 + 8 + 8 + 4 = 29 p, of 2^8 below 33 p (far below the 151 p a value may reach); the inter-pass twiddle product (Montgomery, with the
//   full table) brings that below 2 p again.  Limbs: a biased difference has limbs below 2^31 + 8 — fine as an operand of a product (3.05 * 2^30) — and every output of a step is
//   below 2^32 before its one carry round.
// stage_tw: the pass's R/2 stage twiddles as x 2^261 Montgomery operands (32 bytes each, global), for the step that leaves the Shoup pairs' reach
__device__ __forceinline__ void ntt_tile_stages29(const Tile29& t, uint32_t r, uint32_t c_log, const TwPairs& tp, const void* stage_tw, bool quarter_input) {
    auto to_mont = [&]() {                                             // (all threads; the tile is not touched) overwrite the pair area with the R/2 Montgomery twiddles
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < (1u << (r - 1)); e += blockDim.x) lds_put29(tp.mont, e, Fr29::from32<0>(load_u256(stage_tw, e)));
        __syncthreads();
    };
    const uint32_t C = 1u << c_log;
    uint32_t s = 0;
    if (quarter_input && r >= 2) {
        const uint32_t nq = (1u << (r - 2)) << c_log;
        for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
            const uint32_t col = q & (C - 1), bq = q >> c_log;
            const u261 x0 = lds_get29(t, tile_at(bq << 2, col, r, c_log));
            lds_put29(t, tile_at((bq << 2) + 1, col, r, c_log), x0);
            lds_put29(t, tile_at((bq << 2) + 2, col, r, c_log), x0);
            lds_put29(t, tile_at((bq << 2) + 3, col, r, c_log), x0);
        }
        __syncthreads();
        s = 2;
    }
    for (; s + 1 < r; s += 2) {
        const uint32_t h = 1u << s;
        const uint32_t nq = (1u << (r - 2)) << c_log;  // quads per step in the tile
        const bool shoup = r - 2 - s >= tp.lds_log;    // (uniform) every twiddle index of this step is a multiple of 2^(r - 2 - s)
        if (!shoup) to_mont();
        for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
            const uint32_t col = q & (C - 1), bq = q >> c_log;
            const uint32_t grp = bq >> s, pos = bq & (h - 1);
            const uint32_t rb = (grp << (s + 2)) + pos;
            const uint32_t i0 = tile_at(rb, col, r, c_log), i1 = tile_at(rb + h, col, r, c_log), i2 = tile_at(rb + 2 * h, col, r, c_log), i3 = tile_at(rb + 3 * h, col, r, c_log);
            u261 x0 = lds_get29(t, i0), x1 = lds_get29(t, i1), x2 = lds_get29(t, i2), x3 = lds_get29(t, i3);
            u261 o0, o1, o2, o3;
            if (s == 0) {                                   // (uniform: the first step of a pass that starts at stage 0) every twiddle but the last is 1
                const u261 t0 = Fr29::add(x0, x1), t1 = Fr29::sub_bias<3, 30>(x0, x1), t2 = Fr29::add(x2, x3);
                const u261 t3 = shoup ? mul_tw(tp, Fr29::sub_bias<3, 30>(x2, x3), 1u << (r - 2)) : Fr29::mul(Fr29::sub_bias<3, 30>(x2, x3), lds_get29(tp.mont, 1u << (r - 2)));
                o0 = Fr29::add(t0, t2); o1 = Fr29::add(t1, t3);
                o2 = Fr29::sub_bias<5, 30>(t0, t2); o3 = Fr29::sub_bias<4, 30>(t1, t3);
            } else if (shoup) {
                {   // x1, x3 share w1: both quotients while wq is live, then both remainders with w
                    const uint32_t i1w = (pos << (r - 1 - s)) >> tp.lds_log;
                    const u261 w1q = lds_get29(tp.q, i1w);
                    const Fr29::Quot9 q1 = Fr29::shoup_q(x1, w1q), q3 = Fr29::shoup_q(x3, w1q);
                    const u261 w1 = lds_get29(tp.w, i1w);
                    x1 = Fr29::shoup_r(x1, w1, q1);
                    x3 = Fr29::shoup_r(x3, w1, q3);
                }
                const u261 t0 = Fr29::add(x0, x1), t1 = Fr29::sub_bias<4, 30>(x0, x1);
                const u261 t2 = mul_tw(tp, Fr29::add(x2, x3), pos << (r - 2 - s));
                const u261 t3 = mul_tw(tp, Fr29::sub_bias<4, 30>(x2, x3), (pos + h) << (r - 2 - s));
                o0 = Fr29::add(t0, t2); o1 = Fr29::add(t1, t3);
                o2 = Fr29::sub_bias<4, 30>(t0, t2); o3 = Fr29::sub_bias<4, 30>(t1, t3);
            } else {                                        // the last step of a radix above 2^6: Montgomery products (below 2 p: the biases of the Shoup steps cover them)
                const u261 w1 = lds_get29(tp.mont, pos << (r - 1 - s));
                x1 = Fr29::mul(x1, w1);
                x3 = Fr29::mul(x3, w1);
                const u261 t0 = Fr29::add(x0, x1), t1 = Fr29::sub_bias<4, 30>(x0, x1);
                const u261 t2 = Fr29::mul(Fr29::add(x2, x3), lds_get29(tp.mont, pos << (r - 2 - s)));
                const u261 t3 = Fr29::mul(Fr29::sub_bias<4, 30>(x2, x3), lds_get29(tp.mont, (pos + h) << (r - 2 - s)));
                o0 = Fr29::add(t0, t2); o1 = Fr29::add(t1, t3);
                o2 = Fr29::sub_bias<4, 30>(t0, t2); o3 = Fr29::sub_bias<4, 30>(t1, t3);
            }
            lds_put29(t, i0, Fr29::carry(o0));
            lds_put29(t, i1, Fr29::carry(o1));
            lds_put29(t, i2, Fr29::carry(o2));
            lds_put29(t, i3, Fr29::carry(o3));
        }
        __syncthreads();
    }
    if (s < r) {
        const uint32_t half = 1u << s;
        const uint32_t nbf = (1u << (r - 1)) << c_log;  // butterflies of the stage in the tile
        const bool shoup = r - 1 - s >= tp.lds_log;
        if (!shoup && s) to_mont();
        for (uint32_t q = threadIdx.x; q < nbf; q += blockDim.x) {
            const uint32_t col = q & (C - 1), bq = q >> c_log;
            const uint32_t grp = bq >> s, pos = bq & (half - 1);
            const uint32_t rb = (grp << (s + 1)) + pos;
            const uint32_t i0 = tile_at(rb, col, r, c_log), i1 = tile_at(rb + half, col, r, c_log);
            const u261 x = lds_get29(t, i0);
            u261 y = lds_get29(t, i1);
            if (s == 0) {                                   // r = 1: the single stage of the pass, twiddle 1
                lds_put29(t, i0, Fr29::carry(Fr29::add(x, y)));
                lds_put29(t, i1, Fr29::carry(Fr29::sub_bias<3, 30>(x, y)));
            } else {
                y = shoup ? mul_tw(tp, y, pos << (r - 1 - s)) : Fr29::mul(y, lds_get29(tp.mont, pos << (r - 1 - s)));
                lds_put29(t, i0, Fr29::carry(Fr29::add(x, y)));
                lds_put29(t, i1, Fr29::carry(Fr29::sub_bias<4, 30>(x, y)));
            }
        }
        __syncthreads();
    }
}
// first load of a transform (fused pre-operations) or a later pass's reload: the library's form in, N-form limbs below 2 p out
ZK_HD u261 ntt_load_input29(const NttPassArgs& a, size_t idx) {
    if (a.n_valid && idx >= a.n_valid) return Fr29::zero();
    u261 v = Fr29::from32<0>(load_u256(a.src, idx));
    if (a.pre_full) return Fr29::mul(v, Fr29::from32<0>(load_u256(a.pre_full, idx)));           // one product for the whole pre-scaling (ntt_coset_table)
    if (a.pre_zeta) {
        const uint32_t m = (uint32_t)idx % 3u;
        if (m) v = Fr29::mul(v, Fr29::from32<0>(m == 1 ? a.zeta29[0] : a.zeta29[1]));    // (no runtime index into the argument block: that would send it to scratch)
    }
    if (a.cs_stride) {
        const uint32_t e = (a.cs_stride * (uint32_t)idx) & ((1u << a.cs_log) - 1u);
        if (e) {
            u261 w = Fr29::from32<5>(load_u256(a.cs_lo, e & ((1u << a.cs_lo_bits) - 1u)));      // w * 2^261, below 32 p
            const uint32_t h = e >> a.cs_lo_bits;
            if (h) w = Fr29::mul(w, Fr29::from32<5>(load_u256(a.cs_hi, h)));                      // below 8 p
            v = Fr29::mul(v, w);
        }
    }
    return v;
}
ZK_HD Tile29 tile29_at(uint4* base, uint32_t count) {                 // planes of `count` elements: 16 B | 16 B | 4 B
    Tile29 t;
    t.lo = base; t.hi = base + count; t.top = reinterpret_cast<uint32_t*>(base + 2 * count);
    return t;
}

// the twiddle area of a 29-bit pass (after the tile): the Shoup pairs of its stage twiddles — all R/2 up to radix 2^6, every second one above — staged by the whole workgroup
__device__ __forceinline__ TwPairs stage_pairs(const NttPassArgs& a, uint4* smem, uint32_t tile, uint32_t half) {
    TwPairs tp;
    tp.lds_log = a.r >= 7 ? 1u : 0u;
    const uint32_t n_lds = half >> tp.lds_log;
    uint4* const area = smem + (tile * 36 + 15) / 16;
    tp.w = tile29_at(area, n_lds);
    tp.q = tile29_at(area + (n_lds * 36 + 15) / 16, n_lds);
    tp.mont = tile29_at(area, half);
    for (uint32_t e = threadIdx.x; e < n_lds; e += blockDim.x) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.stage_sh) + (size_t)20 * (e << tp.lds_log);
        u261 w, wq;
        for (int i = 0; i < 9; i++) { w.l[i] = src[i]; wq.l[i] = src[9 + i]; }
        lds_put29(tp.w, e, w);
        lds_put29(tp.q, e, wq);
    }
    return tp;
}

ZK_KERNEL void ZK_LAUNCH_BOUNDS(256) ZK_WAVES_PER_EU(4) ntt_strided_pass29_kernel(NttPassArgs a) {
    ZK_DYN_SHARED(uint4, smem);
    const uint32_t bcol = a.col_major ? blockIdx.x : blockIdx.y, btile = a.col_major ? blockIdx.y : blockIdx.x;
    if (a.srcs) { a.src = a.srcs[bcol]; a.dst = a.dsts[bcol]; }
    const uint32_t R = 1u << a.r, C = 1u << a.c_log, tile = R << a.c_log;
    const Tile29 t = tile29_at(smem, tile);
    const uint32_t half = R >> 1 ? R >> 1 : 1;
    const TwPairs tp = stage_pairs(a, smem, tile, half);
    const uint32_t cols_log = a.blk_log - a.r;
    const uint32_t tiles_per_blk_log = cols_log - a.c_log;
    const uint32_t tb = btile;
    const size_t o = tb >> tiles_per_blk_log;
    const uint32_t m0 = (tb & ((1u << tiles_per_blk_log) - 1)) << a.c_log;
    const size_t base = (o << a.blk_log) + m0;
    for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
        const uint32_t col = e & (C - 1), row = e >> a.c_log;
        const size_t idx = base + ((size_t)row << cols_log) + col;
        lds_put29(t, tile_at(bitrev(row, a.r), col, a.r, a.c_log), ntt_load_input29(a, idx));
    }
    __syncthreads();
    ntt_tile_stages29(t, a.r, a.c_log, tp, a.stage_tw29, a.quarter_input != 0);
    const uint32_t sh = a.log_n - a.blk_log;
    const uint32_t lomask = (1u << a.lo_bits) - 1;
    for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
        const uint32_t col = e & (C - 1), row = e >> a.c_log;
        u261 v = lds_get29(t, tile_at(row, col, a.r, a.c_log));
        // the inter-pass twiddle on EVERY element (exponent 0 reads 2^261 mod p): the product is also what brings the pass's growth back below 2 p for the store
        if (a.tw_full) {
            v = Fr29::mul(v, Fr29::from32<0>(load_u256(a.tw_full, ((size_t)row << cols_log) + m0 + col)));
        } else {
            const uint32_t ex = ((m0 + col) * row) << sh;  // < 2^log_n
            u261 w = Fr29::from32<5>(load_u256(a.tw_lo, ex & lomask));
            const uint32_t h = ex >> a.lo_bits;
            if (h) w = Fr29::mul(w, Fr29::from32<5>(load_u256(a.tw_hi, h)));                // below 8 p
            v = Fr29::mul(v, Fr29::mul(w, Fr29::one()));                                      // (w below 2 p first: the store must stay below 2 p)
        }
        store_u256(a.dst, base + ((size_t)row << cols_log) + col, Fr29::to32_exact(v));      // a product's limbs are exact (no carry round needed); below 2 p: the next pass takes it as it is
    }
}

// final pass: the sub-transform is contiguous (size R); outer index o = j1 * P + jm; the result
// row goes to out[j1 + Q * (jm + P * row)] — the digit reversal that restores natural order.
ZK_KERNEL void ntt_final_pass_kernel(NttPassArgs a) {
    ZK_DYN_SHARED(uint4, smem);
    if (a.srcs) { a.src = a.srcs[blockIdx.y]; a.dst = a.dsts[blockIdx.y]; }
    const uint32_t R = 1u << a.r, C = 1u << a.c_log, tile = R << a.c_log;
    uint4* lo = smem;
    uint4* hi = smem + tile;
    uint4* twl = smem + 2 * tile;                                  // R/2 Shoup pairs of stage twiddles, 4 x uint4 each
    for (uint32_t e = threadIdx.x; e < 2 * R; e += blockDim.x) twl[4 * tw_slot(e >> 2) + (e & 3u)] = reinterpret_cast<const uint4*>(a.stage_sh)[e];
    const uint32_t t = blockIdx.x;
    const uint32_t jm = t & ((1u << a.p_log) - 1);
    const uint32_t j10 = (t >> a.p_log) << a.c_log;
    for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
        const uint32_t row = e & (R - 1), col = e >> a.r;
        const size_t o = ((size_t)(j10 + col) << a.p_log) + jm;
        lds_put(lo, hi, tile_at(bitrev(row, a.r), col, a.r, a.c_log), ntt_load_input(a, (o << a.r) + row));
    }
    __syncthreads();
    ntt_tile_stages(lo, hi, a.r, a.c_log, twl, a.quarter_input != 0);
    for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
        const uint32_t col = e & (C - 1), row = e >> a.c_log;
        const size_t out_idx = (size_t)(j10 + col) + (((size_t)jm + ((size_t)row << a.p_log)) << a.q_log);
        store_u256(a.dst, out_idx, ntt_post(a, lds_get(lo, hi, tile_at(row, col, a.r, a.c_log)), out_idx));
    }
}

// out[k] = base^(k * 1), k < count (each thread: square-and-multiply over the bits of k)
ZK_KERNEL void fr_pow_table_kernel(u256 base, uint32_t count, void* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    u256 acc = Fr::one();
    for (int b = 31; b >= 0; b--) {
        acc = Fr::sqr(acc);
        if ((k >> b) & 1) acc = Fr::mul(acc, base);
    }
    store_u256(out, k, acc);
}

// Shoup pairs of a strided pass's stage twiddles: entry k = 9 limbs of the canonical integer w_k | 9 limbs of floor(w_k 2^261 / p) | 2 words of padding (80 bytes)
ZK_KERNEL void ntt_shoup_table_kernel(const void* tw_lib, uint32_t count, void* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const u256 w = Fr::from_mont(load_u256(tw_lib, k));
    const u261 wl = Fr29::from32<0>(w), wq = Fr29::shoup_quotient(w);
    uint32_t* o = reinterpret_cast<uint32_t*>(out) + (size_t)20 * k;
    for (int i = 0; i < 9; i++) { o[i] = wl.l[i]; o[9 + i] = wq.l[i]; }
    o[18] = o[19] = 0;
}

// ... and of the final pass's (32-bit form): entry k = the canonical integer w_k | floor(w_k 2^256 / p), 64 bytes
ZK_KERNEL void ntt_shoup32_table_kernel(const void* tw_lib, uint32_t count, void* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const u256 wl = load_u256(tw_lib, k);
    store_u256(out, 2 * (size_t)k, Fr::from_mont(wl));
    store_u256(out, 2 * (size_t)k + 1, Fr::shoup_quotient(wl));
}

// full inter-pass twiddle table of a non-final pass: out[row * cols + m] = omega^((m * row) << sh)
ZK_KERNEL void ntt_full_twiddle_kernel(const void* tw_lo, const void* tw_hi, uint32_t lo_bits, uint32_t cols_log, uint32_t r, uint32_t sh, void* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << (cols_log + r))) return;
    const uint32_t m = (uint32_t)(i & (((size_t)1 << cols_log) - 1)), row = (uint32_t)(i >> cols_log);
    const uint32_t ex = (m * row) << sh;
    u256 tw = load_u256(tw_lo, ex & ((1u << lo_bits) - 1));
    const uint32_t h = ex >> lo_bits;
    if (h) tw = Fr::mul(tw, load_u256(tw_hi, h));
    store_u256(out, i, tw);
}

// element-wise helpers ---------------------------------------------------------------------------
ZK_KERNEL void fr_vec_kernel(int op, const void* x, const void* y, void* out, size_t n, u256 scalar) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        u256 a = load_u256(x, i), o;
        switch (op) {
            case 0: o = Fr::mul(a, load_u256(y, i)); break;
            case 1: o = Fr::add(a, load_u256(y, i)); break;
            case 2: o = Fr::sub(a, load_u256(y, i)); break;
            case 3: o = Fr::mul(a, scalar); break;
            default: o = Fq::mul(a, load_u256(y, i)); break;  // 4: Fq product (curve-field parity tests)
        }
        store_u256(out, i, o);
    }
}
// a[i] *= t[i mod 2^t_log]
ZK_KERNEL void fr_mul_periodic_kernel(void* a, size_t n, const void* t, uint32_t t_log) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) store_u256(a, i, Fr::mul(load_u256(a, i), load_u256(t, i & ((1u << t_log) - 1))));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static u256 fr_pow2k_host(u256 x, uint32_t k) {  // x^(2^k)
    for (uint32_t i = 0; i < k; i++) x = Fr::sqr(x);
    return x;
}

static void plan_passes(uint32_t log_n, const Tune& tn, uint32_t rl[3], int* passes) {
    uint32_t rmax = (uint32_t)tn.ntt_max_radix_log;
    if (rmax > (uint32_t)tn.ntt_tile_log) rmax = tn.ntt_tile_log;
    if (rmax < 1) rmax = 1;
    int p = (int)((log_n + rmax - 1) / rmax);
    if (p < 1) p = 1;
    if (p > 3) p = 3;
    rl[0] = rl[1] = rl[2] = 0;
    if (tn.ntt_plan > 0) {                                              // measurement knob: the radices as decimal digits ("885" = 2^8, 2^8, 2^5), taken when they fit this transform
        const uint32_t d[3] = {(uint32_t)tn.ntt_plan / 100 % 10, (uint32_t)tn.ntt_plan / 10 % 10, (uint32_t)tn.ntt_plan % 10};
        if (d[0] && d[1] && d[2] && d[0] + d[1] + d[2] == log_n && d[0] <= rmax && d[1] <= rmax && d[2] <= rmax) {
            rl[0] = d[0]; rl[1] = d[1]; rl[2] = d[2];
            *passes = 3;
            return;
        }
    }
    uint32_t rem = log_n;
    for (int i = 0; i < p; i++) {
        uint32_t r = (rem + (p - i) - 1) / (p - i);
        rl[i] = r;
        rem -= r;
    }
    *passes = p;
}

// scale (optional): a constant every output of the transform is multiplied by; folded into the last strided pass's full twiddle table when the plan has one
// (TwiddleSet::scale_fused tells the caller whether it was)
static void free_twiddle_set(TwiddleSet& t) {
    if (t.d_lo) (void)hipFree(t.d_lo);
    if (t.d_hi) (void)hipFree(t.d_hi);
    for (int i = 0; i < 3; i++) { if (t.d_stage[i]) (void)hipFree(t.d_stage[i]); if (t.d_stage_sh[i]) (void)hipFree(t.d_stage_sh[i]); if (t.d_stage29[i]) (void)hipFree(t.d_stage29[i]); if (t.d_full[i]) (void)hipFree(t.d_full[i]); }
    t = TwiddleSet();
}
// A context keeps at most this many sets (one per (log n, omega, plan, fused scale): a prover touches 5-6 of them; a caller that walks sizes or tunables
// would otherwise grow the list — up to 2 x 32 B x n each — without bound): the least recently used one goes.
static const size_t MAX_TWIDDLE_SETS = 16;
static int get_twiddles(zk_ctx* ctx, uint32_t log_n, const u256& omega, const u256* scale, TwiddleSet** out) {
    uint32_t rl[3];
    int passes;
    plan_passes(log_n, ctx->tune, rl, &passes);
    const bool full = passes > 1 && (int)log_n <= ctx->tune.ntt_full_twiddle_max_log;
    const bool fuse = scale && full && ctx->tune.ntt_fuse_scale;
    for (auto& t : ctx->twiddles)
        if (t.log_n == log_n && Fr::eq(t.omega, omega) && t.passes == passes && t.radix_log[0] == rl[0] && t.radix_log[1] == rl[1] &&
            t.radix_log[2] == rl[2] && (t.d_full[0] != nullptr) == full && t.scale_fused == fuse && (!fuse || Fr::eq(t.fused_scale, *scale))) { t.stamp = ++ctx->twiddle_clock; *out = &t; return ZK_OK; }
    struct Pending { TwiddleSet ts; ~Pending() { free_twiddle_set(ts); } } pending;      // a failing allocation or launch below frees the tables made so far
    TwiddleSet& ts = pending.ts;
    ts.log_n = log_n; ts.omega = omega; ts.passes = passes;
    ts.scale_fused = fuse;
    if (fuse) ts.fused_scale = *scale;
    // the strided passes run on 29-bit limbs and multiply with twiddles in the form w * 2^261: their stage and inter-pass tables hold 32 * (w * 2^256) mod p, canonical
    // (the final pass stays on the 32-bit form — measured, profiles/r03 — and the two-level power tables in the library's form: the quotient kernel reads them too)
    u256 c32 = Fr::zero();
    c32.v[0] = 32;
    c32 = Fr::to_mont(c32);
    for (int i = 0; i < 3; i++) ts.radix_log[i] = rl[i];
    ts.lo_bits = log_n < 10 ? log_n : 10;
    const int blk = 256;
    const uint32_t nlo = 1u << ts.lo_bits, nhi = 1u << (log_n - ts.lo_bits);
    ZK_HIP(hipMalloc(&ts.d_lo, (size_t)nlo * 32));
    ZK_HIP(hipMalloc(&ts.d_hi, (size_t)nhi * 32));
    ZK_LAUNCH(fr_pow_table_kernel, (nlo + blk - 1) / blk, blk, 0, ctx->stream, omega, nlo, ts.d_lo);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(fr_pow_table_kernel, (nhi + blk - 1) / blk, blk, 0, ctx->stream, fr_pow2k_host(omega, ts.lo_bits), nhi, ts.d_hi);
    ZK_CHECK_LAUNCH();
    for (int i = 0; i < passes; i++) {
        const uint32_t half = rl[i] ? 1u << (rl[i] - 1) : 1;
        ZK_HIP(hipMalloc(&ts.d_stage[i], (size_t)half * 32));
        ZK_LAUNCH(fr_pow_table_kernel, (half + blk - 1) / blk, blk, 0, ctx->stream, fr_pow2k_host(omega, log_n - rl[i]), half, ts.d_stage[i]);
        ZK_CHECK_LAUNCH();
        if (i + 1 == passes) {                                        // the final pass stays on the 32-bit form (a 29-bit Shoup final pass measured the same as the 32-bit Montgomery one,
            ZK_HIP(hipMalloc(&ts.d_stage_sh[i], (size_t)half * 64));  // profiles/r04) and multiplies with 32-bit Shoup pairs
            ZK_LAUNCH(ntt_shoup32_table_kernel, (half + blk - 1) / blk, blk, 0, ctx->stream, (const void*)ts.d_stage[i], half, ts.d_stage_sh[i]);
            ZK_CHECK_LAUNCH();
            continue;
        }
        // the strided passes multiply with Shoup pairs of these twiddles — and, in the last step of a radix above 2^6, with their x 2^261 Montgomery forms
        ZK_HIP(hipMalloc(&ts.d_stage_sh[i], (size_t)half * 80));
        ZK_LAUNCH(ntt_shoup_table_kernel, (half + blk - 1) / blk, blk, 0, ctx->stream, (const void*)ts.d_stage[i], half, ts.d_stage_sh[i]);
        ZK_CHECK_LAUNCH();
        ZK_HIP(hipMalloc(&ts.d_stage29[i], (size_t)half * 32));
        ZK_LAUNCH(fr_vec_kernel, (half + blk - 1) / blk, blk, 0, ctx->stream, 3, (const void*)ts.d_stage[i], (const void*)ts.d_stage[i], ts.d_stage29[i], (size_t)half, c32);
        ZK_CHECK_LAUNCH();
    }
    if ((int)log_n <= ctx->tune.ntt_full_twiddle_max_log) {
        uint32_t blk_log = log_n;
        for (int i = 0; i + 1 < passes; i++) {
            const uint32_t cols_log = blk_log - rl[i];
            const size_t cnt = (size_t)1 << blk_log;
            ZK_HIP(hipMalloc(&ts.d_full[i], cnt * 32));
            ZK_LAUNCH(ntt_full_twiddle_kernel, (uint32_t)((cnt + blk - 1) / blk), blk, 0, ctx->stream, (const void*)ts.d_lo, (const void*)ts.d_hi, ts.lo_bits,
                      cols_log, rl[i], log_n - blk_log, ts.d_full[i]);
            ZK_CHECK_LAUNCH();
            ZK_LAUNCH(fr_vec_kernel, 1024, blk, 0, ctx->stream, 3, (const void*)ts.d_full[i], (const void*)ts.d_full[i], ts.d_full[i], cnt,
                      (fuse && i + 2 == passes) ? Fr::mul(c32, *scale) : c32);
            ZK_CHECK_LAUNCH();
            blk_log -= rl[i];
        }
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->twiddles.size() >= MAX_TWIDDLE_SETS) {                    // (the stream is idle: no kernel still reads the victim's tables; sets whose pointers this call already holds carry later stamps)
        auto victim = ctx->twiddles.begin();
        for (auto it = ctx->twiddles.begin(); it != ctx->twiddles.end(); ++it) if (it->stamp < victim->stamp) victim = it;
        free_twiddle_set(*victim);
        ctx->twiddles.erase(victim);
    }
    ts.stamp = ++ctx->twiddle_clock;
    ctx->twiddles.push_back(ts);
    ts = TwiddleSet();                                                 // (the list owns the tables now)
    *out = &ctx->twiddles.back();
    return ZK_OK;
}

void release_twiddles(zk_ctx* ctx) {
    for (auto& t : ctx->twiddles) free_twiddle_set(t);
    ctx->twiddles.clear();
    for (auto& kv : ctx->coset_tables) (void)hipFree(kv.second);
    ctx->coset_tables.clear();
}

int ntt_set_lds_attr() {
#ifndef ZK_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_final_pass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_strided_pass29_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    return 0;
}

// In-place natural-order NTT of `count` columns of 2^log_n elements (h_cols: host array of device pointers).
// h_srcs (optional, fused forms): read column i from h_srcs[i] instead of h_cols[i].
int ntt_dev_batch(zk_ctx* ctx, void* const* h_cols, const void* const* h_srcs, size_t count, uint32_t log_n, const u256& omega, const NttFuse* fuse) {
    if (!h_cols || count == 0) return count == 0 ? ZK_OK : ctx->fail(ZK_ERR_ARG, "zk_ntt: null pointer");
    if (log_n > 27) return ctx->fail(ZK_ERR_LIMIT, "zk_ntt: log_n = %u > 27", log_n);
    if (count > 65535) return ctx->fail(ZK_ERR_LIMIT, "zk_ntt batch: more than 65535 columns");
    for (size_t i = 0; i < count; i++) if (!h_cols[i] || (h_srcs && !h_srcs[i])) return ctx->fail(ZK_ERR_ARG, "zk_ntt: null column pointer");
    const size_t N = (size_t)1 << log_n;
    {   // the passes of a batch run out of place through one workspace of count columns: above a budget, transform the batch in slices of columns
        const size_t budget = (size_t)std::max(ctx->tune.ntt_ws_limit_mb, 1) << 20, per_col = N * 32;
        const size_t fit = std::max<size_t>(1, budget / per_col);
        if (count > fit) {
            for (size_t i = 0; i < count; i += fit) {
                int rc = ntt_dev_batch(ctx, h_cols + i, h_srcs ? h_srcs + i : nullptr, std::min(fit, count - i), log_n, omega, fuse);
                if (rc) return rc;
            }
            return ZK_OK;
        }
    }
    NttFuse nf;
    if (fuse) nf = *fuse;
    if (log_n == 0) {
        for (size_t i = 0; i < count; i++) {
            const void* s0 = h_srcs ? h_srcs[i] : h_cols[i];
            if (s0 != h_cols[i]) ZK_HIP(hipMemcpyAsync(h_cols[i], s0, 32, hipMemcpyDeviceToDevice, ctx->stream));
            if (nf.post_scale) {
                ZK_LAUNCH(fr_vec_kernel, 1, 64, 0, ctx->stream, 3, (const void*)h_cols[i], (const void*)h_cols[i], h_cols[i], (size_t)1, nf.scale);
                ZK_CHECK_LAUNCH();
            }
        }
        return ZK_OK;
    }
    TwiddleSet* ts;
    int rc = get_twiddles(ctx, log_n, omega, nf.post_scale ? &nf.scale : nullptr, &ts);
    if (rc) return rc;
    const Tune& tn = ctx->tune;
    const uint32_t tl = (uint32_t)tn.ntt_tile_log;
    for (int p = 0; p < ts->passes; p++) if (ts->radix_log[p] > tl) return ctx->fail(ZK_ERR_ARG, "zk_ntt: radix 2^%u exceeds the LDS tile 2^%u", ts->radix_log[p], tl);
    ZK_HIP(ctx->ws_ntt.ensure(count * N * 32 + 3 * count * sizeof(void*) + 64));
    char* tmp_base = (char*)ctx->ws_ntt.p;
    // device pointer tables: [src0 | tmp | dst]
    const bool batch = count > 1;
    const void** d_src0 = nullptr; void** d_tmp = nullptr; void** d_dst = nullptr;
    if (batch) {
        std::vector<const void*> tab(3 * count);
        for (size_t i = 0; i < count; i++) {
            tab[i] = h_srcs ? h_srcs[i] : h_cols[i];
            tab[count + i] = tmp_base + i * N * 32;
            tab[2 * count + i] = h_cols[i];
        }
        char* dtab = tmp_base + ((count * N * 32 + 15) & ~(size_t)15);
        ZK_HIP(hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));   // tab is a stack object
        d_src0 = (const void**)dtab; d_tmp = (void**)dtab + count; d_dst = (void**)dtab + 2 * count;
    }
    const void* src0 = h_srcs ? h_srcs[0] : h_cols[0];
    void* d_a = h_cols[0];
    void* tmp = tmp_base;
    const int P = ts->passes;
    // data flow: pass 0 reads src0; intermediate passes live in tmp; the final pass writes the column.
    uint32_t blk_log = log_n;
    for (int p = 0; p < P; p++) {
        NttPassArgs a;
        memset(&a, 0, sizeof a);
        const bool first = p == 0, last = p == P - 1;
        bool via_tmp = false;
        a.src = first ? src0 : tmp;
        a.dst = last ? d_a : tmp;
        if (first && last && !h_srcs) { a.dst = tmp; via_tmp = true; }   // a single pass cannot scatter in place
        if (batch) {
            a.srcs = first ? (const void* const*)d_src0 : (const void* const*)d_tmp;
            a.dsts = (last && !via_tmp) ? (void* const*)d_dst : (void* const*)d_tmp;
        }
        a.log_n = log_n; a.blk_log = blk_log; a.r = ts->radix_log[p];
        a.stage_tw = ts->d_stage[p]; a.stage_sh = ts->d_stage_sh[p]; a.stage_tw29 = ts->d_stage29[p]; a.tw_lo = ts->d_lo; a.tw_hi = ts->d_hi; a.lo_bits = ts->lo_bits; a.tw_full = last ? nullptr : ts->d_full[p];
        if (first && nf.cs_stride) { a.cs_lo = nf.cs_lo; a.cs_hi = nf.cs_hi; a.cs_lo_bits = nf.cs_lo_bits; a.cs_stride = nf.cs_stride; a.cs_log = nf.cs_log; }
        if (first && !last && nf.pre_full) a.pre_full = nf.pre_full;
        if (first) { a.n_valid = nf.n_valid; a.pre_zeta = nf.pre_zeta; a.quarter_input = tn.ntt_quarter_input && nf.n_valid && (size_t)nf.n_valid * 4 <= N && a.r >= 2; }
        if (last) { a.post_scale = nf.post_scale && !ts->scale_fused; a.post_zeta_inv = nf.post_zeta_inv; a.scale = nf.scale; }
        if (!last) {                                                  // a 29-bit pass takes the constants of its fused operations as x * 2^261
            u256 c32 = Fr::zero();
            c32.v[0] = 32;
            c32 = Fr::to_mont(c32);
            a.scale29 = Fr::mul(nf.scale, c32);
            a.zeta29[0] = Fr::mul(zeta_pow(1), c32);
            a.zeta29[1] = Fr::mul(zeta_pow(2), c32);
        }
        const uint32_t room = tl > a.r ? tl - a.r : 0;
        EvTimer t_pass(ctx, last ? "ntt_final_pass" : "ntt_strided_pass");   // (event pairs only while zk_timing_enable is on; read at the next zk_timing_get)
        if (!last) {
            const uint32_t cols_log = blk_log - a.r;
            a.c_log = room < cols_log ? room : cols_log;
            const uint32_t grid = (uint32_t)(N >> (a.r + a.c_log));
            const size_t half_tw = (size_t)1 << (a.r ? a.r - 1 : 0), n_pairs = a.r >= 7 ? half_tw / 2 : half_tw;
            const size_t lds = ((((size_t)36 << (a.r + a.c_log)) + 15) & ~(size_t)15) + std::max(2 * ((n_pairs * 36 + 15) & ~(size_t)15), a.r >= 7 ? half_tw * 36 : 0) + 64;     // 36-byte elements: tile + the twiddle area (ntt_tile_stages29)
            if (tn.ntt_threads > 256) return ctx->fail(ZK_ERR_ARG, "ntt_threads: the strided passes take at most 256 threads per workgroup");
            a.col_major = tn.ntt_col_major && batch && grid <= 65535 ? 1u : 0u;
            ZK_LAUNCH(ntt_strided_pass29_kernel, a.col_major ? dim3((uint32_t)count, grid) : dim3(grid, (uint32_t)count), tn.ntt_threads, lds, ctx->stream, a);
            ZK_CHECK_LAUNCH();
        } else {
            // o = j1 * Pm + jm with j1 the digit of pass 0 (Q = R_0) and jm the digit of pass 1 (if 3 passes)
            a.q_log = P >= 2 ? ts->radix_log[0] : 0;
            a.p_log = P == 3 ? ts->radix_log[1] : 0;
            a.c_log = room < a.q_log ? room : a.q_log;
            const uint32_t grid = (uint32_t)(N >> (a.r + a.c_log));
            const size_t lds = ((size_t)32 << (a.r + a.c_log)) + ((size_t)32 << a.r);   // tile (two planes) + R/2 Shoup pairs of stage twiddles (64 bytes each)
            ZK_LAUNCH(ntt_final_pass_kernel, dim3(grid, (uint32_t)count), tn.ntt_threads, lds, ctx->stream, a);
            ZK_CHECK_LAUNCH();
            if (via_tmp)
                for (size_t i = 0; i < count; i++)
                    ZK_HIP(hipMemcpyAsync(h_cols[i], tmp_base + i * N * 32, N * 32, hipMemcpyDeviceToDevice, ctx->stream));
        }
        t_pass.stop();
        t_pass.defer();
        blk_log -= a.r;
    }
    if (ctx->timing) { ctx->last_ms["ntt_points"] += (double)count * (double)N; ctx->last_ms["ntt_pass_points"] += (double)count * (double)N * P; }
    return ZK_OK;
}
int ntt_dev(zk_ctx* ctx, void* d_a, uint32_t log_n, const u256& omega, const NttFuse* fuse) {
    if (!d_a) return ctx->fail(ZK_ERR_ARG, "zk_ntt: null pointer");
    void* cols[1] = {d_a};
    const void* srcs[1] = {fuse && fuse->src ? fuse->src : d_a};
    return ntt_dev_batch(ctx, cols, (fuse && fuse->src) ? srcs : nullptr, 1, log_n, omega, fuse);
}

// ---- EvaluationDomain constants (host, Fr arithmetic of field.cuh compiled for the CPU) ---------
static u256 fr_const(const uint64_t (&l)[4]) {
    u256 o;
    for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)(l[i >> 1] >> (32 * (i & 1)));
    return o;
}
u256 domain_omega(uint32_t k) {  // ROOT_OF_UNITY^(2^(S-k))
    const uint64_t r[4] = BN254_FR_ROOT_OF_UNITY_M;
    return fr_pow2k_host(fr_const(r), BN254_FR_S - k);
}
u256 fr_two_inv_pow(uint32_t k) {  // (2^k)^-1
    const uint64_t ti[4] = BN254_FR_TWO_INV_M;
    u256 t = fr_const(ti), acc = Fr::one();
    for (uint32_t i = 0; i < k; i++) acc = Fr::mul(acc, t);
    return acc;
}

int domain_lagrange_to_coeff(zk_ctx* ctx, void* d_a, uint32_t k) {
    if (k > BN254_FR_S) return ctx->fail(ZK_ERR_ARG, "k = %u > S", k);
    NttFuse f;
    f.post_scale = 1; f.scale = fr_two_inv_pow(k);
    return ntt_dev(ctx, d_a, k, Fr::inv(domain_omega(k)), &f);
}
int domain_lagrange_to_coeff_batch(zk_ctx* ctx, void* const* cols, size_t count, uint32_t k) {
    if (k > BN254_FR_S) return ctx->fail(ZK_ERR_ARG, "k = %u > S", k);
    NttFuse f;
    f.post_scale = 1; f.scale = fr_two_inv_pow(k);
    return ntt_dev_batch(ctx, cols, nullptr, count, k, Fr::inv(domain_omega(k)), &f);
}
int domain_coeff_to_extended_batch(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek) {
    if (ek > BN254_FR_S || k > ek || !coeffs || !outs) return ctx->fail(ZK_ERR_ARG, "zk_coeff_to_extended: bad k/extended_k/pointer");
    NttFuse f;
    f.n_valid = 1u << k; f.pre_zeta = 1;
    return ntt_dev_batch(ctx, outs, coeffs, count, ek, domain_omega(ek), &f);
}
// the pre-scaling of a coset transform as one table: out[m] = 32 * ZETA^(m mod 3) * ext_omega^((coset * m) mod 2^ek), canonical — the operand x 2^261 of the ONE product the first
// (29-bit) pass then spends on an input, where the two-level powers cost it up to 2.67 (ZETA, lo * hi, value * power).  16 MiB per coset at k = 19, built once per context.
ZK_KERNEL void ntt_coset_table_kernel(const void* lo, const void* hi, uint32_t lo_bits, uint32_t stride, uint32_t log, u256 z0, u256 z1, u256 z2, uint32_t n, void* out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const uint32_t e = (stride * m) & ((1u << log) - 1u), h = e >> lo_bits, z = m % 3u;
    u256 w = load_u256(lo, e & ((1u << lo_bits) - 1u));
    if (h) w = Fr::mul(w, load_u256(hi, h));
    store_u256(out, m, Fr::mul(w, z == 0 ? z0 : (z == 1 ? z1 : z2)));
}
static int coset_table(zk_ctx* ctx, uint32_t k, uint32_t ek, uint32_t coset, const NttFuse& f, const void** out) {
    const uint64_t key = ((uint64_t)k << 48) | ((uint64_t)ek << 32) | coset;
    auto it = ctx->coset_tables.find(key);
    if (it == ctx->coset_tables.end()) {
        DevTmp t;
        ZK_HIP(hipMalloc(&t.p, (size_t)32 << k));
        void* const d = t.p;
        u256 c32 = Fr::zero();
        c32.v[0] = 32;
        c32 = Fr::to_mont(c32);
        ZK_LAUNCH(ntt_coset_table_kernel, (uint32_t)((((size_t)1 << k) + 255) / 256), 256, 0, ctx->stream, f.cs_lo, f.cs_hi, f.cs_lo_bits, coset, ek, c32, Fr::mul(zeta_pow(1), c32), Fr::mul(zeta_pow(2), c32),
                  (uint32_t)1 << k, d);
        if (hipGetLastError() != hipSuccess) return ctx->fail(ZK_ERR_HIP, "ntt_coset_table_kernel: launch failed");
        it = ctx->coset_tables.emplace(key, d).first;
        (void)t.release();
    }
    *out = it->second;
    return ZK_OK;
}
// evaluations of `count` polynomials (n = 2^k coefficients each) on coset `coset` of the extended domain: out[i] = f(ZETA * ext_omega^(i * 2^(ek-k) + coset)),
// i.e. every 2^(ek-k)-th entry of coeff_to_extended starting at `coset` — one size-n NTT of f_m * ZETA^(m mod 3) * ext_omega^(coset * m)
int domain_coeff_to_coset_batch(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek, uint32_t coset) {
    if (ek > BN254_FR_S || k > ek || !coeffs || !outs || coset >= (1u << (ek - k))) return ctx->fail(ZK_ERR_ARG, "zk_coeff_to_coset: bad k/extended_k/coset/pointer");
    NttFuse f;
    f.pre_zeta = 1;
    if (coset) {
        int rc = ntt_pow_tables(ctx, ek, domain_omega(ek), &f.cs_lo, &f.cs_hi, &f.cs_lo_bits);
        if (rc) return rc;
        f.cs_stride = coset; f.cs_log = ek;
        if (ctx->tune.ntt_coset_table && k > (uint32_t)ctx->tune.ntt_tile_log) {       // (a single-pass transform runs on the 32-bit form: it keeps the powers)
            rc = coset_table(ctx, k, ek, coset, f, &f.pre_full);
            if (rc) return rc;
        }
    }
    return ntt_dev_batch(ctx, outs, coeffs, count, k, domain_omega(k), &f);
}
// out[i * count + j] = cosets[j][i]: the 2^e cosets of the extended domain back into its natural (interleaved) order
ZK_KERNEL void fr_interleave_kernel(const void* const* cosets, uint32_t count, size_t n, void* out) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = n * count, stride = (size_t)gridDim.x * blockDim.x;
    for (; idx < total; idx += stride) store_u256(out, idx, load_u256(cosets[idx % count], idx / count));
}
int fr_interleave(zk_ctx* ctx, const void* const* h_cosets, size_t count, size_t n, void* d_out) {
    if (!h_cosets || !d_out || count == 0 || count > 64) return ctx->fail(ZK_ERR_ARG, "zk_fr_interleave_dev: bad argument");
    for (size_t j = 0; j < count; j++) if (!h_cosets[j]) return ctx->fail(ZK_ERR_ARG, "zk_fr_interleave_dev: null coset %zu", j);
    ZK_HIP(ctx->ws_tmp.ensure(count * sizeof(void*) + 64));
    ZK_HIP(hipMemcpyAsync(ctx->ws_tmp.p, h_cosets, count * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    const int blk = ctx->tune.vec_block;
    size_t grid = (n * count + blk - 1) / blk; if (grid > 8192) grid = 8192;
    ZK_LAUNCH(fr_interleave_kernel, (uint32_t)grid, blk, 0, ctx->stream, (const void* const*)ctx->ws_tmp.p, (uint32_t)count, n, d_out);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
// ---- h(X) from its numerator on q of the 2^(ek-k) cosets ---------------------------------------------------------------------------------------------------
// h = N / (X^n - 1) has degree below q n, q = cs_degree - 1 pieces: h = sum_i X^(n i) h_i.  On coset j of the extended domain, X = c_j w^r with c_j = ZETA ext_omega^j, X^n is
// the constant s_j = c_j^n, so the coset's values of h are those of G_j = sum_i s_j^i h_i, a polynomial of degree below n: an inverse size-n transform of N_j / (s_j - 1) gives
// c_j^m G_j[m], and for every coefficient index m the q values G_j[m] determine h_0[m] .. h_{q-1}[m] through the q x q Vandermonde matrix in s_j.  So q cosets are enough —
// the 2^(ek-k) - q others of EvaluationDomain's extended domain carry no information about h (they exist because its size is a power of two) — and the pieces come out directly:
//   piece_i[m] = sum_j W[i][j] ZETA^-(m mod 3) ext_omega^(-j m) a_j[m],   a_j = iNTT_n(N_j),   W = V^-1 diag(1 / (s_j - 1)),  V[j][i] = s_j^i.
// Same field elements as divide_by_vanishing_poly + extended_to_coeff on all cosets (h is unique), at q / 2^(ek-k) of the transforms and of the quotient rows.
ZK_KERNEL void coset_combine_kernel(const void* const* a, void* const* out, const void* wz, const void* tw_lo, const void* tw_hi, uint32_t lo_bits, uint32_t ek, uint32_t q, size_t n) {
    const uint32_t piece = blockIdx.y;
    const uint32_t mask = (1u << ek) - 1u, lomask = (1u << lo_bits) - 1u;
    size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; m < n; m += stride) {
        const uint32_t z = (uint32_t)(m % 3);
        u256 acc = Fr::zero();
        for (uint32_t j = 0; j < q; j++) {
            u256 v = load_u256(a[j], m);
            const uint32_t e = (0u - j * (uint32_t)m) & mask;          // ext_omega^(-j m)
            if (e) {
                u256 w = load_u256(tw_lo, e & lomask);
                const uint32_t h = e >> lo_bits;
                if (h) w = Fr::mul(w, load_u256(tw_hi, h));
                v = Fr::mul(v, w);
            }
            acc = Fr::add(acc, Fr::mul(v, load_u256(wz, (size_t)(z * q + piece) * q + j)));
        }
        store_u256(out[piece], m, acc);
    }
}
// numer[j]: the n numerator values on coset j (device, CLOBBERED: transformed in place), j < q;  pieces[i]: n coefficients each (device, distinct from numer)
int domain_cosets_to_pieces(zk_ctx* ctx, void* const* h_numer, uint32_t q, uint32_t k, uint32_t ek, void* const* h_pieces) {
    if (ek > BN254_FR_S || k > ek || ek - k > 6 || !h_numer || !h_pieces || q < 1 || q > (1u << (ek - k)) || q > 8)
        return ctx->fail(ZK_ERR_ARG, "zk_cosets_to_pieces_dev: bad k / extended_k / piece count (at most 8 pieces, at most one per coset)");
    for (uint32_t j = 0; j < q; j++) {
        if (!h_numer[j] || !h_pieces[j]) return ctx->fail(ZK_ERR_ARG, "zk_cosets_to_pieces_dev: null column %u", j);
        for (uint32_t i = 0; i < q; i++) if (h_numer[j] == h_pieces[i]) return ctx->fail(ZK_ERR_ARG, "zk_cosets_to_pieces_dev: pieces must not alias the numerators");
    }
    int rc = domain_lagrange_to_coeff_batch(ctx, h_numer, q, k);       // a_j (the 1 / n is in there)
    if (rc) return rc;
    // W = V^-1 diag(1 / (s_j - 1)) by Gauss-Jordan on [V | I] (q <= 8), then its three ZETA^-z multiples
    const uint64_t zl[4] = BN254_FR_ZETA_M;
    const u256 zn = fr_pow2k_host(fr_const(zl), k), won = fr_pow2k_host(domain_omega(ek), k);
    u256 V[8][16];
    u256 s = zn, sinv[8];
    for (uint32_t j = 0; j < q; j++) {
        u256 pw = Fr::one();
        for (uint32_t i = 0; i < q; i++) { V[j][i] = pw; pw = Fr::mul(pw, s); V[j][q + i] = i == j ? Fr::one() : Fr::zero(); }
        sinv[j] = Fr::inv(Fr::sub(s, Fr::one()));                       // s_j != 1: ZETA^n has order 3, (ext_omega^n)^j a power-of-two order
        s = Fr::mul(s, won);
    }
    for (uint32_t c = 0; c < q; c++) {
        uint32_t piv = c;
        while (piv < q && Fr::eq(V[piv][c], Fr::zero())) piv++;
        if (piv == q) return ctx->fail(ZK_ERR_ARG, "zk_cosets_to_pieces_dev: singular Vandermonde matrix");
        if (piv != c) for (uint32_t i = 0; i < 2 * q; i++) { const u256 t = V[c][i]; V[c][i] = V[piv][i]; V[piv][i] = t; }
        const u256 inv = Fr::inv(V[c][c]);
        for (uint32_t i = 0; i < 2 * q; i++) V[c][i] = Fr::mul(V[c][i], inv);
        for (uint32_t r = 0; r < q; r++) {
            if (r == c || Fr::eq(V[r][c], Fr::zero())) continue;
            const u256 f = V[r][c];
            for (uint32_t i = 0; i < 2 * q; i++) V[r][i] = Fr::sub(V[r][i], Fr::mul(f, V[c][i]));
        }
    }
    std::vector<u256> wz((size_t)3 * q * q);
    for (uint32_t z = 0; z < 3; z++)
        for (uint32_t i = 0; i < q; i++)
            for (uint32_t j = 0; j < q; j++) {
                u256 w = Fr::mul(V[i][q + j], sinv[j]);
                if (z) w = Fr::mul(w, zeta_pow(3 - z));                 // ZETA^-z = ZETA^(3 - z)
                wz[((size_t)z * q + i) * q + j] = w;
            }
    const void* lo; const void* hi; uint32_t lo_bits;
    rc = ntt_pow_tables(ctx, ek, domain_omega(ek), &lo, &hi, &lo_bits);
    if (rc) return rc;
    const size_t tab = wz.size() * 32, ptrs = 2 * (size_t)q * sizeof(void*);
    ZK_HIP(ctx->ws_tmp.ensure(tab + ptrs + 64));
    std::vector<const void*> pa(2 * q);
    for (uint32_t j = 0; j < q; j++) { pa[j] = h_numer[j]; pa[q + j] = h_pieces[j]; }
    ZK_HIP(hipMemcpyAsync(ctx->ws_tmp.p, wz.data(), tab, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync((char*)ctx->ws_tmp.p + tab, pa.data(), ptrs, hipMemcpyHostToDevice, ctx->stream));
    const size_t n = (size_t)1 << k;
    const int blk = ctx->tune.vec_block;
    size_t grid = (n + blk - 1) / blk; if (grid > 4096) grid = 4096;
    const void* const* d_a = (const void* const*)((char*)ctx->ws_tmp.p + tab);
    ZK_LAUNCH(coset_combine_kernel, dim3((uint32_t)grid, q), blk, 0, ctx->stream, d_a, (void* const*)(d_a + q), (const void*)ctx->ws_tmp.p, lo, hi, lo_bits, ek, q, n);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(ctx->stream));                          // wz / pa live on this frame
    return ZK_OK;
}
int domain_coeff_to_lagrange(zk_ctx* ctx, void* d_a, uint32_t k) {
    if (k > BN254_FR_S) return ctx->fail(ZK_ERR_ARG, "k = %u > S", k);
    return ntt_dev(ctx, d_a, k, domain_omega(k), nullptr);
}
int domain_coeff_to_extended(zk_ctx* ctx, const void* d_coeff, uint32_t k, uint32_t ek, void* d_out) {
    if (ek > BN254_FR_S || k > ek || !d_coeff || !d_out) return ctx->fail(ZK_ERR_ARG, "zk_coeff_to_extended: bad k/extended_k/pointer");
    NttFuse f;
    f.src = d_coeff; f.n_valid = 1u << k; f.pre_zeta = 1;
    return ntt_dev(ctx, d_out, ek, domain_omega(ek), &f);
}
int domain_extended_to_coeff(zk_ctx* ctx, void* d_a, uint32_t k, uint32_t ek) {
    if (ek > BN254_FR_S || k > ek) return ctx->fail(ZK_ERR_ARG, "zk_extended_to_coeff: bad k/extended_k");
    NttFuse f;
    f.post_scale = 1; f.scale = fr_two_inv_pow(ek); f.post_zeta_inv = 1;
    return ntt_dev(ctx, d_a, ek, Fr::inv(domain_omega(ek)), &f);
}
int domain_divide_by_vanishing(zk_ctx* ctx, void* d_a, uint32_t k, uint32_t ek) {
    if (ek > BN254_FR_S || k > ek || ek - k > 6) return ctx->fail(ZK_ERR_ARG, "zk_divide_by_vanishing_poly: bad k/extended_k");
    // t_evaluations[i] = (ZETA^n * (ext_omega^n)^i - 1)^-1
    const uint64_t z[4] = BN254_FR_ZETA_M;
    u256 zn = fr_pow2k_host(fr_const(z), k), won = fr_pow2k_host(domain_omega(ek), k);
    const uint32_t nt = 1u << (ek - k);
    std::vector<u256> t(nt);
    u256 cur = zn;
    for (uint32_t i = 0; i < nt; i++) { t[i] = Fr::inv(Fr::sub(cur, Fr::one())); cur = Fr::mul(cur, won); }
    ZK_HIP(ctx->ws_tmp.ensure(nt * 32));
    ZK_HIP(hipMemcpyAsync(ctx->ws_tmp.p, t.data(), nt * 32, hipMemcpyHostToDevice, ctx->stream));
    const size_t N = (size_t)1 << ek;
    const int blk = ctx->tune.vec_block;
    size_t grid = (N + blk - 1) / blk; if (grid > 4096) grid = 4096;
    ZK_LAUNCH(fr_mul_periodic_kernel, (uint32_t)grid, blk, 0, ctx->stream, d_a, N, (const void*)ctx->ws_tmp.p, ek - k);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(ctx->stream));  // t lives on this frame
    return ZK_OK;
}
int fr_vec_op(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n, const u256* scalar) {
    if (!a || !out || (!b && op != 3)) return ctx->fail(ZK_ERR_ARG, "vector op: null pointer");
    if (n == 0) return ZK_OK;
    const int blk = ctx->tune.vec_block;
    size_t grid = (n + blk - 1) / blk; if (grid > 4096) grid = 4096;
    u256 s = scalar ? *scalar : Fr::one();
    ZK_LAUNCH(fr_vec_kernel, (uint32_t)grid, blk, 0, ctx->stream, op, a, b ? b : a, out, n, s);
    ZK_CHECK_LAUNCH();
    return ZK_OK;
}

}  // namespace zk

namespace zk {
// power tables of omega (shared with the quotient kernel for extended_omega^idx)
int ntt_pow_tables(zk_ctx* ctx, uint32_t log_n, const u256& omega, const void** lo, const void** hi, uint32_t* lo_bits) {
    TwiddleSet* ts;
    int rc = get_twiddles(ctx, log_n, omega, nullptr, &ts);
    if (rc) return rc;
    *lo = ts->d_lo; *hi = ts->d_hi; *lo_bits = ts->lo_bits;
    return ZK_OK;
}
}  // namespace zk

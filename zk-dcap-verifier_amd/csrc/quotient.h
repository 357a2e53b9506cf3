// The quotient micro-program as the compiler (quotient.hip) leaves it and as its two executors read it: the interpreter kernel (quotient.hip) and the
// per-program straight-line kernels generated from the same micro-ops (quotient_jit.hip).
#pragma once
#include "ctx.h"

namespace zk {
// micro-ops
enum { M_ADD = 0, M_SUB, M_MUL, M_SQR, M_DBL, M_NEG, M_MOV, M_MULADD, M_FOLD2 };   // M_FOLD2: acc = acc * c + a * b with ONE Montgomery reduction (value = value * y + product)
enum { K_SLOT = 0, K_CONST, K_COL, K_ACC, K_XPOW, K_NONE = 7 };

struct QuotProgram {
    uint32_t k = 0, ek = 0, n_fixed = 0, n_advice = 0, n_instance = 0, n_challenges = 0, blinding = 0, degree = 0;
    uint32_t n_perm_cols = 0, n_sets = 0, n_lookups = 0;
    std::vector<uint32_t> perm_cols;      // pairs (type, index)
    std::vector<uint4> code;
    std::vector<u256> graph_consts;       // constants that come with the program
    std::vector<int32_t> rotations;       // distinct rotations (rows)
    uint32_t n_slots = 0, n_cols = 0;
    // constant table layout (indices)
    uint32_t c_zero = 0, c_one = 0, c_chal = 0, c_beta = 0, c_gamma = 0, c_theta = 0, c_y = 0, c_delta = 0, c_ypow = 0, n_consts = 0;
    // Degree split (compile_program, `mode`): h's numerator is sum_i y^(N-1-i) id_i over the N identities halo2 folds with y; an identity of degree d (in the columns)
    // contributes a share of h(X) of degree below (d - 1) n, which (d - 1) cosets of the size-n domain determine.  part_hi / part_lo are the SAME program restricted to the
    // identities of degree above / up to SPLIT_LOW_DEGREE (a skipped identity leaves a power of y on the next fold: ypow_exps, constants of the run at c_ypow): the low part
    // is evaluated on SPLIT_LOW_DEGREE - 1 cosets only and joins h(X) through zk_cosets_to_pieces_dev.  Exact for every witness that satisfies the circuit (each identity
    // then vanishes on the domain on its own, so both shares are polynomials).
    std::vector<uint32_t> ypow_exps;      // y^e constants this program reads, e >= 2
    uint32_t folds_taken = 0, folds_skipped = 0;
    std::shared_ptr<QuotProgram> part_hi, part_lo;
    // column ids
    uint32_t col_fixed = 0, col_advice = 0, col_instance = 0, col_l0 = 0, col_llast = 0, col_lactive = 0, col_sigma = 0, col_z = 0,
             col_lk_z = 0, col_lk_a = 0, col_lk_s = 0;
    bool uses_xpow = false;
    void* d_code = nullptr;               // immutable after the load; the constants / column pointers / rotation offsets of a RUN live in the calling context's ws_quot,
    int device = 0;                       // so contexts of one device can share a program (zk_quotient_program_share) and run it concurrently
    std::shared_ptr<struct QuotJit> jit;  // quotient_jit.hip: the same micro-ops as straight-line kernels generated for THIS program (hiprtc, tune quot_jit), or null: the interpreter runs it
    ~QuotProgram() { if (d_code) { (void)hipSetDevice(device); (void)hipFree(d_code); } }
};

#include "quot_args.inc"

// quotient_jit.hip
struct QuotJitKernel {                        // one generated kernel = a run of consecutive micro-ops
    uint32_t first = 0, count = 0;            // micro-ops [first, first + count)
    std::vector<uint32_t> live_in, live_out;  // slots whose values cross the kernel's boundaries (carried in QuotArgs::state, one row-major plane per slot)
    bool reads_acc = false;                   // the accumulator arrives from the previous kernel (in `out`, redundant form)
    std::string name;                         // zkq<program tag>_<index>
};
std::string quot_jit_source(const QuotProgram& P, uint32_t group_ops, std::vector<QuotJitKernel>* kernels, uint32_t waves_per_eu = 0);
int quot_jit_build(zk_ctx* ctx, QuotProgram& P);                                   // tune quot_jit: compile P (and its parts) into kernels; failure leaves the interpreter in charge
bool quot_jit_ready(const QuotProgram& P);
uint32_t quot_jit_kernel_count(const QuotProgram& P);
int quot_jit_launch(zk_ctx* ctx, const QuotProgram& P, const QuotArgs& q, uint64_t rows, uint32_t threads);
}  // namespace zk

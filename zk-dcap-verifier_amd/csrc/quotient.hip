// placeholder until the interpreter lands (same commit series)
#include "ctx.h"
namespace zk {
struct QuotProgram { int dummy; };
int quotient_program_load(zk_ctx* ctx, const void*, size_t, uint64_t*) { return ctx->fail(ZK_ERR_PROGRAM, "quotient: not built yet"); }
int quotient_program_release(zk_ctx* ctx, uint64_t) { return ctx->fail(ZK_ERR_ARG, "quotient: unknown program"); }
int quotient_run(zk_ctx* ctx, uint64_t, const zk_quotient_args*) { return ctx->fail(ZK_ERR_PROGRAM, "quotient: not built yet"); }
void release_programs(zk_ctx*) {}
}

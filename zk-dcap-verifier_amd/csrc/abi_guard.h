// The exception barrier of the C ABI and the test-only fault hooks behind it (every translation unit that defines an extern "C" entry point includes this).
#pragma once
#include "../../include/zkmi355.h"

// ---- the exception barrier of the C ABI (include/zkmi355.h: "nothing throws or aborts") ---------------------------------------------------------------------
// Every extern "C" function of the library is a function-try-block: `int zk_x(zk_ctx* ctx, ...) ZK_ABI_TRY { ... } ZK_ABI_CATCH(ctx)`.  The host side allocates
// (std::vector, std::map, std::function, std::thread): std::bad_alloc becomes ZK_ERR_LIMIT, std::system_error (no thread, no lock) and anything else ZK_ERR_HIP, an
// AbiError carries its own code; the text goes to zk_last_error.  Locks and proof arenas are RAII, so the unwinding leaves the context usable for the next call.
// tests/test_abi_no_throw.py greps that no extern "C" definition is without it and drives the fault hooks below through the emulator build.
namespace zk {
struct AbiError { int code; const char* what; };                      // thrown by library code that sits too deep to return a code (a helper thread's failure surfacing in its consumer)
int abi_exception(zk_ctx* ctx, const char* fn) noexcept;              // capi.hip; call from a catch (...) handler only
// TEST-ONLY fault hooks (compiled into the emulator build alone: -DZK_FAULT_INJECT; the product has neither the counters nor the operator new that reads them)
#ifdef ZK_FAULT_INJECT
void fault_thread_tick();                                             // throws std::system_error when the calling thread armed a thread failure (zk_test_fail_thread)
#else
inline void fault_thread_tick() {}
#endif
}  // namespace zk
#define ZK_ABI_TRY try
#define ZK_ABI_CATCH(ctx_) catch (...) { return zk::abi_exception((ctx_), __func__); }
#define ZK_ABI_CATCH_VALUE(ctx_, value) catch (...) { (void)zk::abi_exception((ctx_), __func__); return value; }
#define ZK_ABI_CATCH_VOID(ctx_) catch (...) { (void)zk::abi_exception((ctx_), __func__); }

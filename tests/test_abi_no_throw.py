"""include/zkmi355.h promises "All functions return 0 or a negative error code; nothing throws or aborts" (the Rust functions the library stands in
for are infallible by signature — circuits/src/sgx_dcap_verifier.rs:814-822 — so an exception that left an extern "C" frame would be std::terminate inside the
prover's process).  Three checks:
  1. mechanical: every function the header declares is DEFINED as a function-try-block with the barrier of csrc/abi_guard.h, and no extern "C" definition is without it;
  2. plain C, emulator build with its fault hooks (tests/csrc/capi_faults.c): injected std::bad_alloc / std::system_error in zk_quotient_program_load,
     zk_plonk_pk_build, zk_plonk_prove (calling thread, helper threads, thread starts) -> negative codes, the same context proves the golden bytes afterwards,
     every device buffer returned;
  3. the same through the Python binding on the sgx-shaped circuit (the program a real key compiles)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, EMU_SO

CSRC = os.path.join(ROOT, "zk-dcap-verifier_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
EMU_TUNE = "msm_sort_threads=64,msm_sort_wgs=3,msm_block=32,ntt_threads=32,ntt_tile_log=6,ntt_max_radix_log=4,msm_target_threads=64,msm_min_chunk=2,vec_block=32,quot_threads=32"


def _declared(header):
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
    return set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", txt)) - {"zk_rng_fn", "zk_allgather_fn"}


def test_every_extern_c_function_has_the_exception_barrier():
    defined, bare = {}, []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        src = open(os.path.join(CSRC, f)).read()
        in_block = [(m.end(), src.index('\n}  // extern "C"', m.end())) for m in re.finditer(r'^extern "C" \{', src, re.M)]
        for m in re.finditer(r'^(extern "C" )?(?:int|double|void|long|uint32_t|uint64_t|const char\*) (zk_\w+)\(', src, re.M):
            if not (m.group(1) or any(a <= m.start() < b for a, b in in_block)):
                continue
            depth, j = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(src[j], 0)
                j += 1
            rest = src[j:j + 40].lstrip()
            if rest.startswith(";"):
                continue                                               # a declaration
            name = m.group(2)
            if name.startswith("zk_test_"):
                continue                                               # the emulator build's fault hooks (not in the header, set two counters)
            if rest.startswith("ZK_ABI_TRY"):
                defined[name] = f
            else:
                bare.append(f"{f}: {name}")
    assert not bare, "extern \"C\" definitions without ZK_ABI_TRY ... ZK_ABI_CATCH: " + ", ".join(bare)
    want = _declared("zkmi355.h")
    rccl = os.path.join(ROOT, "include", "zkmi355_rccl.h")
    if os.path.exists(rccl):
        want |= _declared("zkmi355_rccl.h")
    missing = sorted(want - set(defined))
    assert not missing, "declared in include/ but not defined behind the barrier: " + ", ".join(missing)
    for f in sorted(os.listdir(CSRC)):                                 # and the barrier is the only try/catch idiom at the boundary: count them
        if f.endswith(".hip"):
            src = "\n".join(l for l in open(os.path.join(CSRC, f)).read().splitlines() if not l.startswith("#define"))
            assert src.count("ZK_ABI_TRY") == len(re.findall(r"ZK_ABI_CATCH(?:_VALUE|_VOID)?\(", src)), f


def _build_capi_faults():
    exe = os.path.join(ROOT, "tests", "csrc", "capi_faults")
    src = os.path.join(ROOT, "tests", "csrc", "capi_faults.c")
    deps = [src, os.path.join(ROOT, "tests", "csrc", "zkpk_reader.h"), EMU_SO]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        libdir = os.path.join(ROOT, "tests", "csrc")
        subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-lzkmi355_emu",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,$ORIGIN"])
    return exe


@pytest.mark.parametrize("which", ["toy", "sgx"])
def test_injected_host_failures_become_error_codes_plain_c(emu, orc, tmp_path, which):
    import dump_pk_blob as dp
    exe = _build_capi_faults()
    path = tmp_path / "pk.zkpk"
    path.write_bytes(dp.toy_blob(emu, 6, 7) if which == "toy" else dp.sgx_blob(emu, 7, 3, "chip_estimate"))
    env = dict(os.environ, ZK_TUNE=EMU_TUNE)
    if os.environ.get("ZK_EMU_LIBDIR"):                               # tests/run_sanitizers.sh: the same program against a sanitizer build of the emulator library
        env["LD_LIBRARY_PATH"] = os.environ["ZK_EMU_LIBDIR"] + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and "capi_faults OK" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    assert "no side-lane thread -> one lane" in r.stdout
    assert "-> ZK_ERR_LIMIT each" in r.stdout or os.environ.get("ZK_SANITIZER") in ("asan", "tsan")      # (their runtimes own operator new: no allocation ladder there)


def test_injected_host_failures_through_the_python_binding(emu, orc):
    """zk_quotient_program_load on the sgx-shaped key's own ZKQ1 program and zk_plonk_create_proof through plonk.NativeProver: a failing allocation raises the
    binding's error with the barrier's text; the next proof on the same backend is the golden."""
    import zk_dcap_verifier_amd as z
    from zk_dcap_verifier_amd import plonk
    import sgx_shaped_circuit as sc
    import test_create_proof as tcp
    lib = emu.lib
    if not lib.zk_test_alloc_hook_present():
        pytest.skip("a sanitizer build of the emulator library: the sanitizer's runtime owns operator new")
    lib.zk_test_fail_alloc.argtypes = [C.c_long]
    lib.zk_test_fail_alloc.restype = None
    k = 8
    cs, fixed, asm, advice = sc.build(z, emu, k, census="chip_estimate")
    params = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=emu)
    pk = plonk.keygen(params, cs, fixed, asm)
    native = plonk.NativeProver(params, pk)
    golden = tcp._golden(tcp.GOLDEN_SGX)
    hit = 0
    for nth in (1, 2, 5, 17, 60, 200, 700):
        lib.zk_test_fail_alloc(nth)
        try:
            proof = native.create_proof(advice, [], np.random.default_rng(3))
            lib.zk_test_fail_alloc(0)
            assert proof == golden                                     # (fewer allocations than nth on the calling thread)
        except z.ZkError as e:
            lib.zk_test_fail_alloc(0)
            hit += 1
            assert e.code == -5 and "out of host memory" in str(e), (e.code, str(e))
        assert native.create_proof(advice, [], np.random.default_rng(3)) == golden
    assert hit >= 4
    pk.release()
    params.release()

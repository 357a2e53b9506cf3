"""The quotient program as GENERATED kernels (csrc/quotient_jit.hip, tune quot_jit): the same micro-ops as the interpreter of csrc/quotient.hip, one straight-line
statement each on the same field functions, cut into kernels, compiled by hiprtc when the program is loaded.
CPU: the generator's output for the toy circuit's program and for a random program is valid HIP for gfx950 (hipcc cross-compiles it: no GPU, no hiprtc), every
micro-op became a statement, and the kernels' carried state is consistent (what one kernel leaves is what the next one loads).
GPU: random programs of the quotient test shapes against the ORACLE through the generated kernels — whole domain, cosets, row slices, the degree parts — and the toy
and sgx-shaped golden proofs byte for byte with the tunable on."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk

import parity_cases as pc
import quotient_cases as qc
import test_create_proof as tcp
from conftest import ROOT

CSRC = os.path.join(ROOT, "zk-dcap-verifier_amd", "csrc")


def _source(be, handle, part, group):
    lib = be.lib
    lib.zk_test_quot_jit_source.restype = C.c_long
    nk, npl = C.c_uint32(), C.c_uint32()
    buf = C.create_string_buffer(16 << 20)
    n = lib.zk_test_quot_jit_source(be.ctx, C.c_uint64(handle), C.c_int(part), C.c_uint32(group), buf, C.c_size_t(len(buf)), C.byref(nk), C.byref(npl))
    assert 0 < n < len(buf), n
    return buf.value.decode(), nk.value, npl.value


@pytest.mark.parametrize("group", [4, 200])
def test_generated_source_is_valid_hip_for_gfx950(emu, orc, pyref, tmp_path, group):
    prog = qc.build_program(orc, pyref, seed=3, gate_ops=24, k=6, cs_degree=5, n_fixed=3, n_advice=4, n_instance=1, n_challenges=1, n_perm=5, n_lookups=2)
    h = emu.quotient_program_load(prog.to_blob())
    info = emu.quotient_program_info(h)
    src, n_kernels, n_planes = _source(emu, h, 0, group)
    assert src.count("extern \"C\" __global__") == n_kernels and n_kernels >= (2 if group == 4 else 1)
    assert len(re.findall(r"^    \{ const u256 t = ", src, re.M)) == info["instructions"], "every micro-op is one statement"
    # what a kernel stores for its successors is what they load: PLANE(s) loads only of slots some earlier kernel stored
    stored = set()
    for body in src.split('extern "C" __global__')[1:]:
        loads = set(int(m) for m in re.findall(r"= load_u256\(PLANE\((\d+)\), oidx\)", body))
        assert loads <= stored, (loads, stored)
        stored |= set(int(m) for m in re.findall(r"store_u256\(PLANE\((\d+)\), oidx", body))
    assert (max(stored) + 1 if stored else 0) == n_planes
    path = tmp_path / "zkq.hip"
    path.write_text(src)
    r = subprocess.run(["hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "--cuda-device-only", "-I", CSRC, "-c", str(path), "-o", str(tmp_path / "zkq.co")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    emu.quotient_program_release(h)


@pytest.mark.gpu
def test_generated_kernels_against_the_oracle_gpu(gpu, orc, pyref):
    """one of tests/test_quotient.py's random-program shapes (gates with rotations, a two-set permutation, two lookups, an instance column, a challenge): whole domain, every
    coset, row slices and both degree parts, each against the oracle — through kernels cut after every 6 products, so that accumulator and slots cross many boundaries"""
    gpu.tune(quot_jit=2, quot_jit_group=6)          # 2: the whole program AND its degree parts (1 leaves the whole program of a split key on the interpreter)
    try:
        prog = qc.build_program(orc, pyref, seed=5, gate_ops=24, k=8, cs_degree=5, n_fixed=4, n_advice=6, n_instance=1, n_challenges=1, n_perm=7, n_lookups=3)
        qc.run_case(gpu, orc, pyref, pc, prog, seed=5, expect_kernels=True)
    finally:
        gpu.tune(quot_jit=0, quot_jit_group=200)


@pytest.mark.gpu
def test_golden_proofs_through_generated_kernels_gpu(gpu, orc):
    import verifier
    gpu.tune(quot_jit=1)
    try:
        cs, fixed, asm, advice, instances = tcp.toy_circuit(6)
        params = z.kzg.ParamsKZG.setup(6, tcp.TAU, backend=gpu)
        pk = plonk.keygen(params, cs, fixed, asm)
        assert gpu.quotient_program_kernels(pk.evaluator.handle) >= 1
        proof = plonk.NativeProver(params, pk).create_proof([a.copy() for a in advice], instances, np.random.default_rng(7))
        assert proof == tcp._golden(tcp.GOLDEN_PROOF)
        assert verifier.verify_proof(pk.vk, tcp.TAU, instances, proof) is True
        pk.release()
        params.release()
    finally:
        gpu.tune(quot_jit=0)


@pytest.mark.gpu
def test_level_one_generates_the_degree_parts_only_gpu(gpu, orc, pyref):
    """quot_jit = 1 on a split program: kernels for the high and low parts (what a single-GPU proof launches), the whole program stays on the interpreter — and still
    gives the oracle's values; quot_jit = 2 adds the whole program's kernels"""
    from zk_dcap_verifier_amd import evaluation as ev
    prog = qc.build_program(orc, pyref, seed=11, gate_ops=24, k=6, cs_degree=5, n_fixed=3, n_advice=4, n_instance=1, n_challenges=1, n_perm=5, n_lookups=2)
    counts = {}
    try:
        for level in (1, 2):
            gpu.tune(quot_jit=level, quot_jit_group=8)
            e = ev.Evaluator(prog, backend=gpu)
            assert gpu.quotient_program_split(e.handle)["low_cosets"] == 2
            counts[level] = gpu.quotient_program_kernels(e.handle)
            e.release()
        assert 2 <= counts[1] < counts[2], counts
        gpu.tune(quot_jit=1, quot_jit_group=8)
        qc.run_case(gpu, orc, pyref, pc, prog, seed=11, expect_kernels=True)        # whole domain and cosets on the interpreter, the parts on their kernels
    finally:
        gpu.tune(quot_jit=0, quot_jit_group=200)

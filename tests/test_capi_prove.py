"""The one-call boundary WITHOUT the Python mirror in the proving loop: tools/dump_pk_blob.py writes the host data of a proving key, SRS,
witness and rng stream into one flat file; tests/csrc/capi_prove.c (gcc -std=c99) rebuilds the key with zk_plonk_pk_build, proves with zk_plonk_prove — first on
one context, then on a second context that borrows tables and key — and must reproduce the golden proof of the independent CPU prover (oracle/prover.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
EMU_TUNE = "msm_sort_threads=64,msm_sort_wgs=3,msm_block=32,ntt_threads=32,ntt_tile_log=6,ntt_max_radix_log=4,msm_target_threads=64,msm_min_chunk=2,vec_block=32,quot_threads=32"


def _run(exe, path, tune=None, world=None, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    if tune:
        env["ZK_TUNE"] = tune
    if exe.endswith("_emu") and os.environ.get("ZK_EMU_LIBDIR"):        # tests/run_sanitizers.sh: the same program against a sanitizer build of the emulator library
        env["LD_LIBRARY_PATH"] = os.environ["ZK_EMU_LIBDIR"] + ":" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run([os.path.join(ROOT, "tests", "csrc", exe), str(path)] + ([str(world)] if world else []), capture_output=True, text=True, timeout=900, env=env)


@pytest.mark.parametrize("which", ["toy", "sgx", "p256"])
def test_plain_c_prover_reproduces_the_goldens_emulated(emu, orc, tmp_path, which):
    """("p256": a degree-4 circuit — zk_plonk_pk_build keeps three cosets of the key instead of its extended forms and the prover takes h(X) from them: tests/test_piece_cosets.py)"""
    import dump_pk_blob as dp
    blob = dp.toy_blob(emu, 6, 7) if which == "toy" else dp.p256_blob(emu, 7, 18) if which == "p256" else dp.sgx_blob(emu, 8, 3, "chip_estimate")
    path = tmp_path / "pk.zkpk"
    path.write_bytes(blob)
    r = _run("capi_prove_emu", path, EMU_TUNE)
    assert r.returncode == 0 and "capi_prove OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("world", [2, 8])
def test_plain_c_prover_sharded_over_ranks_emulated(emu, orc, tmp_path, world):
    """W ranks as W threads of the C program, each with 1 / W of both SRS tables and a sharded key built by zk_plonk_pk_build: 2 ranks own two cosets of the quotient
    each, 8 ranks half a coset; the zk_allgather_fn is a barrier + device copies.  Every rank must emit the golden proof."""
    import dump_pk_blob as dp
    path = tmp_path / "pk.zkpk"
    path.write_bytes(dp.toy_blob(emu, 6, 7))
    r = _run("capi_prove_emu", path, EMU_TUNE, world)
    assert r.returncode == 0 and f"{world} ranks" in r.stdout and "8 all-gathers per proof" in r.stdout and "capi_prove OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_plain_c_prover_a_failing_rank_tells_the_others_through_library_owned_buffers_emulated(emu, orc, tmp_path):
    """the poisoned block of a failing rank travels in exchange buffers the LIBRARY owns (no xchg_send / xchg_recv from the caller) — they must outlive the
    failed body.  Four ranks, the last one's witness leaves its lookup table after the first exchange: it returns its own error, the other three ZK_ERR_COMM from the
    same exchange, and the untampered proof that follows on the same contexts is the golden."""
    import dump_pk_blob as dp
    path = tmp_path / "pk.zkpk"
    path.write_bytes(dp.toy_blob(emu, 6, 7))
    r = _run("capi_prove_emu", path, EMU_TUNE, 4, {"ZK_TAMPER_LAST_RANK": "1"})
    assert r.returncode == 0 and "tampered round" in r.stdout and "4 ranks" in r.stdout and "capi_prove OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_plain_c_prover_has_no_cpu_fallback(emu, orc, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dump_pk_blob as dp
    path = tmp_path / "pk.zkpk"
    path.write_bytes(dp.toy_blob(emu, 5, 1))
    r = _run("capi_prove", path)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("which,k", [("toy", 6), ("sgx", 8), ("sgx", 12), ("p256", 10)])
def test_plain_c_prover_on_gpu(gpu, orc, tmp_path, which, k):
    import dump_pk_blob as dp
    blob = dp.toy_blob(gpu, 6, 7) if which == "toy" else dp.p256_blob(gpu, k, 18) if which == "p256" else dp.sgx_blob(gpu, k, 3)
    path = tmp_path / "pk.zkpk"
    path.write_bytes(blob)
    r = _run("capi_prove", path, world=4 if k <= 8 else None)       # then 4 ranks (4 contexts on the one GPU): a coset of the quotient each
    assert r.returncode == 0 and "capi_prove OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_plain_c_prover_eight_ranks_on_gpu(gpu, orc, tmp_path):
    """BASELINE configs[4]'s rank count on the one GPU a builder has: 8 contexts = 8 ranks of the plain-C prover at k = 12 with the sgx census, each with 1/8 of both SRS
    tables and a sharded key; the quotient's units are HALF cosets (8 ranks, 4 cosets), every commitment phase one all-gather of 128-byte points, the numerators one more:
    8 collectives per proof, every rank the single-GPU prover's bytes.  What an 8-GPU node adds to this is RCCL as the collective, nothing else."""
    import dump_pk_blob as dp
    path = tmp_path / "pk.zkpk"
    path.write_bytes(dp.sgx_blob(gpu, 12, 3))
    r = _run("capi_prove", path, world=8)
    assert r.returncode == 0 and "8 ranks" in r.stdout and "8 all-gathers per proof" in r.stdout and "capi_prove OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])

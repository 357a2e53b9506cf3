#!/usr/bin/env python3
"""Regenerate tests/golden/toy_proof_k6_seed7.bin: the proof bytes of tests/test_create_proof.py's toy circuit at k = 6 under
np.random.default_rng(7) and the SRS trapdoor TAU of that test, produced by plonk.create_proof on the kernel EMULATOR (the product's
kernel sources run on CPU threads, tests/csrc/emu_rt.h).  It is a regression + CPU/GPU bit-exactness pin of OUR prover (the GPU must
emit these exact bytes, and the pure-Python verifier must accept them) — not an output of the reference, which holds no proof for
stack A and cannot be built here."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import zk_dcap_verifier_amd as z  # noqa: E402
import test_create_proof as t  # noqa: E402
from conftest import EMU_SO  # noqa: E402


def main():
    be = z.Backend(0, lib_path=EMU_SO)
    be.tune(msm_sort_threads=64, msm_sort_wgs=3, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4,
            msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    vk, instances, proof, info = t.prove(be, 6, seed=7)
    import verifier
    assert verifier.verify_proof(vk, t.TAU, instances, proof) is True
    out = os.path.join(ROOT, "tests", "golden", "toy_proof_k6_seed7.bin")
    open(out, "wb").write(proof)
    print("wrote", out, len(proof), "bytes;", info["commitments"], "commitments,", info["evals"], "evaluations")


if __name__ == "__main__":
    main()

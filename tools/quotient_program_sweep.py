#!/usr/bin/env python3
"""A wider sweep of the quotient compiler than the test suite runs: random constraint-system shapes and random gate / lookup graphs (tests/quotient_cases.py), the compiled
program against the oracle's evaluate_h on every row, coset by coset, on row slices — and the degree split (high + low = whole numerator, extended and coset layouts) and the
common-factor grouping with it, since both are on by default.  usage: python tools/quotient_program_sweep.py [first_seed=1000] [count=120] [jit]   (GPU box; ZK_LIB=<emulator .so> for the CPU)
With `jit` every program runs through the GENERATED kernels (csrc/quotient_jit.hip, tune quot_jit = 2: whole program and degree parts; GPU only: hiprtc), cut after a random number of products per kernel, and the
sweep fails if a program fell back to the interpreter."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import zk_dcap_verifier_amd as z  # noqa: E402
import oracle as orc  # noqa: E402   (checker only)
import pyref  # noqa: E402
import parity_cases as pc  # noqa: E402
import quotient_cases as qc  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
jit = len(sys.argv) > 3 and sys.argv[3] == "jit"
lib = os.environ.get("ZK_LIB") or None
be = z.Backend(0, lib)
if lib and "emu" in lib:
    be.tune(quot_threads=32, vec_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4)
bad, split, shapes = 0, 0, []
for seed in range(first, first + count):
    rnd = random.Random(seed)
    shape = dict(k=rnd.choice([2, 3, 3, 4, 5]), cs_degree=rnd.choice([3, 4, 4, 5, 5, 5, 6, 7, 9]), n_fixed=rnd.randrange(1, 5), n_advice=rnd.randrange(1, 7),
                 n_instance=rnd.randrange(0, 3), n_challenges=rnd.randrange(0, 3), n_perm=rnd.randrange(0, 9), n_lookups=rnd.randrange(0, 4))
    try:
        prog = qc.build_program(orc, pyref, seed=seed, gate_ops=rnd.choice([6, 12, 24, 40, 60, 80]), **shape)
        if jit:
            be.tune(quot_jit=2, quot_jit_group=rnd.choice([3, 6, 24, 200]))
        e = z.evaluation.Evaluator(prog, backend=be)
        split += 1 if be.quotient_program_split(e.handle)["low_cosets"] else 0
        e.release()
        qc.run_case(be, orc, pyref, pc, prog, seed=seed, **(dict(expect_kernels=1) if jit else {}))
        if (seed - first) % 10 == 9:
            print(f"  .. {seed - first + 1} programs, failures so far = {bad}", flush=True)
    except AssertionError as ex:
        bad += 1
        print("FAIL", seed, shape, str(ex)[:300], flush=True)
    except Exception as ex:
        bad += 1
        print("ERR", seed, shape, repr(ex)[:300], flush=True)
print(f"quotient programs{' through generated kernels' if jit else ''}: seeds {first}..{first + count - 1}, {split} of them compiled with a degree split, failures = {bad}")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""A wider differential sweep than the test suite runs: random constraint systems (tests/random_circuits.py), native prover (and every 5th seed its Python twin)
against the independent CPU prover, byte for byte.  usage: python tools/random_circuit_sweep.py [first_seed=200] [count=90] [jit]   (on a GPU box; ~40 s for 90 seeds; `jit`: every key's quotient program
as generated kernels — tune quot_jit = 1, hiprtc, GPU only — so the same bytes must come out of the code-generated evaluate_h)
ZK_LIB=<path to libzkmi355_emu.so> runs it on the kernel emulator instead (tests only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import zk_dcap_verifier_amd as z  # noqa: E402
import test_random_circuits as t  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 200
count = int(sys.argv[2]) if len(sys.argv) > 2 else 90
be = z.Backend(0, os.environ.get("ZK_LIB") or None)
if len(sys.argv) > 3 and sys.argv[3] == "jit":
    be.tune(quot_jit=1, quot_jit_group=12)
bad = 0
for seed in range(first, first + count):
    try:
        t._check(be, 6 + seed % 4, seed, twin=(seed % 5 == 0))
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:300])
    except Exception as e:
        bad += 1
        print("ERR", seed, repr(e)[:300])
print(f"random circuits: seeds {first}..{first + count - 1}, failures = {bad}")
sys.exit(1 if bad else 0)

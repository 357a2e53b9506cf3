#!/usr/bin/env python3
"""GPU-box probe: single-column MSM at several sizes with the one-level and the two-level bucket sort forced (msm_two_level_sort = 0 / 1)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
from perf_probe import rand_fr


def main():
    be = z.Backend(0)
    for lg in [int(x) for x in os.environ.get("MSM_LOGS", "19,21,22,24").split(",")]:
        n = 1 << lg
        dk, dp = be.to_device(rand_fr(n, 1)), be.alloc(n * 64)
        be.g1_fixed_base_mul(dk, n, dp)
        h = be.bases_register((dp, n))
        ds = be.to_device(rand_fr(n, 2))
        for two in (0, 1):
            be.tune(msm_two_level_sort=two)
            be.msm(h, ds, n)
            be.timing(True)
            t = time.time()
            for _ in range(3):
                be.msm(h, ds, n)
            dt = (time.time() - t) / 3
            lab = {k: be.timing_get(k) for k in ("msm_sort", "msm_accumulate", "msm_reduce")}
            be.timing(False)
            print(json.dumps({"log_n": lg, "two_level": two, "ms": round(dt * 1e3, 3), "Mscalar/s": round(n / dt / 1e6, 1),
                              "kernels_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in lab.items() if v[0] is not None}}), flush=True)
        be.tune(msm_two_level_sort=2)
        be.bases_release(h)
        for d in (dk, dp, ds):
            d.free()


main()

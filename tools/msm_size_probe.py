#!/usr/bin/env python3
"""GPU-box probe: one MSM at 2^20 / 2^22 / 2^24 with the event timers on — wall time and its sort / bucket chain / reduction split (the timers add ~10 % at 2^24).  usage: msm_size_probe.py"""
import sys, time, json, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd.fields import rand_fr_array
be = z.Backend(0)
res = {}
for logn in (20, 22, 24):
    n = 1 << logn
    rng = np.random.default_rng(logn)
    ks = be.to_device(rand_fr_array(rng, n)); pts = be.alloc(n * 64)
    be.g1_fixed_base_mul(ks, n, pts)
    h = be.bases_register((pts, n)); pts.free()
    ks.upload(rand_fr_array(rng, n))
    be.msm(h, ks, n)
    be.timing(True)
    t = time.time()
    for _ in range(3): be.msm(h, ks, n)
    dt = (time.time() - t) / 3 * 1e3
    res[logn] = {"ms": round(dt, 3), "Mscalar_per_s": round(n / dt / 1e3, 1), **{l: round((be.timing_get(l)[0] or 0) / 3, 3) for l in ("msm_sort", "msm_accumulate", "msm_reduce")}}
    be.timing(False)
    be.bases_release(h); ks.free()
print(json.dumps(res))

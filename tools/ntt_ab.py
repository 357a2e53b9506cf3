#!/usr/bin/env python3
"""A/B of the prover's NTT phases between library builds / tune settings on one box: lagrange_to_coeff and coeff_to_extended of 64 columns at k = 19,
checked for equal results.  usage: ntt_ab.py [k] [cols] -- each further argument is `label:lib_path_or_-:key=val,key=val`"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd.fields import rand_fr_array

k, cols = int(sys.argv[1]), int(sys.argv[2])
ek, n = k + 2, 1 << k
rng = np.random.default_rng(1)
host = [rand_fr_array(rng, n) for _ in range(cols)]
ref = None
for spec in sys.argv[3:]:
    label, lib, tune = spec.split(":")
    be = z.Backend(0, lib_path=None if lib == "-" else os.path.join(ROOT, lib))
    if tune:
        be.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in tune.split(",")})
    src = [be.to_device(h) for h in host]
    ext = [be.alloc(32 << ek) for _ in range(cols)]
    def t(f, reps=5):
        f(); be.sync()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        be.sync()
        return (time.perf_counter() - t0) / reps * 1e3
    t_ext = t(lambda: be.coeff_to_extended_batch_dev(src, ext, k, ek))
    got = [ext[0].download((1 << ek, 4)), ext[cols - 1].download((1 << ek, 4))]
    t_cos = t(lambda: be.coeff_to_coset_batch_dev(src, ext, k, ek, 1))
    got.append(ext[1].download((n, 4)))
    t_l2c = t(lambda: be.lagrange_to_coeff_batch_dev(src, k), reps=6)          # (in place: 7 applications in all)
    got.append(src[2].download((n, 4)))
    if ref is None: ref = got
    same = all((a == b).all() for a, b in zip(ref, got))
    print("%-28s coeff_to_extended x%d %7.3f ms   coeff_to_coset(1) %7.3f ms   lagrange_to_coeff %7.3f ms   results %s" % (label, cols, t_ext, t_cos, t_l2c, "equal" if same else "differ"), flush=True)
    for b in src + ext: b.free()
    be.close()

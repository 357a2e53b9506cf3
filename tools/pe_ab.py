#!/usr/bin/env python3
"""A/B of the SHPLONK linear combination (60 polynomials of 2^19) and the evaluation phase (175 evaluations) between library builds on one box, results compared.  usage: pe_ab.py label:lib_or_- ..."""
import os, sys, time
import numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd.fields import rand_fr_array
n, cols = 1 << 19, 60
rng = np.random.default_rng(1)
host = [rand_fr_array(rng, n) for _ in range(cols)]
sc = rand_fr_array(rng, cols)
pts = rand_fr_array(rng, 175)
ref = None
for spec in sys.argv[1:]:
    label, lib = spec.split(":")
    be = z.Backend(0, lib_path=None if lib == "-" else os.path.join(ROOT, lib))
    dev = [be.to_device(h) for h in host]
    out = be.alloc(n * 32)
    def t(f, reps=10):
        f(); be.sync(); t0 = time.perf_counter()
        for _ in range(reps): f()
        be.sync(); return (time.perf_counter() - t0) / reps * 1e3
    t_l = t(lambda: be.fr_lincomb_dev(dev, sc, n, out))
    got_l = out.download((n, 4))
    polys = [dev[i % cols] for i in range(175)]
    t_e = t(lambda: be.eval_polynomial_batch_dev(polys, n, pts))
    got_e = be.eval_polynomial_batch_dev(polys, n, pts)
    if ref is None: ref = (got_l, got_e)
    same = (ref[0] == got_l).all() and (np.asarray(ref[1]) == np.asarray(got_e)).all()
    print("%-8s lincomb x%d %.3f ms   eval x175 %.3f ms   %s" % (label, cols, t_l, t_e, "equal" if same else "DIFFER"), flush=True)
    be.close()

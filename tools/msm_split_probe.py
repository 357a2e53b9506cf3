#!/usr/bin/env python3
"""GPU-box probe: does a batch of commitments finish sooner as TWO half-batches on two contexts at once (the sort of one half beside the bucket chain of the other) than as
one batch?  k = 19, `cols` full-width columns, a shared table.  usage: msm_split_probe.py [k=19] [cols=24]"""
import os, sys, time, threading, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd.fields import rand_fr_array
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 1 << k
a, b = z.Backend(0), z.Backend(0)
rng = np.random.default_rng(1)
ks = a.to_device(rand_fr_array(rng, n))
pts = a.alloc(n * 64)
a.g1_fixed_base_mul(ks, n, pts)
ha = a.bases_register((pts, n))
hb = b.bases_share(a, ha)
data = [a.to_device(rand_fr_array(rng, n)) for _ in range(cols)]
def one():
    return a.msm_batch(ha, data, n)
def two():
    out = [None, None]
    def run(be, h, part, i):
        out[i] = be.msm_batch(h, part, n)
    t1 = threading.Thread(target=run, args=(a, ha, data[: cols // 2], 0)); t2 = threading.Thread(target=run, args=(b, hb, data[cols // 2:], 1))
    t1.start(); t2.start(); t1.join(); t2.join()
    return np.concatenate(out)
def seq():
    return np.concatenate([a.msm_batch(ha, data[: cols // 2], n), a.msm_batch(ha, data[cols // 2:], n)])
res = {}
ref = one()
for name, f in (("one_batch", one), ("two_halves_at_once", two), ("two_halves_in_sequence", seq), ("one_batch_again", one), ("two_halves_at_once_again", two)):
    assert (f() == ref).all(), name
    a.sync(); b.sync()
    t = time.time()
    for _ in range(5):
        f()
    a.sync(); b.sync()
    res[name] = round((time.time() - t) / 5 * 1e3, 3)
print(json.dumps({"k": k, "columns": cols, "ms": res}))

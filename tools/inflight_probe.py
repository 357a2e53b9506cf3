#!/usr/bin/env python3
"""GPU-box probe: proofs/hour with 1 vs 2 proofs in flight on one GPU (two contexts = two HIP streams, two host threads)."""
import os, sys, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
import bench

def main():
    res = {}
    for inflight in (1, 2, 3):
        bes = [z.Backend(0) for _ in range(inflight)]
        wls = [bench.ProofWorkload(z, be, 19, 25, 18, 11, 16, 5) for be in bes]
        for w in wls: w.step()
        steps = 4
        def worker(w):
            for _ in range(steps): w.step()
        t = time.time()
        ths = [threading.Thread(target=worker, args=(w,)) for w in wls]
        for th in ths: th.start()
        for th in ths: th.join()
        for be in bes: be.sync()
        dt = time.time() - t
        res[inflight] = {"ms_per_proof": round(dt / (steps * inflight) * 1e3, 2), "proofs_per_hour": round(3600 * steps * inflight / dt)}
        print(json.dumps({inflight: res[inflight]}), flush=True)
        del wls
        for be in bes: be.close()

if __name__ == "__main__":
    main()

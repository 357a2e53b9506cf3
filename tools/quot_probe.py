#!/usr/bin/env python3
"""GPU-box probe: quotient kernel time for the sgx-shaped program vs threads per workgroup."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
import bench

def main():
    be = z.Backend(0)
    wl = bench.ProofWorkload(z, be, 19, 25, 18, 11, 16, 5)
    wl.step()
    A, L, P = wl.A, wl.L, wl.P
    adv, zs, lk = wl.ext_dyn[:A], wl.ext_dyn[A:A + P], wl.ext_dyn[A + P:]
    def run():
        wl.evaluator.evaluate_h(fixed=wl.ext_fixed, advice=adv, instance=[], l0=wl.ext_l[0], l_last=wl.ext_l[1], l_active_row=wl.ext_l[2],
                                perm_cosets=wl.ext_sigma, perm_products=zs, lookup_product=lk[0:L], lookup_input=lk[L:2 * L], lookup_table=lk[2 * L:3 * L],
                                challenges=[], beta=wl.scal[0], gamma=wl.scal[1], theta=wl.scal[2], y=wl.scal[3], out=wl.h_ext)
    print(be.quotient_program_info(wl.evaluator.handle))
    for thr in (64, 128, 256, 512):
        be.tune(quot_threads=thr)
        run()
        t = time.time()
        for _ in range(5): run()
        print(json.dumps({"quot_threads": thr, "ms": round((time.time() - t) / 5 * 1e3, 3)}), flush=True)

if __name__ == "__main__":
    main()

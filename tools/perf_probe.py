#!/usr/bin/env python3
"""GPU-box probe: times MSM / NTT at several sizes with per-kernel HIP-event timings and a few
tunable sweeps.  Exploration tool (bench.py is the contract benchmark)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

def rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1)
    return a

def main():
    be = z.Backend(0)
    print(be.version(), flush=True)
    sizes = [int(x) for x in os.environ.get("MSM_LOGS", "16,19,20,22").split(",")]
    for lg in sizes:
        n = 1 << lg
        ks, sc = rand_fr(n, 1), rand_fr(n, 2)
        dk, dp = be.to_device(ks), be.alloc(n * 64)
        t = time.time(); be.g1_fixed_base_mul(dk, n, dp); t_gen = time.time() - t
        t = time.time(); h = be.bases_register((dp, n)); t_reg = time.time() - t
        ds = be.to_device(sc)
        be.msm(h, ds, n)
        be.timing(True)
        reps = 5 if lg <= 20 else 3
        t = time.time()
        for _ in range(reps): be.msm(h, ds, n)
        dt = (time.time() - t) / reps
        lab = {k: be.timing_get(k) for k in ("msm_sort", "msm_accumulate", "msm_reduce")}
        be.timing(False)
        print(json.dumps({"msm_log_n": lg, "c": be.tune_get("msm_c"), "ms": round(dt * 1e3, 3), "Mscalar/s": round(n / dt / 1e6, 2),
                          "gen_s": round(t_gen, 3), "register_s": round(t_reg, 3),
                          "kernels_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in lab.items() if v[0] is not None}}), flush=True)
        if lg == 19:
            for key, vals in (("msm_block", [64, 256]), ("msm_merge_fanin", [4, 16]), ("msm_tree_fanin", [2, 8])):
                old = be.tune_get(key)
                for v in vals:
                    be.tune(**{key: v})
                    be.msm(h, ds, n)
                    t = time.time()
                    for _ in range(3): be.msm(h, ds, n)
                    print(json.dumps({"sweep": key, "value": v, "ms": round((time.time() - t) / 3 * 1e3, 3)}), flush=True)
                be.tune(**{key: old})
        if lg == 19:
            cols = [be.to_device(rand_fr(n, 10 + i)) for i in range(25)]
            for nb in (1, 4, 8, 25):
                be.msm_batch(h, cols[:nb], n)
                be.timing(True)
                t = time.time()
                for _ in range(3): be.msm_batch(h, cols[:nb], n)
                dt = (time.time() - t) / 3
                lab = {k: be.timing_get(k) for k in ("msm_sort", "msm_accumulate", "msm_reduce")}
                be.timing(False)
                print(json.dumps({"msm_batch": nb, "ms_total": round(dt * 1e3, 3), "ms_per_msm": round(dt * 1e3 / nb, 3),
                                  "kernels_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in lab.items() if v[0] is not None}}), flush=True)
            for mc in (16, 32, 64, 128, 256, 512):
                be.tune(msm_max_chunk=mc)
                be.msm_batch(h, cols[:16], n)
                be.timing(True)
                t = time.time()
                for _ in range(3): be.msm_batch(h, cols[:16], n)
                dt = (time.time() - t) / 3
                lab = {k: be.timing_get(k) for k in ("msm_sort", "msm_accumulate", "msm_reduce")}
                be.timing(False)
                print(json.dumps({"batch16_max_chunk": mc, "ms_per_msm": round(dt * 1e3 / 16, 3),
                                  "kernels_ms_per_msm": {k: round(v[0] / max(v[1], 1) / 16, 3) for k, v in lab.items() if v[0] is not None}}), flush=True)
            be.tune(msm_max_chunk=512)
            for c_ in cols: c_.free()
        be.bases_release(h); dk.free(); dp.free(); ds.free()
    for lg in [int(x) for x in os.environ.get("NTT_LOGS", "16,19,21,22").split(",")]:
        n = 1 << lg
        a = rand_fr(n, 3)
        d = be.to_device(a)
        w = np.array([(7 >> 0)], dtype=np.uint64)  # any Fr value works for timing; use a real root below
        from_root = pow(7, (R - 1) >> lg, R)
        wl = np.array([(from_root * (1 << 256) % R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        for tile, radix in ((11, 8), (11, 11), (12, 11), (12, 8), (10, 7), (12, 12)):
            be.tune(ntt_tile_log=tile, ntt_max_radix_log=radix)
            try:
                be.ntt_dev(d, lg, wl)
                t = time.time()
                for _ in range(5): be.ntt_dev(d, lg, wl)
                dt = (time.time() - t) / 5
                print(json.dumps({"ntt_log_n": lg, "tile": tile, "radix": radix, "ms": round(dt * 1e3, 3), "GB/s": round(64 * n / dt / 1e9, 1)}), flush=True)
            except Exception as e:
                print("ntt", lg, tile, radix, "failed:", e, flush=True)
        be.tune(ntt_tile_log=11, ntt_max_radix_log=8)
        d.free()

if __name__ == "__main__":
    main()

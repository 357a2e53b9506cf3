#!/usr/bin/env python3
"""GPU-box probe: NTT plan sweep (tile size, max radix, threads per workgroup)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_dcap_verifier_amd as z
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

def main():
    be = z.Backend(0)
    rng = np.random.default_rng(1)
    for lg in (19, 21, 22):
        n = 1 << lg
        a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        d = be.to_device(a)
        w = pow(7, (R - 1) >> lg, R)
        wl = np.array([((w << 256) % R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        best = None
        for tile in (9, 10, 11, 12):
            for radix in (6, 7, 8, 10, 11):
                if radix > tile: continue
                for thr in (128, 256, 512, 1024):
                    if (1 << tile) < thr * 2: continue
                    be.tune(ntt_tile_log=tile, ntt_max_radix_log=radix, ntt_threads=thr)
                    try:
                        be.ntt_dev(d, lg, wl)
                        t = time.time()
                        for _ in range(5): be.ntt_dev(d, lg, wl)
                        dt = (time.time() - t) / 5
                    except Exception as e:
                        print("fail", lg, tile, radix, thr, e); continue
                    rec = {"log_n": lg, "tile": tile, "radix": radix, "threads": thr, "ms": round(dt * 1e3, 4)}
                    if best is None or dt < best[0]: best = (dt, rec)
                    print(json.dumps(rec), flush=True)
        print("BEST", json.dumps(best[1]), flush=True)
        d.free()

if __name__ == "__main__":
    main()

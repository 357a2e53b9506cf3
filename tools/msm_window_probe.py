"""GPU box: one BN254 MSM of 2^log_n uniform scalars per window width c (tune msm_c; 0 = the library's own choice), closed form checked, with the
library's phase timers (sort / bucket chain / reduction).  usage: python tools/msm_window_probe.py <log_n> <c> [<c> ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import zk_dcap_verifier_amd as z

log_n = int(sys.argv[1])
be = z.Backend(0, os.environ.get("ZK_LIB") or None)
for kv in os.environ.get("ZK_TUNE", "").split(","):
    if kv:
        be.tune(**{kv.split("=")[0]: int(kv.split("=")[1])})
for c in [int(a) for a in sys.argv[2:]]:
    be.tune(msm_c=c)
    be.timing(True)
    out = bench.msm_microbench(be, log_n, 20241010, reps=3, verify=True)
    t = {k: be.timing_get(k) for k in ("msm_sort", "msm_accumulate", "msm_reduce")}
    be.timing(False)
    out.update({"c": c, "phase_ms_per_call": {k: round(v[0] / max(v[1], 1), 3) for k, v in t.items() if v[0] is not None}})
    print(json.dumps(out), flush=True)
be.tune(msm_c=0)

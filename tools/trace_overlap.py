"""Where the GPU's time goes with several proofs in flight: a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --no-extras` (four in flight).
usage: python tools/trace_overlap.py <s_kernel_trace.csv> [proofs=12] [accumulate_launches_per_proof=7]
The timed region is taken as the last `proofs` proofs (by msm_accumulate launches).  Time is cut at every kernel start / end; a slice of length dt with m kernels
running gives dt / m to each of them ("share": the kernels' shares add up to the busy time of the region) and dt to each of them ("wall": what --stats sums).
Also printed: how long m kernels ran together, the class mixes, and which kernels hold the GPU while none of the three arithmetic classes runs."""
import collections, csv, re, sys

path = sys.argv[1]
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
per_proof = int(sys.argv[3]) if len(sys.argv) > 3 else 7


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("zk::", "")
    n = re.sub(r"<.*", "", n)
    return "zkq_generated" if re.match(r"zkq[0-9a-f]{6}_", n) else n


rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if "msm_accumulate" in r["Kernel_Name"]]
t0 = int(rows[acc[-per_proof * proofs]]["Start_Timestamp"]) - 3_000_000
sel = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows if int(r["Start_Timestamp"]) >= t0]
ev = []
for s, e, n in sel:
    ev.append((s, 1, n))
    ev.append((e, 0, n))
ev.sort(key=lambda x: (x[0], x[1]))
run = collections.Counter()
share, wall, cnt = collections.Counter(), collections.Counter(), collections.Counter()
conc = collections.Counter()
beside = collections.defaultdict(collections.Counter)
only = collections.Counter()
CLASS = {"msm_accumulate_kernel": "acc", "ntt_strided_pass29_kernel": "ntt", "ntt_final_pass_kernel": "ntt", "zkq_generated": "quot", "quotient_kernel": "quot"}
last = ev[0][0]
for t, kind, n in ev:
    dt = t - last
    if dt > 0:
        m = sum(run.values())
        conc[m] += dt
        if m:
            classes = collections.Counter()
            for k, c in run.items():
                if c:
                    share[k] += dt * c / m
                    wall[k] += dt * c
                    classes[CLASS.get(k, "other")] += c
            key = "+".join(f"{c}{k}" for k, c in sorted(classes.items()))
            beside[key]["t"] += dt
            if set(classes) == {"other"}:
                for k, c in run.items():
                    if c:
                        only[k] += dt * c / m
    last = t
    if kind:
        run[n] += 1
        cnt[n] += 1
    else:
        run[n] -= 1
span = ev[-1][0] - ev[0][0]
print(f"# {path}: last {proofs} proofs; span {span / proofs / 1e6:.2f} ms/proof")
print("# kernels running together: " + ", ".join(f"{m}: {100 * v / span:.1f} %" for m, v in sorted(conc.items())))
print(f"{'kernel':34s} {'launches':>8s} {'share ms/proof':>15s} {'wall ms/proof':>14s} {'wall/share':>10s}")
for k, v in share.most_common():
    print(f"{k:34s} {cnt[k] / proofs:8.1f} {v / proofs / 1e6:15.3f} {wall[k] / proofs / 1e6:14.3f} {wall[k] / v:10.2f}")
print("# class mixes (acc = msm_accumulate, ntt = both NTT passes, quot = quotient kernels), share of the span:")
for k, v in sorted(beside.items(), key=lambda kv: -kv[1]["t"])[:25]:
    print(f"  {k:40s} {100 * v['t'] / span:5.1f} %")
print("# slices with NO msm_accumulate / NTT / quotient kernel running, by kernel (share, ms/proof):")
for k, v in only.most_common(12):
    print(f"  {k:34s} {v / proofs / 1e6:7.3f}")

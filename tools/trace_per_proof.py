"""Per-proof kernel table from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --inflight 1 --no-extras`.
usage: python tools/trace_per_proof.py <s_kernel_trace.csv> [proofs=3] [accumulate_launches_per_proof=7]
The timed region is taken as the last `proofs` proofs: it starts a little before the (28 * proofs)-th msm_accumulate launch from the end."""
import collections, csv, re, sys

path = sys.argv[1]
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if "msm_accumulate" in r["Kernel_Name"]]
per_proof = int(sys.argv[3]) if len(sys.argv) > 3 else 7           # msm_accumulate launches per proof (one per commitment phase)
t0 = int(rows[acc[-per_proof * proofs]]["Start_Timestamp"]) - 3_000_000
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
tot, cnt = collections.Counter(), collections.Counter()
for r in sel:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("zk::", "")
    tot[nm] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[nm] += 1
T = sum(tot.values())
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"# {path}: last {proofs} proofs; span {span / proofs / 1e6:.2f} ms/proof, kernels busy {T / proofs / 1e6:.2f} ms/proof ({100 * T / span:.0f} % of the span)")
print(f"{'kernel':45s} {'launches':>8s} {'ms/proof':>9s} {'share':>6s}")
for k, v in tot.most_common():
    print(f"{k:45s} {cnt[k] / proofs:8.1f} {v / proofs / 1e6:9.3f} {100 * v / T:5.1f}%")

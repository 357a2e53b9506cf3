#!/usr/bin/env python3
"""Where one proof's wall time goes when it runs alone: kernels and the idle gaps between them, from a `rocprofv3 --kernel-trace --output-format csv` trace.
The last proof = from the last-but-one occurrence of the anchor kernel (default msm_rc_class_kernel's final launch is not unique, so: the LAST `count` launches of the trace, where
`count` = launches per proof = total launches of the anchor / proofs).  usage: trace_gaps.py <kernel_trace.csv> <proofs in the trace> [anchor kernel = lpb_canon_kernel]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
proofs = int(sys.argv[2])
anchor = sys.argv[3] if len(sys.argv) > 3 else "lpb_canon_kernel"
short = lambda r: r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").replace("__amd_rocclr_", "").split("<")[0]
idx = [i for i, r in enumerate(rows) if short(r) == anchor]
per = len(idx) // proofs
a, b = idx[-2 * per], idx[-per]                                          # the last-but-one proof, anchor to anchor: one whole proof period
seg = rows[a:b]
span = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print(f"one proof period: {span:.2f} ms, kernels busy {busy:.2f} ms in {len(seg)} launches (the period includes the caller's work between two proofs)")
agg = collections.Counter(); cnt = collections.Counter()
for r in seg:
    agg[short(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[short(r)] += 1
for n, v in agg.most_common(16):
    print(f"  {v:8.3f} ms {cnt[n]:4d}  {n}")
gaps = collections.Counter(); gc = collections.Counter()
for x, y in zip(seg, seg[1:] + [rows[b]]):
    g = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e6
    if g > 0.01:
        gaps[(short(x), short(y))] += g; gc[(short(x), short(y))] += 1
print(f"idle gaps above 10 us: {sum(gaps.values()):.2f} ms")
for kk, v in gaps.most_common(14):
    print(f"  {v:8.3f} ms {gc[kk]:4d}  {kk[0]} -> {kk[1]}")

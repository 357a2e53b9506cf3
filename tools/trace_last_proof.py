#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --output-format csv` trace of tools/prover_probe.py: kernel time of the LAST proof, by kernel and per MSM batch.
usage: trace_last_proof.py <kernel_trace.csv> [msm batches per proof = 7]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    per_proof = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    short = lambda r: r["Kernel_Name"].split("(")[0].replace("zk::", "")
    first, prev_msm = [], False                      # an MSM batch starts at the first msm_* kernel after a non-msm one
    for i, r in enumerate(rows):
        is_msm = short(r).replace("void ", "").startswith("msm_")
        if is_msm and not prev_msm:
            first.append(i)
        prev_msm = is_msm
    start = first[-per_proof]
    agg = collections.OrderedDict()
    for r in rows[start:]:
        a = agg.setdefault(short(r), [0, 0.0])
        a[0] += 1
        a[1] += dur(r)
    print("kernel ms", round(sum(v[1] for v in agg.values()), 2), " span ms", round((int(rows[-1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])) / 1e6, 2))
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{d:9.3f} ms {c:5d}  {n}")
    for i in first[-per_proof:]:
        out = []
        for r in rows[i:i + 40]:
            if not short(r).replace("void ", "").startswith("msm"):
                break
            out.append((short(r).replace("void ", "")[4:12], round(dur(r), 3)))
        print("cols", rows[i].get("Grid_Size_Y"), out)


main()

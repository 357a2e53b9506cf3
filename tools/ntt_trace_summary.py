#!/usr/bin/env python3
"""NTT launches (> 0.2 ms) of the LAST proof in a `rocprofv3 --kernel-trace --output-format csv` trace of tools/prover_probe.py:
(kernel, grid x, ms).  usage: ntt_trace_summary.py <kernel_trace.csv>   (profiles/r01/run48_ntt_pass_memory_vs_arithmetic.txt)"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
short=lambda r:r["Kernel_Name"].split("(")[0].replace("zk::","")
first=[i for i,r in enumerate(rows) if short(r)=="msm_hist_kernel"]
start=first[-7]
out=[]
for r in rows[start:]:
    n=short(r)
    if n.startswith("ntt_"):
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
        if d>0.2: out.append((n[4:11],r["Grid_Size_X"],round(d,3)))
print(out)

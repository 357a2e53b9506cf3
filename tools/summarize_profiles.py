#!/usr/bin/env python3
"""Turn one profiling session's rocprofv3 CSVs (gpurun_out/<dir>) into the summaries committed under profiles/<round>/ (usage: <dir> <tag> [round = r02]).
Session layout (see tools/collect_profiles.sh): <dir>/runNN_bench_default.json, runNN_bench_under_rocprofv3.json,
stats/s_kernel_{stats,trace}.csv, fetch/f_counter_collection.csv, write/w_counter_collection.csv, lds/l_counter_collection.csv"""
import collections
import csv
import shutil
import json
import os
import re
import sys


def short(n):
    return re.sub(r"<.*", "", re.sub(r"\(.*", "", n).replace("zk::", "").replace("void ", ""))      # templates: quotient_kernel<1> -> quotient_kernel


def last_json_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def main():
    O, tag = sys.argv[1], sys.argv[2]
    rnd = sys.argv[3] if len(sys.argv) > 3 else "r02"
    P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", rnd)
    os.makedirs(P, exist_ok=True)
    b = last_json_line(f"{O}/{tag}_bench_under_rocprofv3.json")
    rows = list(csv.DictReader(open(f"{O}/stats/s_kernel_trace.csv")))
    acc = sorted((r for r in rows if "msm_accumulate" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    nl = b["roofline"]["launches"]
    timed = acc[-nl:]
    avg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed) / len(timed) / 1e6
    inflight = b["extra"]["proofs_in_flight"]
    with open(f"{P}/{tag}_rocprofv3_kernel_stats_bench_steps3_warmup1_noextras_inflight{inflight}.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras   (default mode: real create_proof)\n")
        f.write(f"# the table covers the WHOLE process (SRS setup with the EC-NTT, keygen, warm-up step, 3 timed steps); msm_accumulate_kernel: {len(acc)} launches in the process,\n")
        f.write(f"# the last {nl} of them are the timed region: average {avg:.4f} ms per launch from this trace vs {b['roofline']['avg_launch_ms']:.4f} ms from the in-bench HIP events of the same run ({tag}_bench_under_rocprofv3.json)\n")
        f.write(open(f"{O}/stats/s_kernel_stats.csv").read())
    print("msm_accumulate timed-region avg (trace) %.4f ms vs HIP events %.4f ms" % (avg, b["roofline"]["avg_launch_ms"]))
    # the same command with ONE proof in flight: no other stream's kernels inside an event pair, so the two clocks must agree closely; plus the per-proof kernel table
    if os.path.exists(f"{O}/stats1/s_kernel_trace.csv"):
        b1 = last_json_line(f"{O}/{tag}_bench_under_rocprofv3_inflight1.json")
        rows1 = sorted(csv.DictReader(open(f"{O}/stats1/s_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
        acc1 = [r for r in rows1 if "msm_accumulate" in r["Kernel_Name"]]
        nl1 = b1["roofline"]["launches"]
        avg1 = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in acc1[-nl1:]) / nl1 / 1e6
        proofs = b1["steps"]
        t0 = int(acc1[-nl1]["Start_Timestamp"]) - 3_000_000
        sel = [r for r in rows1 if int(r["Start_Timestamp"]) >= t0]
        tot, cnt = collections.Counter(), collections.Counter()
        for r in sel:
            tot[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            cnt[short(r["Kernel_Name"])] += 1
        T = sum(tot.values())
        span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
        with open(f"{P}/{tag}_rocprofv3_kernel_trace_per_proof_inflight1.txt", "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --inflight 1\n")
            f.write(f"# msm_accumulate_kernel over the {nl1} launches of the timed region: {avg1:.4f} ms per launch from the trace vs {b1['roofline']['avg_launch_ms']:.4f} ms from the in-bench HIP events\n")
            f.write(f"# timed region = {proofs} proofs, one at a time: {span / proofs / 1e6:.2f} ms per proof, kernels busy {T / proofs / 1e6:.2f} ms per proof; bench line of the run: {b1['ms_per_step']:.2f} ms per proof\n")
            f.write(f"{'kernel':45s} {'launches/proof':>14s} {'ms/proof':>9s} {'share':>6s}\n")
            for k_, v in tot.most_common():
                f.write(f"{k_:45s} {cnt[k_] / proofs:14.1f} {v / proofs / 1e6:9.3f} {100 * v / T:5.1f}%\n")
        shutil.copy(f"{O}/{tag}_bench_under_rocprofv3_inflight1.json", f"{P}/{tag}_bench_under_rocprofv3_inflight1.json")
        print("one proof in flight: trace %.4f ms vs HIP events %.4f ms; kernels busy %.2f ms per proof" % (avg1, b1["roofline"]["avg_launch_ms"], T / proofs / 1e6))

    def agg(path, counter):
        d = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                d[k][0] += 1
                d[k][1] += float(r["Counter_Value"])
        return d
    fe, wr = agg(f"{O}/fetch/f_counter_collection.csv", "FETCH_SIZE"), agg(f"{O}/write/w_counter_collection.csv", "WRITE_SIZE")
    keep = ["msm_accumulate_kernel", "quotient_kernel", "ntt_strided_pass_kernel", "ntt_final_pass_kernel", "msm_hist_kernel", "msm_scatter_kernel", "lpb_scatter_kernel",
            "lpb_hist_kernel", "msm_merge_kernel", "msm_rowcol_kernel", "pe_eval_partial_kernel", "pe_lincomb_kernel", "gp_batch_divide_kernel"]
    traffic = {}
    with open(f"{P}/{tag}_rocprofv3_pmc_hbm_traffic.csv", "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras  (one real proof + setup/keygen)\n")
        f.write("# bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024: the gfx950 correction of MI355X_MICROARCH.md (HBM section); launches include keygen launches of the same kernel\n")
        f.write("kernel,launches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,hbm_bytes_per_launch\n")
        for k in keep:
            if k in fe:
                n = fe[k][0]
                by = (2 * fe[k][1] + wr.get(k, [0, 0])[1]) * 1024 / n
                traffic[k] = by
                f.write(f"{k},{n},{fe[k][1]:.1f},{wr.get(k, [0, 0])[1]:.1f},{by:.0f}\n")
    tj = {"source": f"profiles/{rnd}/{tag}_rocprofv3_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0 --inflight 1 --no-extras; "
                    "bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 per the gfx950 correction of MI355X_MICROARCH.md HBM section)",
          "msm_accumulate_bytes_per_launch": round(traffic["msm_accumulate_kernel"]), "quotient_bytes_per_launch": round(traffic["quotient_kernel"]),
          "ntt_strided_pass_bytes_per_launch": round(traffic["ntt_strided_pass_kernel"]), "ntt_final_pass_bytes_per_launch": round(traffic["ntt_final_pass_kernel"])}
    json.dump(tj, open(os.path.join(os.path.dirname(P), "traffic.json"), "w"), indent=1)
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f"{O}/lds/l_counter_collection.csv")):
        k = short(r["Kernel_Name"])
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
    with open(f"{P}/{tag}_rocprofv3_pmc_lds_bucket_pass.csv", "w") as f:
        f.write("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras\n")
        f.write("# LDS has no hit rate; what the counters give for the LDS-staged bucket pass (msm_hist / msm_scatter: counting sort of (scalar, window) pairs on LDS atomics) and for the\n")
        f.write("# lookup radix sort is the share of LDS cycles lost to bank conflicts: conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (both in LDS-array cycles), and LDS busy share of wave cycles\n")
        f.write("kernel,launches,SQ_INSTS_LDS,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,conflict_frac,lds_active_over_wave_cycles\n")
        for k in ["msm_hist_kernel", "msm_scatter_kernel", "msm_rowcol_kernel", "lpb_hist_kernel", "lpb_scatter_kernel", "ntt_strided_pass_kernel", "ntt_final_pass_kernel", "quotient_kernel"]:
            if k in d:
                v = d[k]
                act = v["SQ_LDS_IDX_ACTIVE"] or 1
                f.write(f"{k},{cnt[k]},{v['SQ_INSTS_LDS']:.0f},{v['SQ_LDS_IDX_ACTIVE']:.0f},{v['SQ_LDS_BANK_CONFLICT']:.0f},{v['SQ_LDS_BANK_CONFLICT'] / act:.4f},"
                        f"{v['SQ_LDS_IDX_ACTIVE'] / (v['SQ_WAVE_CYCLES'] or 1):.4f}\n")
    for name in (f"{tag}_bench_default.json", f"{tag}_bench_under_rocprofv3.json"):
        open(f"{P}/{name}", "w").write(json.dumps(last_json_line(f"{O}/{name}")) + "\n")


if __name__ == "__main__":
    main()

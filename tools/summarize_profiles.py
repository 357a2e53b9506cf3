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
    n = re.sub(r"<.*", "", re.sub(r"\(.*", "", n).replace("zk::", "").replace("void ", ""))      # templates: quotient_kernel<1> -> quotient_kernel
    return "zkq_generated" if re.match(r"zkq[0-9a-f]*_\d+$", n) else n                            # the kernels generated per quotient program (csrc/quotient_jit.hip): one class


def last_json_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


# instruction issue rates measured on the box by tools/microbench, lane-ops per clock per CU: v_mad_u64_u32 with 12 independent accumulators per lane (profiles/r03/run93_microbench_mad_issue_rate.txt:
# 55.2; the four-chain loop of rounds 1-2 read 39.6 because it is latency-limited) and simple VALU instructions (114.6, run66)
RATE_MAD64, RATE_SIMPLE, N_CU, N_SIMD = 55.2, 114.6, 256, 1024


def valu_section(O, P, tag, rnd):
    """VALU side of the integer roofline from two PMC passes of ONE proof (tools/collect_profiles.sh): per kernel the share of wave cycles spent issuing VALU /
    stalled / parked, VALUBusy (= 4 * SQ_ACTIVE_INST_VALU / SIMDs / GRBM_GUI_ACTIVE-per-XCD: the gfx94x formula, ROCm 7.2 ships none for gfx950), the effective clock
    under load (GRBM_GUI_ACTIVE / 8 / dispatch duration, MI355X_MICROARCH.md DVFS section) and, from the instruction mix, an ISSUE model that owes nothing to this
    repo's own loops: cycles the VALU needs = 64 * (INT64 insts / 39.6 + other VALU insts / 114.6) / CUs, over the cycles the dispatch had."""
    vp, mp = f"{O}/valu/v_counter_collection.csv", f"{O}/mix/m_counter_collection.csv"
    if not (os.path.exists(vp) and os.path.exists(mp)):
        print("no VALU counter passes in", O)
        return

    def per_kernel(path):
        d = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = set()
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            d[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                d[k]["launches"] += 1
                d[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        return d
    v, m = per_kernel(vp), per_kernel(mp)
    keep = ["msm_accumulate_kernel", "ntt_strided_pass_kernel", "ntt_strided_pass29_kernel", "ntt_final_pass_kernel", "ntt_final_pass29_kernel", "quotient_kernel", "zkq_generated", "msm_scatter_kernel", "msm_hist_kernel", "msm_merge_kernel",
            "msm_rowcol_kernel", "lpb_scatter_kernel", "pe_lincomb_kernel", "pe_eval_partial_kernel", "gp_batch_divide_kernel"]
    out = {}
    with open(f"{P}/{tag}_rocprofv3_pmc_valu.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE  and, separately,\n")
        f.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR   -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras\n")
        f.write("# per-wave shares are of SQ_WAVE_CYCLES (quad-cycles); valu_busy = 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); eff_clock = GRBM_GUI_ACTIVE / 8 / duration;\n")
        f.write(f"# issue_model = 64 * (INT64 insts / {RATE_MAD64} + (VALU - INT64) insts / {RATE_SIMPLE}) / {N_CU} CUs / (GRBM_GUI_ACTIVE / 8): the share of the dispatch's cycles the VALU needs at the MEASURED issue rates\n")
        f.write("# of v_mad_u64_u32 and simple VALU instructions (tools/microbench) — an instruction-mix bound, independent of any loop of this repo\n")
        f.write("kernel,launches,ms_per_launch,active_valu_per_wave_cycle,wait_inst_per_wave_cycle,wait_any_per_wave_cycle,valu_busy,eff_clock_ghz,valu_insts_per_launch,int64_share,int32_share,salu_per_valu,issue_model\n")
        for k in keep:
            if k not in v or k not in m or not v[k]["SQ_WAVE_CYCLES"]:
                continue
            a, b = v[k], m[k]
            gui = a["GRBM_GUI_ACTIVE"] / 8
            wc = a["SQ_WAVE_CYCLES"]
            i64, i32, iv = b["SQ_INSTS_VALU_INT64"], b["SQ_INSTS_VALU_INT32"], b["SQ_INSTS_VALU"] or 1
            scale = a["SQ_INSTS_VALU"] / iv if iv else 1.0            # the two passes may see a different number of launches (keygen): bring the mix to the first pass's count
            need = 64 * (i64 * scale / RATE_MAD64 + (iv - i64) * scale / RATE_SIMPLE) / N_CU
            rec = {"launches": int(a["launches"]), "ms_per_launch": a["ns"] / a["launches"] / 1e6, "active_valu_per_wave_cycle": a["SQ_ACTIVE_INST_VALU"] / wc,
                   "wait_inst_per_wave_cycle": a["SQ_WAIT_INST_ANY"] / wc, "wait_any_per_wave_cycle": a["SQ_WAIT_ANY"] / wc,
                   "valu_busy": 4 * a["SQ_ACTIVE_INST_VALU"] / N_SIMD / gui if gui else None, "eff_clock_ghz": gui / a["ns"] if a["ns"] else None,
                   "valu_insts_per_launch": a["SQ_INSTS_VALU"] / a["launches"], "int64_share": i64 / iv, "int32_share": i32 / iv, "salu_per_valu": b["SQ_INSTS_SALU"] / iv,
                   "issue_model": need / gui if gui else None}
            out[k] = {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in rec.items()}
            f.write(k + "," + ",".join("" if rec[c] is None else (f"{rec[c]:.4f}" if isinstance(rec[c], float) else str(rec[c])) for c in
                                       ("launches", "ms_per_launch", "active_valu_per_wave_cycle", "wait_inst_per_wave_cycle", "wait_any_per_wave_cycle", "valu_busy", "eff_clock_ghz",
                                        "valu_insts_per_launch", "int64_share", "int32_share", "salu_per_valu", "issue_model")) + "\n")
    tjp = os.path.join(os.path.dirname(P), "traffic.json")
    tj = json.load(open(tjp)) if os.path.exists(tjp) else {}
    tj["valu_source"] = f"profiles/{rnd}/{tag}_rocprofv3_pmc_valu.csv"
    tj["valu"] = out
    json.dump(tj, open(tjp, "w"), indent=1)
    for k, r in out.items():
        print(k, r)


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
    keep = ["msm_accumulate_kernel", "quotient_kernel", "zkq_generated", "ntt_strided_pass_kernel", "ntt_strided_pass29_kernel", "ntt_final_pass_kernel", "ntt_final_pass29_kernel", "msm_hist_kernel", "msm_scatter_kernel", "lpb_scatter_kernel",
            "lpb_hist_kernel", "msm_merge_kernel", "msm_rowcol_kernel", "pe_eval_partial_kernel", "pe_lincomb_kernel", "gp_batch_divide_kernel"]
    traffic, raw = {}, {}
    with open(f"{P}/{tag}_rocprofv3_pmc_hbm_traffic.csv", "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras  (one real proof + setup/keygen)\n")
        f.write("# two readings per kernel: raw = (FETCH_SIZE + WRITE_SIZE) KB * 1024 as counted; doubled = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024, the gfx950 correction of MI355X_MICROARCH.md (HBM section),\n")
        f.write("# which is calibrated for wide coalesced streaming reads (the NTT passes, the quotient's column reads) and NOT for 64-byte gathers (msm_accumulate: tools/microbench gather64 under the same counter);\n")
        f.write("# launches include keygen launches of the same kernel\n")
        f.write("kernel,launches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,raw_bytes_per_launch,doubled_fetch_bytes_per_launch\n")
        for k in keep:
            if k in fe:
                n = fe[k][0]
                by = (2 * fe[k][1] + wr.get(k, [0, 0])[1]) * 1024 / n
                traffic[k] = by
                raw[k] = {"launches": n, "fetch_bytes_per_launch": round(fe[k][1] * 1024 / n), "write_bytes_per_launch": round(wr.get(k, [0, 0])[1] * 1024 / n)}
                f.write(f"{k},{n},{fe[k][1]:.1f},{wr.get(k, [0, 0])[1]:.1f},{(fe[k][1] + wr.get(k, [0, 0])[1]) * 1024 / n:.0f},{by:.0f}\n")
    tj = {"source": f"profiles/{rnd}/{tag}_rocprofv3_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0 --inflight 1 --no-extras; "
                    "bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 per the gfx950 correction of MI355X_MICROARCH.md HBM section)",
          "msm_accumulate_bytes_per_launch": round(traffic["msm_accumulate_kernel"]), "quotient_bytes_per_launch": round(traffic["quotient_kernel"]),
          "quotient_generated_bytes_per_launch": round(traffic["zkq_generated"]) if "zkq_generated" in traffic else None,      # the kernels generated for the key's program (tune quot_jit): one launch = one kernel of a part
          "ntt_strided_pass_bytes_per_launch": round(traffic.get("ntt_strided_pass29_kernel") or traffic["ntt_strided_pass_kernel"]),      # (the 29-bit-limb kernel when the plan uses it)
          "ntt_final_pass_bytes_per_launch": round(traffic.get("ntt_final_pass_kernel") or traffic["ntt_final_pass29_kernel"]),
          "kernels": raw}                                              # FETCH_SIZE / WRITE_SIZE as counted, per launch: bench.py reports raw and doubled readings side by side
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
        for k in ["msm_hist_kernel", "msm_scatter_kernel", "msm_rowcol_kernel", "lpb_hist_kernel", "lpb_scatter_kernel", "ntt_strided_pass_kernel", "ntt_strided_pass29_kernel", "ntt_final_pass_kernel",
                  "ntt_final_pass29_kernel", "quotient_kernel", "zkq_generated"]:
            if k in d:
                v = d[k]
                act = v["SQ_LDS_IDX_ACTIVE"] or 1
                f.write(f"{k},{cnt[k]},{v['SQ_INSTS_LDS']:.0f},{v['SQ_LDS_IDX_ACTIVE']:.0f},{v['SQ_LDS_BANK_CONFLICT']:.0f},{v['SQ_LDS_BANK_CONFLICT'] / act:.4f},"
                        f"{v['SQ_LDS_IDX_ACTIVE'] / (v['SQ_WAVE_CYCLES'] or 1):.4f}\n")
    valu_section(O, P, tag, rnd)
    calibration_section(O, P, tag)
    for name in (f"{tag}_bench_default.json", f"{tag}_bench_under_rocprofv3.json"):
        open(f"{P}/{name}", "w").write(json.dumps(last_json_line(f"{O}/{name}")) + "\n")


def calibration_section(O, P, tag):
    """FETCH_SIZE of tools/microbench --gather64 / --stream32 (known byte counts) next to what the kernels requested: settles which reading of the counter applies to
    64-byte gathers (msm_accumulate) and to 32-byte-per-lane streams (NTT passes, quotient): profiles/<round>/<tag>_fetch_size_calibration.txt"""
    src = f"{O}/{tag}_fetch_size_calibration.txt"
    if not os.path.exists(src):
        return
    lines = [l.rstrip("\n") for l in open(src) if l.startswith("k_")]
    counted = {}
    for d, kern in (("cal_gather", "k_gather64"), ("cal_stream", "k_stream32")):
        for root, _, files in os.walk(f"{O}/{d}"):
            for fn in files:
                if fn.endswith("counter_collection.csv"):
                    for r in csv.DictReader(open(os.path.join(root, fn))):
                        if r["Counter_Name"] == "FETCH_SIZE" and kern in r["Kernel_Name"]:
                            counted[kern] = counted.get(kern, 0.0) + float(r["Counter_Value"]) * 1024
    with open(f"{P}/{tag}_fetch_size_calibration.txt", "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE --output-format csv -- ./tools/microbench --gather64 | --stream32   (one kernel each, known byte count)\n")
        for l in lines:
            f.write(l + "\n")
            kern = l.split(":")[0]
            req = float(l.split(" bytes")[0].split()[-1]) if kern == "k_stream32" else float(l.split("= ")[1].split(" bytes")[0])
            if kern in counted:
                f.write(f"    FETCH_SIZE as counted: {counted[kern]:.0f} bytes = {counted[kern] / req:.3f} of the bytes requested  (x2: {2 * counted[kern] / req:.3f})\n")
    print(open(f"{P}/{tag}_fetch_size_calibration.txt").read())


if __name__ == "__main__":
    main()

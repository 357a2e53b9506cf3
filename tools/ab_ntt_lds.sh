set -u
export TMPDIR=/tmp
O=gpurun_out/r05_ntt
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_coset.py tests/test_field29.py tests/test_baseline_sizes.py -m gpu -x -q > $O/parity.log 2>&1; tail -2 $O/parity.log
python tools/ntt_ab.py 19 64 base:tools/libzkmi355_base.so: new:-: base2:tools/libzkmi355_base.so: new2:-: > $O/ntt_ab.txt 2>&1; cat $O/ntt_ab.txt
for lib in base new; do
  if [ $lib = base ]; then export ZK_LIB=$PWD/tools/libzkmi355_base.so; else unset ZK_LIB; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/lds_$lib -o l -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras > /dev/null 2>> $O/bench.err
  python - <<PY
import csv,glob,collections
f=glob.glob("$O/lds_$lib/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0].split("::")[-1]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "ntt" in k or "msm_hist" in k or "lpb_hist" in k or "msm_scatter" in k:
        print("$lib", k, "conflict_frac %.4f" % (v["SQ_LDS_BANK_CONFLICT"]/max(1,v["SQ_LDS_IDX_ACTIVE"])), "lds_active/wave_cycles %.4f" % (v["SQ_LDS_IDX_ACTIVE"]/max(1,v["SQ_WAVE_CYCLES"])))
PY
done

#!/bin/bash
# Run ON THE GPU BOX (gpurun): one profiling session of the default bench.  usage: tools/collect_profiles.sh <tag>   e.g. run32
# Counter passes are separate rocprofv3 runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC never together with trace domains other than --kernel-trace).
set -u
TAG=$1
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/${TAG}_bench_default.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-extras > $O/${TAG}_bench_under_rocprofv3.json 2>> $O/bench.err
# (one proof at a time with the side lane OFF: the per-kernel tables and counters below are of kernels that run alone, not beside the helper context's transforms)
export ZK_TUNE=prover_side_lane=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o s -- python3 bench.py --steps 3 --warmup 1 --no-extras --inflight 1 > $O/${TAG}_bench_under_rocprofv3_inflight1.json 2>> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras > /dev/null 2>> $O/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras > /dev/null 2>> $O/bench.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/lds -o l -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras > /dev/null 2>> $O/bench.err
# FETCH_SIZE on this repo's own access patterns with KNOWN byte counts (MI355X_MICROARCH.md: other widths than wide streaming reads are uncalibrated): a 64-byte gather per lane
# (msm_accumulate's table reads) and a 32-byte-per-lane stream (NTT passes, quotient columns)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_gather -o c -- ./tools/microbench --gather64 > $O/${TAG}_fetch_size_calibration.txt 2>> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_stream -o c -- ./tools/microbench --stream32 >> $O/${TAG}_fetch_size_calibration.txt 2>> $O/bench.err
# VALU side of the integer roofline: issue / stall split + the chip's effective clock (GRBM_GUI_ACTIVE / 8 / duration), then the instruction mix
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/valu -o v -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras > /dev/null 2>> $O/bench.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/mix -o m -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-extras > /dev/null 2>> $O/bench.err
unset ZK_TUNE
ZK_PROVER=native python tools/prover_probe.py 19 3 > $O/${TAG}_prover_probe_k19.json 2>> $O/bench.err
ZK_CENSUS=reference_exact python tools/prover_probe.py 19 3 > $O/${TAG}_prover_probe_k19_census_reference_exact.json 2>> $O/bench.err
python tools/prover_probe.py 21 1 > $O/${TAG}_prover_probe_k21.json 2>> $O/bench.err
grep -h '^{' $O/${TAG}_bench_default.json | cut -c1-330

#!/bin/bash
# GPU box: same-box sweep of library tunables on the default bench (four in flight unless INFLIGHT is set), every candidate bracketed by the defaults.
# usage: tools/tune_sweep.sh <out-file> "<ZK_TUNE string>" ...      ("-" = the defaults)
set -u
OUT=$1; shift
F=${INFLIGHT:-4}
one() {
  if [ "$1" = "-" ]; then unset ZK_TUNE; else export ZK_TUNE=$1; fi
  python bench.py --steps ${STEPS:-10} --warmup 3 --no-extras --inflight $F 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']
print('%-44s inflight $F  %9.1f proofs/h  %7.3f ms/proof  acc %.4f ms/launch' % ('$1', d['value'], e['ms_per_proof'], d['roofline']['avg_launch_ms']))" | tee -a $OUT
}
one -
for t in "$@"; do one "$t"; one -; done

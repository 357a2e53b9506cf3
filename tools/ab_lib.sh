# GPU box: same-box A/B of two builds of the library (ZK_LIB = tools/libzkmi355_prev.so against the tree's) on the default bench, one and four proofs in flight, alternating
set -u
for i in 1 2 3; do for lib in prev new; do
  if [ $lib = prev ]; then export ZK_LIB=$PWD/tools/libzkmi355_prev.so; else unset ZK_LIB; fi
  for f in 1 4; do python bench.py --steps 10 --warmup 3 --no-extras --inflight $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; a=e.get('event_span_ms_per_proof_pipelined') or {}
print('$lib inflight $f', d['value'], e['ms_per_proof'], 'acc', d['roofline']['avg_launch_ms'])"; done
done; done

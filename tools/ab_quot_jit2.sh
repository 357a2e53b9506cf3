set -u
O=gpurun_out/r05_jit
mkdir -p $O
python tools/quot_jit_probe.py 19 3 jit29_24:quot_jit=2,quot_jit_group=24 jit29_32:quot_jit=2,quot_jit_group=32 jit29_48:quot_jit=2,quot_jit_group=48 jit29_64:quot_jit=2,quot_jit_group=64 2>&1 | tee $O/probe29b.txt
for spec in "interp:" "jit29_40:quot_jit=2,quot_jit_group=40" "jit32_200:quot_jit=1,quot_jit_group=200" "interp2:" "jit29_40b:quot_jit=2,quot_jit_group=40" "jit32_200b:quot_jit=1,quot_jit_group=200" "interp3:" "jit29_40c:quot_jit=2,quot_jit_group=40"; do
  name=${spec%%:*}; tune=${spec#*:}
  ZK_TUNE=$tune python bench.py --steps 10 --warmup 3 --no-extras --inflight 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']
print('$name', 'proofs/h', d['value'], 'ms/proof', e.get('ms_per_proof'))
"
done 2>&1 | tee $O/ab2.txt

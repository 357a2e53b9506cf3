# GPU box: the generated quotient kernels (tune quot_jit) against the interpreter — parity suites under the knob, then same-box timing at one and four proofs in flight
set -u
O=gpurun_out/r05_jit
mkdir -p $O
ZK_TUNE=quot_jit=1 python -m pytest tests/test_quotient.py tests/test_native_prover.py tests/test_create_proof.py tests/test_piece_cosets.py -m gpu -x -q > $O/parity_jit.log 2>&1; tail -3 $O/parity_jit.log
for spec in "interp:" "jit24:quot_jit=1" "jit24w4:quot_jit=1,quot_jit_waves=4" "jit16:quot_jit=1,quot_jit_group=16" "jit40:quot_jit=1,quot_jit_group=40" "interp2:" "jit24b:quot_jit=1"; do
  name=${spec%%:*}; tune=${spec#*:}
  for f in 1 4; do
    ZK_TUNE=$tune python bench.py --steps 6 --warmup 2 --no-extras --inflight $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']
km=(e.get('event_span_ms_per_proof_pipelined') or {})
print('$name inflight $f', 'proofs/h', d['value'], 'ms/proof', e.get('ms_per_proof'), 'quotient ms/proof', km.get('quotient'))
"
  done
done 2>&1 | tee $O/ab.txt

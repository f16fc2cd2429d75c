#!/bin/bash
# Round 3, GPU call 3: magnitude-class accumulators in the distance pass (structured stacks, tau A/B), MFMA rounding probe,
# whole suite, Bulyan pass-2 burst A/B, the multi-GPU extras under a one-rank torchrun, default bench with live traffic.
out=gpurun_out/r3c3
mkdir -p $out
export TMPDIR=/tmp
echo "== mfma rounding probe"; timeout 60 scripts/probes/mfma_round_probe | tee $out/mfma_round_probe.txt | tail -8
for tau in 2e-3 2e-2; do
  echo "== structured stacks, BM_PAIR_TAU=$tau"
  ( BM_PAIR_TAU=$tau timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "structured" ) > $out/pytest_struct_$tau.log 2>&1
  grep -E "^FAILED|passed|failed|Error: " $out/pytest_struct_$tau.log | cut -c1-200 | tail -12
done
( time timeout 1800 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $out/pytest.log | cut -c1-300 | tail -15
echo "== pair probe"; ( timeout 600 python scripts/pair_probe.py time ) 2>&1 | grep "^time"
for b in 0 8; do echo "== BM_BUL_BURST=$b"; BM_BUL_BURST=$b timeout 300 python scripts/bulyan_pass2_probe.py 2>&1 | grep "pass 2" | cut -c1-200; done | tee $out/bulyan_pass2_ab.txt
echo "== one-rank torchrun, C4 with the multi-GPU extras"
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload bulyan --steps 10 --warmup 3 --no-cpu-baseline --no-traffic ) > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err
python3 - <<PY
import json
try:
  l=json.loads([x for x in open('$out/bench_torchrun1.json').read().strip().splitlines() if x.startswith('{')][-1])
  print('value', l['value'], l['config']['collectives'])
  for k,v in l['per_gar'].items(): print('  ', k, round(v['avg_ms'],4), v.get('bytes_sent_per_rank'))
except Exception as e:
  print('torchrun bench failed', e); print(open('$out/bench_torchrun1.err').read()[-2000:])
PY
echo "== default bench (live traffic, cpu baselines)"
( time python bench.py ) > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err
python3 - <<PY
import json
l=json.loads([x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], l['roofline']['traffic'])
for k,v in (l['roofline'].get('traffic_per_kernel') or {}).items(): print('  traffic', k, v['traffic'], round(v['ratio'],4))
for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps','distance_pass_ms','scalar_form_ms')}, (v.get('cpu_baseline') or {}).get('value'))
print('cpu', l.get('cpu_baseline',{}).get('value'), l.get('cpu_baseline',{}).get('cores'))
PY

#!/bin/bash
# brute without enumeration (tests + bench entries), HIP graphs of whole rules (test, per-rank probe, bench leg)
out=gpurun_out/r3c25
mkdir -p $out
export TMPDIR=/tmp
( time timeout 170 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -m gpu -x -q -k "brute or graphed or influence" ) > $out/pytest_new.log 2>&1; tail -6 $out/pytest_new.log
( time timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err
python3 - <<PY
import json
try:
  l=json.loads([x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1])
  print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], l['roofline']['traffic'])
  for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps')})
except Exception as e:
  print('bench parse failed', e)
PY
( timeout 80 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 scripts/per_rank_probe.py --only-c4 ) > $out/per_rank_p8_graph.txt 2>&1; grep -v "^\[\|Warning\|warn" $out/per_rank_p8_graph.txt | tail -6
( timeout 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload bulyan --graph-replay --steps 20 --warmup 5 --no-cpu-baseline --no-traffic ) > $out/bench_torchrun_graph.json 2> $out/bench_torchrun_graph.err; tail -c 1500 $out/bench_torchrun_graph.json

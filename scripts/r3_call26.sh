#!/bin/bash
# bench.py: the N > 1 line (colwise shards, weak; sharded Bulyan with its all-reduce under per_gar) exercised on one rank
out=gpurun_out/r3c26
mkdir -p $out
export TMPDIR=/tmp
( time timeout 110 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --sharded-extras --no-cpu-baseline ) > $out/bench_torchrun_1rank_sharded_extras.json 2> $out/t1.err; tail -3 $out/t1.err
( time timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err
( time timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --workload bulyan --steps 10 --warmup 3 ) > $out/bench_torchrun_1rank_bulyan.json 2> $out/t3.err; tail -3 $out/t3.err
python3 - <<PY
import json
for name in ('bench_torchrun_1rank_sharded_extras','bench_default','bench_torchrun_1rank_bulyan'):
  try:
    l=json.loads([x for x in open('$out/'+name+'.json').read().strip().splitlines() if x.startswith('{')][-1])
    print(name, 'value', round(l['value'],1), 'scaling', l['scaling'], 'ms', round(l['ms_per_step'],4), 'frac', round(l['roofline']['frac'],4), l['config']['workload'][:90])
    for k,v in l['per_gar'].items():
      print('    ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps','agg_per_s','error')})
    print('    extra:', {k: l[k] for k in ('single_gpu_same_workload','cpu_baseline') if k in l})
  except Exception as e:
    print(name, 'parse failed', e)
PY

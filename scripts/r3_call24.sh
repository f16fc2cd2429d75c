#!/bin/bash
out=gpurun_out/r3c24
mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -4 $out/bench_default.err
python3 - <<PY
import json
l=json.loads([x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], l['roofline']['traffic'])
for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps')}, (v.get('cpu_baseline') or {}).get('value'))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|rror" | tail -2

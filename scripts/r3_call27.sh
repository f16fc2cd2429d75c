#!/bin/bash
# Brute round trip through page-locked memory; the brute checks of pair_mode_check at n = 25 / 51; final default line
out=gpurun_out/r3c27
mkdir -p $out
export TMPDIR=/tmp
( time timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -m gpu -x -q -k "brute or graphed or influence or pair_mode or plugin or placements_against" ) > $out/pytest_subset.log 2>&1; grep -E "passed|failed|error" $out/pytest_subset.log | tail -3
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

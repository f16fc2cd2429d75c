#!/bin/bash
# the factor search against the median from two order statistics: GPU tests + the default line with its new entry
out=gpurun_out/r3c28
mkdir -p $out
export TMPDIR=/tmp
( time timeout 60 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -m gpu -x -q -k "search or two_order" ) > $out/pytest_search.log 2>&1; grep -E "passed|failed|error" $out/pytest_search.log | tail -3
( time timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err
python3 - <<PY
import json
try:
  l=json.loads([x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1])
  print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], l['roofline']['traffic'])
  for k in ('attack_search_c3_krum','attack_search_c2_median'):
    print(k, l['per_gar'].get(k))
except Exception as e:
  print('bench parse failed', e)
PY

#!/bin/bash
out=gpurun_out/r2c7
mkdir -p $out
export TMPDIR=/tmp
run() { # tag, env..., -- bench args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --no-traffic --no-extras > $out/$tag.json 2> $out/$tag.err
  python - <<PY
import json
l=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1])
print('$tag', ' '.join(f"{k}={v['avg_ms']*1e3:.1f}us/{v['gbps']:.0f}GB/s" for k,v in l['per_gar'].items()))
PY
}
run col_default X=1 -- --steps 40
run col_ablate BM_COL_ABLATE=1 -- --steps 40
run col_vec2 BM_FORCE_VEC=2 -- --steps 40
for b in 512 1024 2048 4096 8192; do run col_blocks$b BM_COL_MAX_BLOCKS=$b -- --steps 40; done
run step_default X=1 -- --workload step --steps 12
run step_plainstore BM_STEP_STORE=1 -- --workload step --steps 12
for b in 256 512 1024; do run step_blocks$b BM_STEP_BLOCKS=$b -- --workload step --steps 12; done
run step_median X=1 -- --workload step --gar median --steps 12
timeout 1200 python -m pytest tests/test_gpu_parity_r2.py -q -k "nan_attack or zero_length or momentum_stats or plugin" > $out/pytest.log 2>&1; tail -5 $out/pytest.log

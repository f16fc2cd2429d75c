#!/bin/bash
out=gpurun_out/r2c6
mkdir -p $out
export TMPDIR=/tmp
for cfg in "BM_STEP_STREAM=0" "BM_STEP_STREAM=1" "BM_STEP_STREAM=0 BM_STEP_VEC=2" "BM_STEP_STREAM=1 BM_STEP_VEC=2"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$tag -o s -- python bench.py --workload step --steps 12 --no-cpu-baseline --no-traffic > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "== $cfg"; python -c "
import json
l=json.loads(open('$out/bench_$tag.json').read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'], 'step ms', l['per_gar']['step']['avg_ms'])"
  find $out/st_$tag -name "*kernel_stats.csv" | head -1 | xargs grep "momentum_stats" | cut -d, -f1-4 | cut -c1-60,200-
done
timeout 2400 python -m pytest tests/test_gpu_parity_r2.py -q -s --durations=8 > $out/pytest.log 2>&1
tail -22 $out/pytest.log

#!/bin/bash
out=gpurun_out/r2c8
mkdir -p $out
export TMPDIR=/tmp
prof() { # tag env... -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$tag -o s -- python bench.py "$@" --no-cpu-baseline --no-traffic --no-extras > $out/$tag.json 2> $out/$tag.err
  echo "== $tag"; grep 'bm::' $out/st_$tag/s_kernel_stats.csv | sed -E 's/"(void )?bm::([a-z_0-9]+<?[0-9, a-z]*>?)[^"]*"/\2/' | cut -d, -f1-4 | head -7
}
prof col_plain X=1 -- --steps 40
prof col_nt BM_RESULT_NT=1 -- --steps 40
prof krum X=1 -- --workload krum --steps 20
prof bulyan X=1 -- --workload bulyan --steps 20
prof step_nt X=1 -- --workload step --steps 12
prof step_plain BM_STEP_STORE=1 -- --workload step --steps 12
python bench.py --steps 50 > $out/bench_default.json 2> $out/bench_default.err
python - <<PY
import json
l=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print('value',l['value'],'frac',l['roofline']['frac'],'traffic',l['roofline']['traffic'])
for k,v in l['per_gar'].items(): print(k, round(v['avg_ms'],4),'ms', round(v['gbps']),'GB/s')
PY
timeout 1500 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -q -x -k "zero_length or golden or full_size_colwise or every_n or unaligned or nan" > $out/pytest.log 2>&1; tail -4 $out/pytest.log

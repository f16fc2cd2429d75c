#!/bin/bash
# A/B of the result-store policy inside ONE call (same GPU): BM_RESULT_NT=0 (cacheable) vs 1 (non-temporal)
out=gpurun_out/r2c9
mkdir -p $out
export TMPDIR=/tmp
prof() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$tag -o s -- python bench.py "$@" --no-cpu-baseline --no-traffic --no-extras > $out/$tag.json 2> $out/$tag.err
  python3 - <<PY
import csv, json
l=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1])
print('== $tag', 'ms_per_step %.4f' % l['ms_per_step'])
for r in csv.DictReader(open('$out/st_$tag/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 30000:
        print('   %-42s avg %9.1f us' % (r['Name'].split('(')[0].replace('void bm::','').replace('bm::',''), float(r['AverageNs'])/1e3))
PY
}
for rep in 1 2; do
  prof step_cache_$rep BM_RESULT_NT=0 -- --workload step --steps 12
  prof step_nt_$rep BM_RESULT_NT=1 -- --workload step --steps 12
  prof krum_cache_$rep BM_RESULT_NT=0 -- --workload krum --steps 20
  prof krum_nt_$rep BM_RESULT_NT=1 -- --workload krum --steps 20
  prof bulyan_cache_$rep BM_RESULT_NT=0 -- --workload bulyan --steps 20
  prof bulyan_nt_$rep BM_RESULT_NT=1 -- --workload bulyan --steps 20
done
timeout 600 python -m pytest tests/test_gpu_parity_r2.py -q -x -k "zero_length or nan_attack" > $out/pytest.log 2>&1; tail -3 $out/pytest.log

#!/bin/bash
# deferred stores (vmcnt) in colwise / selected_mean / bulyan pass 2 / streaming momentum kernel
out=gpurun_out/r2c10
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
        print('   %-42s avg %9.1f us  min %9.1f' % (r['Name'].split('(')[0].replace('void bm::','').replace('bm::',''), float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
}
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -q -x -k "golden or every_n or unaligned or nan or momentum_stats or step_all or aggregation_step or long_columns or generic_bulyan or full_size_colwise or full_size_c4 or empty" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
prof colwise X=1 -- --steps 40
prof krum X=1 -- --workload krum --steps 20
prof bulyan X=1 -- --workload bulyan --steps 20
prof step_resident BM_STEP_STREAM=0 -- --workload step --steps 12
prof step_stream BM_STEP_STREAM=1 -- --workload step --steps 12
prof step_resident2 BM_STEP_STREAM=0 -- --workload step --steps 12
prof step_stream2 BM_STEP_STREAM=1 -- --workload step --steps 12

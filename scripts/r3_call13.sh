#!/bin/bash
# Round 3, GPU call 13: the coordinate-wise rule riding along with the first pass of the step.
out=gpurun_out/r3c13
mkdir -p $out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "riding or burst_form or c5_steady or steady_state" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-250 | tail -12
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $out/pytest.log | cut -c1-300 | tail -8
for gar in median trmean krum; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$gar -o s -- python bench.py --workload step --gar $gar --steps 15 --no-cpu-baseline --no-traffic > $out/bench_$gar.json 2> $out/bench_$gar.err
  python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench_$gar.json').read().strip().splitlines() if x.startswith('{')][-1])
print('== step $gar: ms_per_step %.4f  frac %.3f' % (l['ms_per_step'], l['roofline']['frac']))
for r in csv.DictReader(open('$out/st_$gar/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 20000:
        print('   %-78s calls %3s avg %9.1f us' % (r['Name'][:78].replace('void bm::','').replace('bm::',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
done

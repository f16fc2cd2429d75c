#!/bin/bash
# Round 3, GPU call 4: dithered running sums + magnitude-class accumulators (structured stacks), folded launches of the
# distance pass, one-wave-per-row rank kernel; whole suite; pair probe; default bench.
out=gpurun_out/r3c4
mkdir -p $out
export TMPDIR=/tmp
for tau in 2e-3; do
  echo "== structured stacks, BM_PAIR_TAU=$tau"
  ( BM_PAIR_TAU=$tau timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "structured" ) > $out/pytest_struct_$tau.log 2>&1
  grep -E "^FAILED|passed|failed|Error: " $out/pytest_struct_$tau.log | cut -c1-200 | tail -12
done
( time timeout 1800 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $out/pytest.log | cut -c1-300 | tail -15
echo "== pair probe"; ( timeout 600 python scripts/pair_probe.py time ) 2>&1 | grep "^time"
echo "== default bench under the kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- python bench.py --no-cpu-baseline --no-traffic > $out/bench.json 2> $out/bench.err
python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench.json').read().strip().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'])
for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps','distance_pass_ms','scalar_form_ms')})
for r in csv.DictReader(open('$out/trace/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 3000:
        print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY

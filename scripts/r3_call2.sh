#!/bin/bash
# Round 3, GPU call 2: fp64 running sums of the distance pass (tau A/B), switch-form trimmed sum (burst A/B),
# momentum pass (burst A/B), the one-pass study kernel; whole suite; kernel trace.
out=gpurun_out/r3c2
mkdir -p $out
export TMPDIR=/tmp
for tau in 2e-3 2e-2; do
  echo "== structured stacks, BM_PAIR_TAU=$tau"
  ( BM_PAIR_TAU=$tau timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "structured" ) > $out/pytest_struct_$tau.log 2>&1
  grep -E "^FAILED|passed|failed|worst relative" $out/pytest_struct_$tau.log | cut -c1-200 | tail -20
done
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; tail -15 $out/pytest.log | cut -c1-300
echo "== pair probe"; ( timeout 600 python scripts/pair_probe.py time ) 2>&1 | grep "^time"
for cfg in "BM_COL_BURST=8" "BM_COL_BURST=0"; do
  echo "== colwise $cfg"; env $cfg python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('   value %.1f  median %.1f us  trmean %.1f us' % (l['value'], l['per_gar']['median']['avg_ms']*1e3, l['per_gar']['trmean']['avg_ms']*1e3))"
done
for cfg in "BM_STEP_BURST=8" "BM_STEP_BURST=0"; do
  echo "== step $cfg"
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$cfg -o s -- python bench.py --workload step --steps 15 --no-cpu-baseline --no-traffic > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench_$cfg.json').read().strip().splitlines() if x.startswith('{')][-1])
print('   ms_per_step %.4f' % l['ms_per_step'])
for r in csv.DictReader(open('$out/st_$cfg/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 20000:
        print('   %-60s calls %3s avg %9.1f us' % (r['Name'][:60].replace('void bm::','').replace('bm::',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- python bench.py --no-cpu-baseline --no-traffic > $out/bench.json 2> $out/bench.err
python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench.json').read().strip().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'])
for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps','distance_pass_ms','scalar_form_ms')})
for r in csv.DictReader(open('$out/trace/s_kernel_stats.csv')):
    if float(r['AverageNs']) > 20000 and 'bm::' in r['Name']:
        print('   %-60s calls %4s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY

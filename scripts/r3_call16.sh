#!/bin/bash
# Round 3, GPU call 16: validation of the final tree — smoke, GPU suite, the default bench as the driver runs it, the
# same under the kernel trace.
out=gpurun_out/r3c16
mkdir -p $out
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -E "smoke|Error|error" | tail -3
( time timeout 1800 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|^E  " $out/pytest.log | cut -c1-300 | tail -15
echo "== driver-style default bench"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -4 $out/bench_default.err
python3 - <<PY
import json
l=json.loads([x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], l['roofline']['traffic'])
for k,v in (l['roofline'].get('traffic_per_kernel') or {}).items(): print('  traffic', k, v['traffic'], round(v['ratio'],4))
for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps','distance_pass_ms','scalar_form_ms')}, (v.get('cpu_baseline') or {}).get('value'))
print('cpu', l.get('cpu_baseline',{}).get('value'), l.get('cpu_baseline',{}).get('cores'))
PY
echo "== the same under the kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- python bench.py --no-cpu-baseline --no-traffic > $out/bench_trace.json 2> $out/bench_trace.err
python3 - <<PY
import csv
for r in csv.DictReader(open('$out/trace/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 20000:
        print('   %-78s calls %4s avg %9.1f us' % (r['Name'][:78], r['Calls'], float(r['AverageNs'])/1e3))
PY

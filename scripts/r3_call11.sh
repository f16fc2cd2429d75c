#!/bin/bash
# Round 3, GPU call 11: validation of the final tree (smoke, GPU suite, driver-style default bench) and SQ / traffic
# counter passes of the C5 step and of C3 (scripts/pmc_collect.sh: every counter group is its own rocprofv3 run with
# --kernel-trace only).
out=gpurun_out/r3c11
mkdir -p $out
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -E "smoke|Error|error" | tail -3
( time timeout 1800 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $out/pytest.log | cut -c1-300 | tail -15
echo "== driver-style default bench"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -4 $out/bench_default.err
python3 - <<PY
import json
l=json.loads([x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], l['roofline']['traffic'], l['config'].get('row_placement','')[:60])
for k,v in (l['roofline'].get('traffic_per_kernel') or {}).items(): print('  traffic', k, v['traffic'], round(v['ratio'],4))
for k,v in l['per_gar'].items():
    print('  ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('avg_ms','frac_of_8TBps','distance_pass_ms','scalar_form_ms')}, (v.get('cpu_baseline') or {}).get('value'))
print('cpu', l.get('cpu_baseline',{}).get('value'), l.get('cpu_baseline',{}).get('cores'))
PY
echo "== counters: C5 step (krum)"
bash scripts/pmc_collect.sh $out step -- python bench.py --workload step --steps 6 --no-cpu-baseline --no-traffic > $out/pmc_step.txt 2>&1
grep -A12 -E "momentum_stats_kernel<20|study_stats_kernel<true, 3|gram3_partial_kernel<7|selected_mean_burst" $out/pmc_step.txt | cut -c1-120 | head -80
echo "== counters: C3 (krum)"
bash scripts/pmc_collect.sh $out krum -- python bench.py --workload krum --steps 6 --no-cpu-baseline --no-traffic > $out/pmc_krum.txt 2>&1
grep -A12 -E "gram3_partial_kernel<13" $out/pmc_krum.txt | cut -c1-120 | head -30

#!/bin/bash
# full validation: GPU suite, smoke, default bench (timed), per-workload kernel stats
out=gpurun_out/r2c12
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q --durations=10 ) > $out/pytest.log 2>&1; tail -22 $out/pytest.log
( time python bench.py ) > $out/bench_default.json 2> $out/bench_default.err; tail -4 $out/bench_default.err
python - <<PY
import json
l=[x for x in open('$out/bench_default.json').read().strip().splitlines() if x.startswith('{')][-1]
l=json.loads(l)
print('value',l['value'],'frac',l['roofline']['frac'],'traffic',l['roofline']['traffic'])
for k,v in l['per_gar'].items(): print(k, round(v['avg_ms'],4),'ms', round(v['gbps']),'GB/s', v.get('distance_pass_ms'))
print(l.get('cpu_baseline',{}).get('value'))
PY
for w in krum bulyan step; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$w -o s -- python bench.py --workload $w --steps 15 --no-cpu-baseline --no-traffic > $out/bench_$w.json 2> $out/bench_$w.err
  python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench_$w.json').read().strip().splitlines() if x.startswith('{')][-1])
print('== $w ms_per_step %.4f' % l['ms_per_step'])
for r in csv.DictReader(open('$out/st_$w/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 5000:
        print('   %-42s calls %3s avg %9.1f us' % (r['Name'].split('(')[0].replace('void bm::','').replace('bm::',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
done

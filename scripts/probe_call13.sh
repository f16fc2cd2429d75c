#!/bin/bash
out=gpurun_out/r2c13
mkdir -p $out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; tail -6 $out/pytest.log
for cfg in "BM_PAIR_DEDUPE=1" "BM_PAIR_DEDUPE=0"; do
  tag=$(echo $cfg | tr '=' '_')
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$tag -o s -- python bench.py --workload step --steps 15 --no-cpu-baseline --no-traffic > $out/bench_$tag.json 2> $out/bench_$tag.err
  python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench_$tag.json').read().strip().splitlines() if x.startswith('{')][-1])
print('== $cfg ms_per_step %.4f' % l['ms_per_step'])
for r in csv.DictReader(open('$out/st_$tag/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 30000:
        print('   %-42s calls %3s avg %9.1f us' % (r['Name'].split('(')[0].replace('void bm::','').replace('bm::',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
done
( BM_PAIR_DEDUPE=1 python scripts/pair_probe.py acc ) 2>&1 | grep "^acc" | awk '{print $2,$3,$4,$6,$8,$10}' | sort | uniq -c | sort -rn | head -5

#!/bin/bash
# Round 3, GPU call 10: the skew sequence continued across allocations (sampled sets, momentum buffers, step outputs):
# C5 step alone and inside the default line, against one allocation per row, alternating on one box.
out=gpurun_out/r3c10
mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for flag in "" "--separate-rows"; do
  for gar in krum median; do
    python bench.py --workload step --gar $gar --steps 15 --no-cpu-baseline --no-traffic $flag 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('%-16s step %-6s %.4f ms' % ('${flag:-slab rows}', '$gar', l['ms_per_step']))"
  done
done; done | tee $out/step_placement_ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- python bench.py --workload step --steps 15 --no-cpu-baseline --no-traffic > $out/bench_step.json 2> $out/bench_step.err
python3 - <<PY
import csv
for r in csv.DictReader(open('$out/trace/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 20000:
        print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); p=l['per_gar']
print('default: value %.1f median %.1f trmean %.1f krum_c3 %.1f bulyan %.1f step krum %.1f median %.1f' % (l['value'], p['median']['avg_ms']*1e3, p['trmean']['avg_ms']*1e3, p['krum_c3']['avg_ms']*1e3, p['bulyan_c4_1gpu']['avg_ms']*1e3, p['step_c5_krum']['avg_ms']*1e3, p['step_c5_median']['avg_ms']*1e3))"

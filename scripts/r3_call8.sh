#!/bin/bash
# Round 3, GPU call 8: placement A/B of the synthetic stacks in the default bench (same box), new tests.
out=gpurun_out/r3c8
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "allocator or row_sqnorms or study_stats or steady" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-250 | tail -8
for rep in 1 2; do
for flag in "" "--separate-rows"; do
  python bench.py --no-cpu-baseline --no-traffic $flag > $out/bench_${rep}_${flag:-slab}.json 2>/dev/null
  python3 - <<PY
import json
l=json.loads([x for x in open('$out/bench_${rep}_${flag:-slab}.json').read().strip().splitlines() if x.startswith('{')][-1])
p=l['per_gar']
print('%-16s value %7.1f  median %.1f trmean %.1f | krum_c3 %.1f (dist %.1f) bulyan_c4 %.1f | aksel %.1f cge %.1f | step krum %.1f median %.1f' % ('${flag:-slab rows}', l['value'], p['median']['avg_ms']*1e3, p['trmean']['avg_ms']*1e3, p['krum_c3']['avg_ms']*1e3, p['krum_c3']['distance_pass_ms']*1e3, p['bulyan_c4_1gpu']['avg_ms']*1e3, p['aksel_c2']['avg_ms']*1e3, p['cge_c2']['avg_ms']*1e3, p['step_c5_krum']['avg_ms']*1e3, p['step_c5_median']['avg_ms']*1e3))
PY
done; done | tee $out/placement_ab.txt

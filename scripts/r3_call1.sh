#!/bin/bash
# Round 3, GPU call 1: the new parity tests, the whole GPU suite, kernel trace of the default bench, dither A/B.
out=gpurun_out/r3c1
mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q ) > $out/pytest_r3.log 2>&1; tail -25 $out/pytest_r3.log
( time timeout 1500 python -m pytest tests -m gpu -q -x --ignore=tests/test_gpu_parity_r3.py ) > $out/pytest.log 2>&1; tail -6 $out/pytest.log
for cfg in "BM_PAIR_DITHER=0" "BM_PAIR_DITHER=-1"; do
  echo "== $cfg"; ( env $cfg timeout 600 python scripts/pair_probe.py time ) 2>&1 | tail -8
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- python bench.py --no-cpu-baseline --no-traffic > $out/bench.json 2> $out/bench.err
tail -c 3000 $out/bench.json
python3 - <<PY
import csv
for r in csv.DictReader(open('$out/trace/s_kernel_stats.csv')):
    if float(r['AverageNs']) > 20000:
        print('   %-60s calls %4s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY

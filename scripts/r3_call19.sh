#!/bin/bash
out=gpurun_out/r3c19
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "riding or burst_form" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|^E  " $out/pytest.log | cut -c1-300 | tail -8
for gar in phocas meamed; do
  python bench.py --workload step --gar $gar --steps 12 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('step $gar: %.4f ms' % l['ms_per_step'])"
done

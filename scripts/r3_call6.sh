#!/bin/bash
out=gpurun_out/r3c6
mkdir -p $out
export TMPDIR=/tmp
for b in 8 0; do echo "== BM_STEP_BURST=$b"; BM_STEP_BURST=$b timeout 600 python scripts/momentum_layout_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out/momentum_layout_probe.txt; done
echo "== step workload alone (fresh allocator)"; python bench.py --workload step --steps 12 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('  step alone ms', l['ms_per_step'])"

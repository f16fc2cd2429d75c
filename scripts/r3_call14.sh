#!/bin/bash
out=gpurun_out/r3c14
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -x -k "distance_pass_riding" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
timeout 600 python scripts/fused_distance_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/fused_distance_probe.txt

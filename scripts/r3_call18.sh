#!/bin/bash
out=gpurun_out/r3c18
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "fed_from_the_first_pass or riding" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
timeout 600 python scripts/fused_distance_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/fused_distance_probe.txt
( timeout 900 python -m pytest tests/test_gpu_parity_r2.py -m gpu -q -k "c5_step or single_call" ) 2>&1 | tail -2

#!/bin/bash
out=gpurun_out/r3c22
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "update_placement" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
( timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -m gpu -q -x -k "placements or step or sharded or rccl" ) 2>&1 | tail -3

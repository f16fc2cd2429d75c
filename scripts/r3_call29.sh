#!/bin/bash
# Bulyan's factor search ranked from scalars: the GPU tests that touch it
out=gpurun_out/r3c29
mkdir -p $out
( time timeout 30 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -m gpu -x -q -k "bulyan and (search or factor)" ) > $out/pytest_bulyan_search.log 2>&1; grep -E "passed|failed|rror" $out/pytest_bulyan_search.log | tail -5

#!/bin/bash
out=gpurun_out/r3c17
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "fed_from_the_first_pass or riding or burst_form" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|^E  " $out/pytest.log | cut -c1-300 | tail -8

#!/bin/bash
out=gpurun_out/r2c5
mkdir -p $out
export TMPDIR=/tmp
( python scripts/pair_probe.py dup time:25,51 ) > $out/pair.log 2>&1; grep -h "^dup\|^time" $out/pair.log
( BM_PAIR_MODE=2 python scripts/pair_probe.py time:25,51 ) > $out/pair_mode2.log 2>&1; grep -h "^time" $out/pair_mode2.log
( BM_PAIR_MODE=1 python scripts/pair_probe.py time:25 ) > $out/pair_mode1.log 2>&1; grep -h "^time" $out/pair_mode1.log
timeout 2400 python -m pytest tests/test_gpu_parity_r2.py "tests/test_gpu_parity.py::test_rccl_path_on_one_gpu" "tests/test_gpu_parity.py::test_direct_difference_pairwise_mode" "tests/test_gpu_parity.py::test_seeded_stack_100k" "tests/test_gpu_parity.py::test_largest_row_counts_distance_rules" -q -s --durations=12 > $out/pytest.log 2>&1
tail -25 $out/pytest.log

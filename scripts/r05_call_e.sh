#!/bin/bash
# Round-5 call e: in-process A/Bs — ranking algorithm, load policy of the column kernel; the launcher tests again.
set -u
out=gpurun_out/r05_e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
timeout 300 python scripts/rank_probe.py > $out/rank_probe.txt 2>&1
timeout 300 python scripts/col_placement_probe.py > $out/col_load_policy_probe.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_y_bench_exchange.py -m gpu -q 2>&1 | tail -15 > $out/pytest_launcher.log
ls -la $out

#!/bin/bash
# Round-5 first GPU call: full suite on the current tree, driver-style bench, a clean kernel trace of the headline
# kernels, the 2 x 2 A/B of the second-pass walk / non-temporal hint.  Outputs under gpurun_out/r05_a/.
set -u
out=gpurun_out/r05_a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > $out/pytest_gpu.log 2>&1
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_colwise -o colwise -- python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 20 > $out/prof_colwise.log 2>&1
find $out/prof_colwise -name "*kernel_stats.csv" -exec cp {} $out/colwise_only_kernel_stats.csv \;
for nt in 1 0; do for r in 0 1 0 1; do
  BM_PAIR_LOAD_NT=$nt BM_SECOND_PASS_REVERSE=$r timeout 200 python scripts/second_pass_walk_probe.py
done; done > $out/second_pass_walk_ab.txt 2>&1
rm -rf $out/prof_colwise/*/*.db 2>/dev/null
ls -la $out

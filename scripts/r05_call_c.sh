#!/bin/bash
# Round-5 third GPU call: the suite without the multi-process files, bench default, kernel trace of it.
set -u
out=gpurun_out/r05_c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_zz_multirank.py --deselect tests/test_gpu_reference_loop.py 2>&1 | tail -60 ) > $out/pytest_gpu.log 2>&1
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full -o full -- python bench.py --no-cpu-baseline --no-traffic --steps 10 > $out/prof_full.log 2>&1
find $out/prof_full -name "*kernel_stats.csv" -exec cp {} $out/full_kernel_stats.csv \;
find $out/prof_full -name "*kernel_trace.csv" -exec cp {} $out/full_kernel_trace.csv \;
rm -rf $out/prof_full
ls -la $out

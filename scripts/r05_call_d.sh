#!/bin/bash
# Round-5 call d: ranking by counting + launcher tests, bench default.
set -u
out=gpurun_out/r05_d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
( time timeout 600 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_y_bench_exchange.py tests/test_gpu_y_rccl_one_rank.py -m gpu -q -k "krum or bulyan or brute or rank or golden or full_size or launcher or deadline or rccl or step or search" 2>&1 | tail -40 ) > $out/pytest_gpu.log 2>&1
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python bench.py --workload bulyan --no-cpu-baseline --no-traffic --steps 10 > $out/prof_bulyan.log 2>&1
find $out/prof -name "*kernel_trace.csv" -exec cp {} $out/bulyan_kernel_trace.csv \;
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof2 -o t -- python bench.py --workload krum --no-cpu-baseline --no-traffic --steps 10 > $out/prof_krum.log 2>&1
find $out/prof2 -name "*kernel_trace.csv" -exec cp {} $out/krum_kernel_trace.csv \;
rm -rf $out/prof $out/prof2
ls -la $out

#!/bin/bash
# Round-5 call i: rules on poisoned memory; the multi-rank file behind the full-size file, twice, full output kept.
set -u
out=gpurun_out/r05_i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
timeout 400 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q > $out/poison.log 2>&1; tail -3 $out/poison.log
for i in 1 2; do
  timeout 500 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q > $out/multirank_after_full_size_$i.log 2>&1
  tail -3 $out/multirank_after_full_size_$i.log
done
ls -la $out

#!/bin/bash
# Round-5 call j: the multi-rank file behind the full-size file WITHOUT the allocator poisoning, three times, full output.
set -u
out=gpurun_out/r05_l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
for i in 1 2; do
  BM_TEST_POISON=0 timeout 500 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q > $out/multirank_after_full_size_$i.log 2>&1
  tail -3 $out/multirank_after_full_size_$i.log
done
ls -la $out

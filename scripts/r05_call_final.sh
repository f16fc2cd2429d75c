#!/bin/bash
# Round-5 final call: the whole GPU suite as the driver runs it (-x), smoke, bench default as the driver runs it, then —
# with what is left of the budget — the multi-rank file behind the full-size file without the allocator poisoning.
set -u
out=gpurun_out/r05_final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
( time timeout 1000 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
for i in 1 2; do
  BM_TEST_POISON=0 timeout 400 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q > $out/multirank_after_full_size_$i.log 2>&1
  tail -2 $out/multirank_after_full_size_$i.log
done
ls -la $out

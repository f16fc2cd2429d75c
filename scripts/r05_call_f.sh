#!/bin/bash
# Round-5 call f: the Gram kernel against its version before commit 369a08a (A/B), the brute search, bench default.
set -u
out=gpurun_out/r05_f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
for i in 1 2 3; do
  timeout 120 python scripts/gram_variant_probe.py
  BM_GAR_LIB=scratch/gram_parent/libbm_gar_gram_parent.so timeout 120 python scripts/gram_variant_probe.py
done > $out/gram_scalar_wave_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r4.py tests/test_gpu_parity_r3.py -m gpu -q -k "brute or golden or step" 2>&1 | tail -15 > $out/pytest_brute.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
ls -la $out

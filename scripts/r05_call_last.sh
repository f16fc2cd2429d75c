#!/bin/bash
# Round-5 last call: where the scalar-form factor search spends its time; the multi-rank file behind the full-size file.
set -u
out=gpurun_out/r05_last; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python scripts/attack_search_probe.py > $out/attack_search_probe.txt 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/attack_search_kernel_stats.csv \;
rm -rf $out/prof
timeout 330 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q -rx > $out/multirank_after_full_size.log 2>&1
tail -5 $out/multirank_after_full_size.log

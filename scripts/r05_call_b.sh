#!/bin/bash
# Round-5 second GPU call: the multi-rank file in a loop with the mismatch diagnostics, the placement probe.
set -u
out=gpurun_out/r05_b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
timeout 400 python scripts/col_placement_probe.py > $out/col_placement_probe.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_colwise -o colwise -- python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 20 > $out/prof_colwise.log 2>&1
find $out/prof_colwise -name "*kernel_stats.csv" -exec cp {} $out/colwise_only_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full -o full -- python bench.py --no-cpu-baseline --no-traffic --steps 10 > $out/prof_full.log 2>&1
find $out/prof_full -name "*kernel_stats.csv" -exec cp {} $out/full_kernel_stats.csv \;
find $out/prof_full -name "*kernel_trace.csv" -exec cp {} $out/full_kernel_trace.csv \;
rm -rf $out/prof_colwise $out/prof_full
for i in 1 2 3 4 5; do
  timeout 300 python -m pytest tests/test_gpu_zz_multirank.py -m gpu -q -x -W always 2>&1 | tail -120 > $out/multirank_$i.log
  grep -q "passed" $out/multirank_$i.log && ! grep -q "failed\|repeating" $out/multirank_$i.log && echo "run $i clean" >> $out/multirank_summary.txt || echo "run $i NOT clean" >> $out/multirank_summary.txt
done
ls -la $out

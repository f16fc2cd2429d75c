#!/bin/bash
# Round-5 end: the multi-rank file behind the full-size file once more (default settings), the report flags shown.
set -u
out=gpurun_out/r05_end; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
timeout 280 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q -rxX > $out/multirank_after_full_size.log 2>&1
tail -6 $out/multirank_after_full_size.log

#!/bin/bash
# Round-5 call g: the whole GPU suite on the current tree (no -x: every failure shows).
set -u
out=gpurun_out/r05_g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
( time timeout 1100 python -m pytest tests -m gpu -q -W always 2>&1 | tail -80 ) > $out/pytest_gpu.log 2>&1
ls -la $out

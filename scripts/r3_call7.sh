#!/bin/bash
out=gpurun_out/r3c7
mkdir -p $out
timeout 900 python scripts/layout_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/layout_probe.txt

#!/bin/bash
# pk.hip with the compiler's defaults, ref.hip without packed fp32, main.hip: one binary (see main.hip).
set -e
here=$(cd $(dirname $0) && pwd)
F="-O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off"
/opt/rocm/bin/hipcc $F -c $here/pk.hip -o $here/pk.o
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -c $here/ref.hip -o $here/ref.o 2>&1 | grep -v "packed-fp32-ops" || true
/opt/rocm/bin/hipcc $F -c $here/main.hip -o $here/main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o $here/slp_pair_probe $here/main.o $here/pk.o $here/ref.o
ls -la $here/slp_pair_probe

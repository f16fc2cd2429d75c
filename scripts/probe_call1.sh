#!/bin/bash
# GPU call 1 of round 2: new bf16x3 centred Gram kernel vs the fp32 Gram and direct kernels.
mkdir -p gpurun_out/r2c1
out=gpurun_out/r2c1
( python scripts/pair_probe.py acc time ) > $out/mode0.log 2>&1
( BM_PAIR_CENTRE=0 python scripts/pair_probe.py acc time ) > $out/mode0_nocentre.log 2>&1
( BM_PAIR_MODE=2 python scripts/pair_probe.py acc time ) > $out/mode2.log 2>&1
( BM_PAIR_TAU=0 python scripts/pair_probe.py acc ) > $out/mode0_nogate.log 2>&1
( BM_PAIR_MODE=2 BM_PAIR_TAU=0 python scripts/pair_probe.py acc ) > $out/mode2_nogate.log 2>&1
( BM_PAIR_MODE=1 python scripts/pair_probe.py time ) > $out/mode1.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
tail -3 $out/pytest.log
grep -h "^time" $out/*.log

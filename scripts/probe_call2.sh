#!/bin/bash
# GPU call 2 of round 2: kernel v2 (2/3 planes, per-chunk flush, median-of-3 centre).
out=gpurun_out/r2c2
mkdir -p $out
export TMPDIR=/tmp
( python scripts/pair_probe.py acc time:25,51,16,64 ) > $out/default.log 2>&1
( BM_PAIR_PLANES=3 python scripts/pair_probe.py time:25,51 ) > $out/planes3.log 2>&1
( BM_PAIR_PLANES=2 python scripts/pair_probe.py acc ) > $out/planes2_acc.log 2>&1
( BM_PAIR_CENTRE=0 python scripts/pair_probe.py time:25,51 ) > $out/centre0.log 2>&1
( BM_PAIR_CENTRE=1 python scripts/pair_probe.py time:25,51 ) > $out/centre1.log 2>&1
( BM_PAIR_TAU=0 python scripts/pair_probe.py acc ) > $out/nogate_acc.log 2>&1
( BM_PAIR_BLOCKS=256 python scripts/pair_probe.py time:25,51 ) > $out/blocks256.log 2>&1
( BM_PAIR_BLOCKS=1024 python scripts/pair_probe.py time:25,51 ) > $out/blocks1024.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python scripts/pair_probe.py time:51,25 > $out/rocprof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r2c2_pmc1 -o r2c2_pmc1 -- python scripts/pair_probe.py time:51 > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES --kernel-trace --output-format csv -d $out/r2c2_pmc2 -o r2c2_pmc2 -- python scripts/pair_probe.py time:51 > $out/pmc2.log 2>&1
python scripts/pmc_summary.py $out r2c2 > $out/pmc_summary.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity_r2.py -x -q -s > $out/pytest_r2.log 2>&1
tail -5 $out/pytest_r2.log
grep -h "^time" $out/*.log
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs head -8

#!/bin/bash
# PMC passes for one command (run ON the GPU box through gpurun). Each counter group is its own
# rocprofv3 run with --kernel-trace only (no sys/hip/hsa trace domains next to --pmc).
# usage: scripts/pmc_collect.sh <outdir> <tag> -- <command...>
set -u
out=$1; tag=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd /root/repo
mkdir -p "$out"
i=0
for group in \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/${tag}_pmc$i" -o "${tag}_pmc$i" -- "$@" > "$out/${tag}_pmc$i.log" 2>&1
done
python scripts/pmc_summary.py "$out" "$tag"

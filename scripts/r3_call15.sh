#!/bin/bash
# Round 3, GPU call 15: the distance pass of Krum / Bulyan riding along with the first pass of the step.
out=gpurun_out/r3c15
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "fed_from_the_first_pass or riding" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|^E  " $out/pytest.log | cut -c1-300 | tail -8
for gar in krum bulyan; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$gar -o s -- python bench.py --workload step --gar $gar --steps 15 --no-cpu-baseline --no-traffic > $out/bench_$gar.json 2> $out/bench_$gar.err
  python3 - <<PY
import csv, json
l=json.loads([x for x in open('$out/bench_$gar.json').read().strip().splitlines() if x.startswith('{')][-1])
print('== step $gar: ms_per_step %.4f' % (l['ms_per_step'],))
for r in csv.DictReader(open('$out/st_$gar/s_kernel_stats.csv')):
    if 'bm::' in r['Name'] and float(r['AverageNs']) > 6000:
        print('   %-78s calls %3s avg %9.1f us' % (r['Name'][:78].replace('void bm::','').replace('bm::',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
done
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 scripts/per_rank_probe.py ) 2>&1 | grep -E "^C[45]" | tee $out/per_rank_p8.txt

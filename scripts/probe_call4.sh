#!/bin/bash
# GPU call 4 of round 2: full GPU suite + bench lines + rocprof kernel stats for profiles/.
out=gpurun_out/r2c4
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > $out/pytest.log 2>&1
tail -12 $out/pytest.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json,sys
l=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print('value',l['value'],'roofline',l['roofline']['frac'],'traffic',l['roofline']['traffic'])
for k,v in l['per_gar'].items(): print(k, round(v['avg_ms'],4),'ms', round(v['gbps']),'GB/s', v.get('distance_pass_ms'))
"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_krum -o krum -- python bench.py --workload krum --steps 20 --no-cpu-baseline --no-traffic > $out/bench_krum.json 2> $out/bench_krum.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_bulyan -o bulyan -- python bench.py --workload bulyan --steps 20 --no-cpu-baseline --no-traffic > $out/bench_bulyan.json 2> $out/bench_bulyan.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_step -o step -- python bench.py --workload step --steps 10 --no-cpu-baseline --no-traffic > $out/bench_step.json 2> $out/bench_step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_colwise -o colwise -- python bench.py --steps 20 --no-cpu-baseline --no-traffic --no-extras > $out/bench_colwise.json 2> $out/bench_colwise.err
for w in krum bulyan step colwise; do echo "== $w"; find $out/stats_$w -name "*kernel_stats.csv" | head -1 | xargs cut -c1-160 | grep -v "at::native" | head -12; done

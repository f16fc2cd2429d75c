#!/bin/bash
# GPU call 3 of round 2: full GPU suite, smoke, bench (all workloads), kernel stats.
out=gpurun_out/r2c3
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1
tail -8 $out/pytest.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 3000 $out/bench_default.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload bulyan --steps 30 --no-traffic > $out/bench_bulyan_torchrun.json 2> $out/bench_bulyan_torchrun.err; tail -c 1500 $out/bench_bulyan_torchrun.json
( python scripts/pair_probe.py time:25,51 ) > $out/pair_default.log 2>&1; grep -h "^time" $out/pair_default.log

#!/bin/bash
out=gpurun_out/r3c21
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "update_placement" ) > $out/pytest_r3.log 2>&1; grep -E "^FAILED|passed|failed|^E  " $out/pytest_r3.log | cut -c1-300 | tail -12
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|^E  " $out/pytest.log | cut -c1-300 | tail -8
python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import byzantinemomentum_amd as bm
from byzantinemomentum_amd.step import AggregationStep
d, n, f = 36546980, 25, 5
h = n - f
gen = torch.Generator(device="cuda").manual_seed(1)
mu = 0.1 * torch.randn(d, device="cuda", generator=gen)
sets = [[mu + s * torch.randn(d, device="cuda", generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()] for _ in range(2)]
for gar in ("median", "krum"):
  step = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, momentum_at="update", attack_factor=1.1, nb_past=25)
  for i in range(28):
    step.run(sets[i & 1]); step.floats()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(10):
    step.run(sets[i & 1]); step.floats()
  torch.cuda.synchronize()
  print(f"C5-size step, momentum at the update, rule {gar}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
PY

#!/bin/bash
# bench.attack_search for the median / Bulyan entries on a small stack (the default line's new side entries)
mkdir -p gpurun_out/r3c30
timeout 18 python - > gpurun_out/r3c30/attack_search_small.txt 2>&1 <<'PY'
import importlib.util, torch, json
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import byzantinemomentum_amd as bm
dev = torch.device("cuda", 0); n, f, d = 25, 5, 1 << 20
gen = torch.Generator(device=dev).manual_seed(1)
hon = [0.1 * torch.randn(d, device=dev, generator=gen) + s * torch.randn(d, device=dev, generator=gen) for s in torch.linspace(0.5, 1.5, n - f).tolist()]
for gar in ("bulyan", "median"):
  print(json.dumps(b.attack_search(bm, hon, n, f, d, gar=gar)))
PY
tail -3 gpurun_out/r3c30/attack_search_small.txt

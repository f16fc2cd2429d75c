"""The factor search against Multi-Krum in its scalar form (one distance pass over h + 2 rows, then the host): where
its time goes.  `per_gar.attack_search_c3_krum.scalar_form_ms` was 0.54-0.57 ms on two boxes of round 5 and 4.4-6.1 ms
on three others (0.54-0.56 ms in every run of rounds 3-4).

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python scripts/attack_search_probe.py
"""
import os
import sys
import time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import byzantinemomentum_amd as bm
from byzantinemomentum_amd.step import AggregationStep

dev = torch.device("cuda:0")
n, f, d = 51, 12, bench.D_RESNET18
bench.SEPARATE_ROWS = True
stacks = bench.make_stacks(n, f, d, dev, 1, 4321, False)
honests = stacks[0][:n - f]
avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
runner = AggregationStep(n, f, f, gar="krum", attack_evals=16, line_search="host", nb_past=0)  # the host form: the one with a copy and a host leg to time
runner._search_factor(honests, avg, direction)
torch.cuda.synchronize()
for rep in range(5):
  t0 = time.perf_counter()
  rows = list(honests) + [avg, direction]
  sq = runner.agg.global_sqdist(rows)
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  ext = sq.cpu().contiguous()
  t2 = time.perf_counter()
  factor = runner._search_factor(honests, avg, direction)
  torch.cuda.synchronize()
  t3 = time.perf_counter()
  gate = bm.gars._workspace(dev, bm._lib.WS_PAIRWISE, len(rows), d, "ws_pair")[:8].view(torch.int32)[:2].tolist()
  print(f"rep {rep}: distance pass over {len(rows)} rows {(t1 - t0) * 1e3:.3f} ms, copy out {(t2 - t1) * 1e3:.3f} ms, whole search "
        f"{(t3 - t2) * 1e3:.3f} ms (factor {factor}); rows listed by the accuracy gate: {gate[0]}", flush=True)

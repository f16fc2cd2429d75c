"""Where do the ~175 us of attack_search_kernel go?  Three forms of the kernel (one wave from LDS tables, one wave with
batched loads, sixteen waves with the rows in registers) took the same time, so the candidates' arithmetic is not it.
Hypothesis: a cold start — the kernel is ~15 KB of straight-line code that a d-sized kernel in front of it has pushed
out of the instruction cache and of L2.  This probe times the kernel alone with events: back to back (warm) and behind
a 1 GB fill (cold), at 1 / 4 / 16 / 64 evaluations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402

dev = torch.device("cuda:0")
h, k, f, d = 39, 12, 12, 200003
gen = torch.Generator(device=dev).manual_seed(3)
base = torch.randn(d, device=dev, generator=gen)
honests = [base + (0.3 + 0.02 * i) * torch.randn(d, device=dev, generator=gen) for i in range(h)]
avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
unit = torch.empty_like(avg)
bm.stats.multi_fma3([unit], [avg], [direction], 1.0, 1.0)
sq = bm.gars.pairwise_sqdist(honests + [avg, unit])
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
torch.cuda.synchronize()


def timed(evals, cold, reps=8):
  out = []
  for _ in range(reps):
    if cold:
      big.fill_(1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    bm.stats.attack_search_device(sq, h, k, f, "krum", evals=evals)
    b.record()
    torch.cuda.synchronize()
    out.append(a.elapsed_time(b) * 1e3)
  return out


for rule_evals in (16, 1, 4, 16, 64):
  for cold in (False, True):
    each = timed(rule_evals, cold)
    print(f"evals={rule_evals:3d} {'behind a 1 GB fill' if cold else 'back to back      '}: " + " ".join(f"{v:7.1f}" for v in each) + " us", flush=True)

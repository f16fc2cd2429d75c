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

# Phase timestamps of one traced launch (BM_SEARCH_TRACE=1: csrc/search_device.hip, attack_search_kernel<true>)
os.environ["BM_SEARCH_TRACE"] = "1"
evals = 16
for _ in range(2):
  out = bm.stats.attack_search_device(sq, h, k, f, "krum", evals=evals)
torch.cuda.synchronize()
tr = out[1 + 2 * evals:].view(evals, 2, 12).cpu()
names = ["propose", "dq+share+B0", "bin. search", "fold", "B1", "partials+B2", "rank sum", "row sums", "butterfly", "objective", "B3+report"]
print("wave 0: cycles since the previous stamp (2.4 GHz), per candidate; last column = the candidate's total")
print("        " + " ".join(f"{n[:11]:>11s}" for n in names[1:]) + "       total |  wave 1 (Byzantine row) done")
for e in range(evals):
  w0 = tr[e, 0, :11].tolist()
  deltas = [w0[i] - w0[i - 1] for i in range(1, 11)]
  w15 = tr[e, 1, 3].item() - w0[0]
  print(f"  e={e:2d}  " + " ".join(f"{v:11.0f}" for v in deltas) + f" {w0[10] - w0[0]:11.0f} | {w15:11.0f}")
print(f"first candidate starts {tr[0, 0, 0].item():.0f} cycles into the kernel (the set-up)")
os.environ["BM_SEARCH_TRACE"] = "0"

"""Time bm_colwise (median, trmean) for several (n, d); run once per value of BM_COL_BURST (the library reads its
knobs once per process) for A/B comparisons inside one gpurun call.  Checks the two forms against each other
through a checksum of the whole result."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  cases = [(25, 5, 11173962), (25, 5, 36546980), (25, 5, 5000000), (25, 5, 2800000), (8, 1, 11173962), (15, 3, 11173962),
           (19, 4, 11173962), (26, 5, 11173962), (11, 2, 11173962)]
  if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
  for n, f, d in cases:
    gen = torch.Generator(device=dev).manual_seed(3)
    stacks = [[torch.randn(d, device=dev, generator=gen) for _ in range(n)] for _ in range(2)]
    line = f"BM_COL_BURST={os.environ.get('BM_COL_BURST', 'default')} n={n} d={d}:"
    for name, fn in (("median", lambda st: bm.median(st)), ("trmean", lambda st: bm.trmean(st, f))):
      for i in range(3):
        out = fn(stacks[i & 1])
      torch.cuda.synchronize()
      reps, rounds = 20, 5
      us = []
      for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
          out = fn(stacks[i & 1])
        b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / reps)
      us.sort()
      nbytes = 4 * d * (n + 1)
      line += f" {name} {us[rounds // 2]:.1f} us = {nbytes / us[rounds // 2] / 1e3:.0f} GB/s (sum {float(out.double().sum()):.6f});"
    print(line)
    del stacks
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()

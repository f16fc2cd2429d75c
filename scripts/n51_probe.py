"""The six rules of reproduce.py:122-162 at n = 51, f = 12, d = 11 173 962 (one torch.empty per row, as attack.py hands
them over): median time of each whole rule over 5 rounds of `reps` calls + a checksum of its output (bits must not move
when a kernel is rewritten).  With an argument: that many calls per rule, no timing — the command scripts/pmc_collect.sh
profiles.
    python scripts/n51_probe.py            python scripts/n51_probe.py 3"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  n, f, d = 51, 12, 11173962
  gen = torch.Generator(device=dev).manual_seed(3)
  base = 0.1 * torch.randn(d, device=dev, generator=gen)
  stacks = [[base + (0.5 + 0.02 * i) * torch.randn(d, device=dev, generator=gen) for i in range(n)] for _ in range(2)]
  rules = (("median", lambda st: bm.median(st), n + 1), ("trmean", lambda st: bm.trmean(st, f), n + 1),
           ("phocas", lambda st: bm.phocas(st, f), n + 1), ("meamed", lambda st: bm.meamed(st, f), n + 1),
           ("bulyan", lambda st: bm.bulyan(st, f), n + (n - f - 2) + 1), ("aksel", lambda st: bm.aksel(st, f), n + (n + 1) // 2 + 1),
           ("krum", lambda st: bm.krum(st, f), n + (n - f - 2) + 1))
  only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  for name, fn, units in rules:
    if only:
      for i in range(only):
        bm.gars.invalidate_rank_cache()
        out = fn(stacks[i & 1])
      torch.cuda.synchronize()
      continue
    for i in range(3):
      out = fn(stacks[i & 1])
    torch.cuda.synchronize()
    reps, rounds = 10, 5
    us = []
    for _ in range(rounds):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for i in range(reps):
        bm.gars.invalidate_rank_cache()
        out = fn(stacks[i & 1])
      b.record()
      torch.cuda.synchronize()
      us.append(a.elapsed_time(b) * 1e3 / reps)
    us.sort()
    nbytes = 4 * d * units
    print(f"n=51 f=12 d={d} {name:7s} {us[rounds // 2]:8.1f} us  {nbytes / us[rounds // 2] / 1e3:6.0f} GB/s  frac {nbytes / us[rounds // 2] / 8e6:.3f}"
          f"  checksum {float(out.double().sum()):.9e} {float(out.double().abs().sum()):.9e}", flush=True)


if __name__ == "__main__":
  main()

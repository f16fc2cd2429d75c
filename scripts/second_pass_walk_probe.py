"""A/B of the walking order of the second pass of the two-pass rules:

    for nt in 1 0; do for r in 0 1 0 1; do BM_PAIR_LOAD_NT=$nt BM_SECOND_PASS_REVERSE=$r python scripts/second_pass_walk_probe.py; done; done

in ONE gpurun call (the library reads its knobs once per process; boxes differ by more than the effect).  Whole rules —
distance pass, ranking, second pass — on two alternating stacks at the bench's shapes, so that the second pass of a
call finds in the Infinity Cache what the distance pass of the SAME call left there and nothing of the previous call;
plus the 2-rank and 8-rank shard lengths of C4 (where 46 % / all of the shard is still cached).  BM_PAIR_LOAD_NT=0 makes the distance pass load with the default cache
policy instead of the non-temporal hint (the hint may be what keeps the rows out of the cache).  The checksums of the two
settings must agree to the last digit: the walk does not change a column's arithmetic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402

D = 11173962


def stack(n, f, d, dev, gen):
  mu = 0.1 * torch.randn(d, device=dev, generator=gen)
  honest = [mu + s * torch.randn(d, device=dev, generator=gen) for s in torch.linspace(0.5, 1.5, n - f).tolist()]
  byz = torch.stack(honest).mean(dim=0).mul_(-0.1)
  return honest + [byz + 0.3 * torch.randn(d, device=dev, generator=gen) for _ in range(f)]


def main():
  dev = torch.device("cuda:0")
  knob = os.environ.get("BM_SECOND_PASS_REVERSE", "default(1)") + " BM_PAIR_LOAD_NT=" + os.environ.get("BM_PAIR_LOAD_NT", "default(1)")
  cases = [("krum C3", 51, 12, D, lambda st, f: bm.krum(st, f)),
           ("bulyan C4", 25, 5, D, lambda st, f: bm.bulyan(st, f)),
           ("bulyan C4, shard of 2 ranks", 25, 5, -(-D // 2), lambda st, f: bm.bulyan(st, f)),
           ("bulyan C4, shard of 8 ranks", 25, 5, -(-D // 8), lambda st, f: bm.bulyan(st, f)),
           ("aksel C2", 25, 5, D, lambda st, f: bm.aksel(st, f)),
           ("cge C2", 25, 5, D, lambda st, f: bm.cge(st, f))]
  for name, n, f, d, rule in cases:
    gen = torch.Generator(device=dev).manual_seed(3)
    stacks = [stack(n, f, d, dev, gen) for _ in range(2)]
    for i in range(4):
      out = rule(stacks[i & 1], f)
    torch.cuda.synchronize()
    reps, rounds, us = 20, 5, []
    for _ in range(rounds):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for i in range(reps):
        out = rule(stacks[i & 1], f)
      b.record()
      torch.cuda.synchronize()
      us.append(a.elapsed_time(b) * 1e3 / reps)
    us.sort()
    print(f"BM_SECOND_PASS_REVERSE={knob} {name} (n={n}, f={f}, d={d}): {us[rounds // 2]:.1f} us per rule (best round "
          f"{us[0]:.1f}); checksum {float(out.double().sum()):.9f}", flush=True)
    del stacks, out
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()

"""Time bm_bulyan_pass2 ALONE (given rankings) at C4 (n=25, f=5, d=11 173 962) and two neighbours, once per value of an
environment knob of the library — which reads its knobs once per process — for A/B comparisons inside ONE gpurun call:

    for v in 0 1 0 1; do BM_BULYAN_SHORT=$v python scripts/bulyan_pass2_probe.py; done

(knob of this kernel: BM_BULYAN_SHORT, the window search).  The checksums of the runs must agree to the last digit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402

if os.environ.get("BM_PROBE_LIB"):  # experiment only: time another build of the library (A/B inside one gpurun call)
  import pathlib
  bm._lib.LIB_PATH = pathlib.Path(os.environ["BM_PROBE_LIB"]).resolve()


def main():
  dev = torch.device("cuda:0")
  for n, f in ((25, 5), (51, 12), (15, 3)):
    d = 11173962
    gen = torch.Generator(device=dev).manual_seed(3)
    mu = 0.1 * torch.randn(d, device=dev, generator=gen)
    stacks = []
    for _ in range(2):
      honest = [mu + s * torch.randn(d, device=dev, generator=gen) for s in torch.linspace(0.5, 1.5, n - f).tolist()]
      byz = torch.stack(honest).mean(dim=0).mul_(-0.1)
      stacks.append(honest + [byz + 0.3 * torch.randn(d, device=dev, generator=gen) for _ in range(f)])
    m = n - f - 2
    orders = [bm.gars._rank(st, f, m, bm._lib.RANK_BULYAN)[0] for st in stacks]
    for i in range(4):
      out = bm.gars.bulyan_pass2(stacks[i & 1], orders[i & 1], f, m)
    torch.cuda.synchronize()
    reps, rounds = 40, 5
    us = []
    for _ in range(rounds):  # the queue stays full: the kernel (>= 80 us) is longer than the host side of a call
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for i in range(reps):
        out = bm.gars.bulyan_pass2(stacks[i & 1], orders[i & 1], f, m)
      b.record()
      torch.cuda.synchronize()
      us.append(a.elapsed_time(b) * 1e3 / reps)
    us.sort()
    nbytes = 4 * d * (m + 1)
    print(f"lib={os.environ.get('BM_PROBE_LIB', 'in-tree')} BM_BULYAN_SHORT={os.environ.get('BM_BULYAN_SHORT', 'default')} n={n} f={f}: pass 2 {us[rounds // 2]:.1f} us per call "
          f"(best round {us[0]:.1f}) = {nbytes / us[rounds // 2] / 1e3:.0f} GB/s for {m}+1 rows; checksum {float(out.double().sum()):.9f}")
    del stacks
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()

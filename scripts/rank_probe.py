"""Two ways to order a row's distances for the scores (csrc/rank_body.h, BM_RANK_ALGO): counting against a bitonic
network per row.  (a) the rank kernel alone on an n x n matrix, 300 launches back to back; (b) the whole distance pass
with its ranking (bm_pairwise_rank) at the C4 / C3 shapes; ONE process, the two alternate, the orders must agree.

    python scripts/rank_probe.py        # on the MI355X
"""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import byzantinemomentum_amd as bm

dev = torch.device("cuda:0")
lib = bm._lib.load()
D = 11173962


def timed(fn, reps):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


gen = torch.Generator(device=dev).manual_seed(1)
for n, f in ((11, 2), (25, 5), (32, 7), (33, 7), (51, 12), (64, 15)):
  pts = torch.randn(n, 40, device=dev, dtype=torch.float64, generator=gen)
  sq = torch.cdist(pts, pts).pow(2).contiguous()
  orders = {}
  line = f"rank kernel alone, n = {n:2d}:"
  for rep in range(2):
    for algo, name in ((1, "bitonic"), (2, "counting")):
      assert lib.bm_tuning_set(b"BM_RANK_ALGO", algo) == 0
      orders[algo] = bm.gars.rank_from_sqdist(sq, n, f, n - f - 2, bm._lib.RANK_KRUM)[0][:n].tolist()
      line += f"  {name} {timed(lambda: bm.gars.rank_from_sqdist(sq, n, f, n - f - 2, bm._lib.RANK_KRUM), 300):6.2f} us"
  print(line, " same order", orders[1] == orders[2], flush=True)

bench.SEPARATE_ROWS = True
for n, f in ((25, 5), (51, 12)):
  stacks = bench.make_stacks(n, f, D, dev, 2, 4321, False)
  line = f"distance pass + ranking, n = {n}, d = {D}:"
  got = {}
  for rep in range(2):
    for algo, name in ((1, "bitonic"), (2, "counting")):
      assert lib.bm_tuning_set(b"BM_RANK_ALGO", algo) == 0
      got[algo] = bm.gars._rank(stacks[0], f, n - f - 2, bm._lib.RANK_BULYAN)[0][:n].tolist()
      i = [0]

      def one():
        i[0] += 1
        bm.gars._rank(stacks[i[0] & 1], f, n - f - 2, bm._lib.RANK_BULYAN)
      line += f"  {name} {timed(one, 30):7.1f} us"
  print(line, " same order", got[1] == got[2], flush=True)
  del stacks
  torch.cuda.empty_cache()
lib.bm_tuning_set(b"BM_RANK_ALGO", 0)

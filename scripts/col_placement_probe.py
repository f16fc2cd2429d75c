"""Where a wave's loads fall, for rows whose start addresses are congruent (one torch.empty per row: the caller's layout,
attack.py:676,803-804): median / trimmed mean at n = 25, d = 11 173 962 with the PLACED instance of the burst kernel —
BM_COL_ROTATE k (wave w loads the rows in the order (i + k w) mod n) and BM_COL_PAGE_STRIDE s (the four 256-byte quarters
of a wave's 1 KB load lie 2^s * 256 B apart) — against the plain kernel, on separate rows and on the skewed slab.  ONE
process, the settings alternate on the same data (bm_tuning_set), every output is compared with the plain kernel's.

    python scripts/col_placement_probe.py        # on the MI355X
"""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import byzantinemomentum_amd as bm

dev = torch.device("cuda:0")
n, f, d = 25, 5, 11173962
lib = bm._lib.load()


def knobs(rotate, stride):
  assert lib.bm_tuning_set(b"BM_COL_ROTATE", rotate) == 0 and lib.bm_tuning_set(b"BM_COL_PAGE_STRIDE", stride) == 0


def timeit(fn, st, reps=40):
  for i in range(4):
    fn(st[i & 1])
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for i in range(reps):
    fn(st[i & 1])
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


bench.SEPARATE_ROWS = True
separate = bench.make_stacks(n, f, d, dev, 2, 1234, False)
bench.SEPARATE_ROWS = False
slab = bench.make_stacks(n, f, d, dev, 2, 1234, False)
print("row starts modulo 2 MB (separate):", sorted({g.data_ptr() % (1 << 21) for g in separate[0]}),
      " spacing of the first rows / 2 MB:", [(separate[0][i + 1].data_ptr() - separate[0][i].data_ptr()) / (1 << 21) for i in range(4)])
configs = [(0, 0), (1, 0), (2, 0), (7, 0), (12, 0), (0, 2), (0, 4), (0, 5), (0, 6), (0, 7), (0, 8), (0, 10), (1, 7), (7, 4)]
knobs(0, 0)
want = {("sep", "median"): bm.median(separate[0]), ("sep", "trmean"): bm.trmean(separate[0], f)}
for rep in range(2):
  for rotate, stride in configs:
    knobs(rotate, stride)
    same = torch.equal(bm.median(separate[0]), want[("sep", "median")]) and torch.equal(bm.trmean(separate[0], f), want[("sep", "trmean")])
    line = f"rotate {rotate:2d} stride 2^{stride:<2d} x 256 B  same bits {same!s:5s}"
    for name, st in (("separate", separate), ("slab", slab)):
      line += f"   {name}: median {timeit(lambda s: bm.median(s), st):6.1f} us  trmean {timeit(lambda s: bm.trmean(s, f), st):6.1f} us"
    print(line, flush=True)
knobs(0, 0)

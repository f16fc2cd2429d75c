"""Load policy for rows whose start addresses are congruent (one torch.empty per row: the caller's layout,
attack.py:676,803-804): median / trimmed mean at n = 25, d = 11 173 962 with the row loads non-temporal (the default)
and plain (BM_COL_LOAD_PLAIN=1), on separate rows and on the skewed slab.  ONE process, the settings alternate on the
same data (bm_tuning_set), every output is compared with the default's.  (An earlier version of this probe measured two
placements of a wave's loads — row order rotated per wave, the quarters of a wave's load spread apart — that bought
nothing: profiles/r05_b_col_placement_probe.txt; the code went with commit "Placement experiments for congruent rows".)

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
assert lib.bm_tuning_set(b"BM_COL_LOAD_PLAIN", 0) == 0
want = (bm.median(separate[0]), bm.trmean(separate[0], f))
for rep in range(3):
  for plain in (0, 1):
    assert lib.bm_tuning_set(b"BM_COL_LOAD_PLAIN", plain) == 0
    same = torch.equal(bm.median(separate[0]), want[0]) and torch.equal(bm.trmean(separate[0], f), want[1])
    line = f"row loads {'plain       ' if plain else 'non-temporal'}  same bits {same!s:5s}"
    for name, st in (("separate", separate), ("slab", slab)):
      line += f"   {name}: median {timeit(lambda s: bm.median(s), st):6.1f} us  trmean {timeit(lambda s: bm.trmean(s, f), st):6.1f} us"
    print(line, flush=True)
lib.bm_tuning_set(b"BM_COL_LOAD_PLAIN", 0)

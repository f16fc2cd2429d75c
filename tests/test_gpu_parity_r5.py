"""Round 5: the rules on POISONED memory.  torch's caching allocator hands out blocks that were never written by this
process (or that still hold the identical result of the previous call), so a kernel that reads a slot nobody wrote — an
arrival counter assumed zero, an index past the rows, an output block never stored — passes almost always.  Here a fresh
process first fills what the allocator will hand out with NaN bit patterns (0x7fc00000: NaN as a float, 2 143 289 344 as
a count) and then runs one aggregation of every kind against the oracle (`__graft_entry__.smoke`) and the sharded rules
with forced collectives."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

POISON = """
import math, sys, torch
sys.path.insert(0, %r)
dev = "cuda:0"
big = torch.full((1 << 28,), math.nan, dtype=torch.float32, device=dev)
mid = [torch.full((1 << 14,), math.nan, dtype=torch.float32, device=dev) for _ in range(512)]
tiny = [torch.full((128,), math.nan, dtype=torch.float32, device=dev) for _ in range(4096)]
torch.cuda.synchronize()
del big, mid, tiny
""" % ROOT


def _run(body):
  done = subprocess.run([sys.executable, "-c", POISON + body], cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert done.returncode == 0, (done.stdout[-2000:], done.stderr[-4000:])
  return done.stdout


def test_smoke_on_poisoned_memory():
  out = _run("import __graft_entry__\n__graft_entry__.smoke()\n")
  assert "smoke ok" in out


def test_rules_twice_on_poisoned_memory_at_several_shapes():
  """Every rule at n = 11 / 25 / 51 and three lengths (a multiple of 4, ragged, tiny), TWICE on freshly poisoned output
  blocks: the second call must not depend on what the first one left behind."""
  _run('''
import byzantinemomentum_amd as bm
from oracle import gar_oracle as O
def dev_rows(rows):
    seen = {}
    return [seen.setdefault(id(g), g.to(dev)) for g in rows]
def close(a, b, tol):
    scale = max(float(b.abs().max()), 1e-30)
    return float((a.cpu() - b).abs().max()) <= tol * scale
for n, f in ((11, 2), (25, 5), (51, 12)):
    for d in (4096, 40007, 3):
        rows, h = O.make_stack("hetero", n, f, d, seed=n + d)
        dv = dev_rows(rows)
        for rep in range(2):
            junk = [torch.full((d,), math.nan, device=dev) for _ in range(8)]   # poison the blocks of the next outputs
            del junk
            assert torch.equal(bm.median(dv).cpu(), O.median(rows)), (n, d, "median")
            assert close(bm.trmean(dv, f), O.trmean(rows, f), 1e-6), (n, d, "trmean")
            assert close(bm.phocas(dv, f), O.phocas(rows, f), 2e-6), (n, d, "phocas")
            assert close(bm.meamed(dv, f), O.meamed(rows, f), 2e-6), (n, d, "meamed")
            assert torch.equal(bm.krum(dv, f).cpu(), O.krum(rows, f)), (n, d, "krum")
            assert close(bm.bulyan(dv, f), O.bulyan(rows, f), 2e-6), (n, d, "bulyan")
            assert torch.equal(bm.aksel(dv, f).cpu(), O.aksel(rows, f)), (n, d, "aksel")
            assert torch.equal(bm.cge(dv, f).cpu(), O.cge(rows, f)), (n, d, "cge")
            if n <= 25:
                assert torch.equal(bm.brute(dv, f).cpu(), O.brute(rows, f)), (n, d, "brute")
            bm.gars.invalidate_rank_cache()
print("ok")
''')

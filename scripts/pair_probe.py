"""Probe of bm_pairwise_sqdist alone: accuracy against fp64 on the same GPU, exact-tie properties and
HIP-event timings, for the mode selected by the BM_PAIR_* environment (read once per process).

    python scripts/pair_probe.py [acc] [time]
"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm

TAG = " ".join(k + "=" + v for k, v in sorted(os.environ.items()) if k.startswith("BM_")) or "defaults"


def truth(rows):
  st = torch.stack([r.double() for r in rows])
  g = st @ st.T
  dg = g.diagonal()
  # direct fp64 differences (no cancellation): row by row to bound memory
  n = len(rows)
  out = torch.zeros((n, n), dtype=torch.float64, device=rows[0].device)
  for i in range(n):
    diff = st - st[i]
    out[i] = (diff * diff).sum(dim=1)
  return out


def rel_err(sq, want):
  mask = want > 0
  return float(((sq - want).abs()[mask] / want[mask]).max()) if bool(mask.any()) else 0.0


def accuracy():
  gen = torch.Generator(device="cuda").manual_seed(5)
  for n, d in ((2, 1000), (3, 64), (5, 4097), (13, 4099), (16, 70001), (17, 65), (25, 100003), (32, 50000),
               (33, 1234), (48, 9999), (49, 20000), (51, 100003), (64, 6007)):
    for kind in ("iid", "cluster", "outlier"):
      if kind == "iid":
        rows = [torch.randn(d, device="cuda", generator=gen) for _ in range(n)]
      elif kind == "cluster":  # large common component, tiny spread: the Gram cancellation case
        mu = 10.0 * torch.randn(d, device="cuda", generator=gen)
        rows = [mu + 0.01 * (1 + i / n) * torch.randn(d, device="cuda", generator=gen) for i in range(n)]
      else:  # tight cluster + far outliers (what centring on the mean cannot fix)
        mu = torch.randn(d, device="cuda", generator=gen)
        rows = [mu + 1e-3 * torch.randn(d, device="cuda", generator=gen) for i in range(n)]
        rows[-1] = -50.0 * mu
      if n >= 4:
        rows[n - 2] = rows[n - 3]  # aliased pair
      sq = bm.gars.pairwise_sqdist(rows)
      want = truth(rows)
      ok_sym = bool(torch.equal(sq, sq.T)) and bool((sq.diagonal() == 0).all())
      tie = True
      if n >= 4:
        tie = sq[n - 2, n - 3].item() == 0.0 and bool(torch.equal(sq[n - 2], sq[n - 3]))
      print(f"acc n={n:2d} d={d:6d} {kind:8s} maxrel {rel_err(sq, want):.2e} sym {ok_sym} ties {tie} [{TAG}]",
            flush=True)
  # unaligned rows + non-finite values
  n, d = 13, 4099
  flat = torch.randn(n * (d + 8), device="cuda", generator=gen)
  for off in (1, 2, 3):
    rows = [flat[i * (d + 8) + off: i * (d + 8) + off + d] for i in range(n)]
    print(f"acc unaligned off={off} maxrel {rel_err(bm.gars.pairwise_sqdist(rows), truth(rows)):.2e} [{TAG}]")
  rows = [torch.randn(d, device="cuda", generator=gen) for _ in range(n)]
  rows[3] = rows[3].clone()
  rows[3][17] = float("nan")
  rows[5] = rows[5].clone()
  rows[5][100] = float("inf")
  sq = bm.gars.pairwise_sqdist(rows)
  want = truth(rows)
  fin = torch.isfinite(want)
  good = bool((torch.isfinite(sq) == fin).all()) or bool((torch.isfinite(sq) | ~fin).all())
  print(f"acc nonfinite: finite pattern ok {good}; finite maxrel "
        f"{float(((sq - want).abs()[fin & (want > 0)] / want[fin & (want > 0)]).max()):.2e} "
        f"rows3/5 all nonfinite {bool((~torch.isfinite(sq[3, [0, 1, 2, 4]])).all())} [{TAG}]", flush=True)


def timing(ns=(25, 51, 16, 64)):
  d = 11173962
  for n in ns:
    gen = torch.Generator(device="cuda").manual_seed(n)
    stacks = [[torch.randn(d, device="cuda", generator=gen) for _ in range(n)] for _ in range(2)]
    for i in range(3):
      bm.gars.pairwise_sqdist(stacks[i & 1])
    torch.cuda.synchronize()
    evs = []
    for i in range(12):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      bm.gars.pairwise_sqdist(stacks[i & 1])
      b.record()
      evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    sq = bm.gars.pairwise_sqdist(stacks[0])
    errs = []
    for (i, j) in ((0, 1), (2, n - 1), (n // 2, n // 2 + 3)):
      want = (stacks[0][i].double() - stacks[0][j].double()).pow(2).sum().item()
      errs.append(abs(sq[i, j].item() - want) / want)
    algo = 4 * d * n
    print(f"time n={n} pairwise median {ts[6]*1e3:.0f} us best {ts[0]*1e3:.0f} us  {algo/ts[6]/1e6:.0f} GB/s "
          f"maxrelerr {max(errs):.1e} [{TAG}]", flush=True)
    del stacks


def near_duplicates():
  """Timing of the accuracy gate's slow path: f rows that are near-duplicates of each other (1e-3 relative
  jitter) inside an otherwise ordinary stack -> the direct kernel recomputes that sub-stack."""
  d = 11173962
  for n, f in ((25, 5), (51, 12)):
    gen = torch.Generator(device="cuda").manual_seed(n)
    rows = [torch.randn(d, device="cuda", generator=gen) for _ in range(n - f)]
    byz = -0.1 * torch.stack(rows[:4]).mean(dim=0)
    rows += [byz + 1e-3 * byz.abs().mean() * torch.randn(d, device="cuda", generator=gen) for _ in range(f)]
    for i in range(3):
      bm.gars.pairwise_sqdist(rows)
    torch.cuda.synchronize()
    evs = []
    for i in range(10):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      sq = bm.gars.pairwise_sqdist(rows)
      b.record()
      evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    i, j = n - 1, n - 2
    want = (rows[i].double() - rows[j].double()).pow(2).sum().item()
    print(f"dup n={n} f={f} pairwise median {ts[5]*1e3:.0f} us best {ts[0]*1e3:.0f} us; near-duplicate pair relerr "
          f"{abs(sq[i, j].item() - want) / want:.1e} [{TAG}]", flush=True)


if __name__ == "__main__":
  what = sys.argv[1:] or ["acc", "time"]
  if "acc" in what:
    accuracy()
  if "dup" in what:
    near_duplicates()
  for w in what:
    if w.startswith("time"):
      timing(tuple(int(v) for v in w.split(":")[1].split(",")) if ":" in w else (25, 51, 16, 64))

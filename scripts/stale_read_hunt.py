"""Hunt for the transient wrong result of a second-pass kernel under GPU sharing — and name the culprit.

  python scripts/stale_read_hunt.py --procs 4 --iters 150 --hold-gb 6      (the parent holds a context + 6 GB, like pytest)
  python scripts/stale_read_hunt.py --procs 4 --iters 150 --hold-gb 0      (the parent never touches the GPU)

Every child shares cuda:0 and repeats two scenarios of tests/test_gpu_zz_multirank.py at n = 25, f = 5, d = 200 003:

  R (rule)  n - f distinct rows are uploaded from pageable host memory INTO THE SAME device tensors every iteration (so the
            previous content of every address is known: the previous iteration's rows), then each consumer kind runs three
            times on the unchanged rows: distance pass (Gram kernel), selected mean (Krum's second pass), Bulyan pass 2,
            coordinate-wise median.  A kernel is deterministic: launches that differ bitwise are transient read errors.
            Which kind goes first rotates with the iteration.
  S (step)  one AggregationStep(bulyan) step (fused first pass -> reduce/rank -> pass 2 -> study block) on uploaded
            sampled gradients, then the rule again on the step's own buffers.

For every wrong coordinate of a Bulyan pass 2 the report solves for the single (ranked row, value) substitution that
reproduces the kernel's output, with the candidates "what that address held one / two iterations ago", and prints the
128-byte line the address falls in.  Output: one JSON line per child + one summary line.
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, F = 25, 5
MMAX, THETA, BETA = N - F - 2, N - 2 * F - 2, N - 4 * F - 2


def pass2_column(x):
  """Bulyan's second pass on ONE column given its MMAX ranked values (aggregators/bulyan.py:64-84, static scores): fp32,
  the kernel's summation order."""
  import numpy as np
  x = np.asarray(x, dtype=np.float32)
  sel = []
  for i in range(THETA):
    s = np.float32(0.0)
    for t in range(i, MMAX):
      s = np.float32(s + x[t])
    sel.append(np.float32(s / np.float32(MMAX - i)))
  sel = np.sort(np.asarray(sel, dtype=np.float32), kind="stable")
  med = sel[(THETA - 1) // 2]
  lo, hi = 0, THETA - 1
  for _ in range(THETA - BETA):
    if abs(np.float32(sel[lo] - med)) > abs(np.float32(sel[hi] - med)):
      lo += 1
    else:
      hi -= 1
  w = np.float32(0.0)
  for i in range(lo, hi + 1):
    w = np.float32(w + sel[i])
  return float(np.float32(w / np.float32(BETA)))


def culprits(torch, rows_now, order, wrong, got, history, limit=24):
  """For the first `limit` wrong coordinates: which single ranked row, replaced by what its address held k iterations
  ago, reproduces the kernel's output?  rows_now: device rows; history: [rows one iteration ago, two ago] as host lists."""
  import numpy as np
  idx = wrong.nonzero().flatten().tolist()
  found = []
  ranked = order[:MMAX].tolist()
  for j in idx[:limit]:
    now = [float(rows_now[r][j]) for r in ranked]
    target = float(got[j])
    clean = pass2_column(now)
    hit = None
    for age, old in enumerate(history, start=1):
      if old is None:
        continue
      past = [float(old[r][j]) if old[r] is not None else now[t] for t, r in enumerate(ranked)]
      groups = {}
      for t, r in enumerate(ranked):  # aliased rows are ONE address: substitute all their ranks together
        groups.setdefault(rows_now[r].data_ptr(), []).append(t)
      for ptr, ts in groups.items():
        trial = list(now)
        for t in ts:
          trial[t] = past[t]
        val = pass2_column(trial)
        if val == target or abs(val - target) <= 3e-7 * max(abs(target), 1e-30):
          hit = {"row": ranked[ts[0]], "ranks": ts, "age": age, "line": (ptr + 4 * j) >> 7, "word_in_line": ((ptr + 4 * j) >> 2) & 31}
          break
      if hit:
        break
      if hit is None:
        val = pass2_column(past)
        if val == target or abs(val - target) <= 3e-7 * max(abs(target), 1e-30):
          hit = {"row": "ALL", "age": age}
          break
    found.append({"col": j, "got": target, "recomputed_now": clean, "explained_by": hit})
  return found


def child(args):
  import torch
  torch.cuda.set_device(0)
  import byzantinemomentum_amd as bm
  from byzantinemomentum_amd import _lib, gars
  from byzantinemomentum_amd.sharded import ShardedAggregator
  from byzantinemomentum_amd.step import AggregationStep
  dev, d, h = "cuda:0", args.d, N - F
  gen = torch.Generator().manual_seed(1000 + args.rank)
  K = 4
  pool = []
  for k in range(K):
    base = 0.2 * torch.randn(d, generator=gen)
    hon = [base + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(h)]
    avg = torch.stack(hon).mean(dim=0)
    pool.append(hon + [avg * -0.1])  # an "empire"-like Byzantine vector, aliased F times
  devrows = [torch.empty(d, device=dev) for _ in range(h + 1)]
  rows = devrows[:h] + [devrows[h]] * F
  step = AggregationStep(N, F, F, gar="bulyan", momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2,
                         aggregator=ShardedAggregator(local_only=True))
  params, origin = torch.randn(d, generator=gen).to(dev), torch.randn(d, generator=gen).to(dev)
  searchers = {"mediansearch": AggregationStep(N, F, F, gar="median", attack_evals=16, nb_past=0),
               "akselsearch": AggregationStep(N, F, F, gar="aksel", attack_evals=16, nb_past=0),
               # ABI 23: every candidate ranked by one workgroup from the factor in device memory, pass 2 evaluate only
               "bulyansearch": AggregationStep(N, F, F, gar="bulyan", attack_evals=16, nb_past=0)}
  kinds = args.kinds.split(",")
  stats = {k: {"launch_sets": 0, "transient": 0, "first_launch_wrong": 0, "wrong_words": 0} for k in kinds + ["step"]}
  ranked_now = lambda rws, ordr, cols: [[float(rws[r][j]) for r in ordr[:MMAX].tolist()] for j in cols]
  reports = []
  m = MMAX
  t0 = time.time()
  for it in range(args.iters):
    for dst, src in zip(devrows, pool[it % K]):
      dst.copy_(src)  # H2D from pageable memory into the SAME addresses
    order, _ = gars._rank(rows, F, m, _lib.RANK_BULYAN)
    korder, _ = gars._rank(rows, F, m, _lib.RANK_KRUM)

    def run(kind):
      if kind == "gram":
        return gars.pairwise_sqdist(rows)
      if kind == "mean":
        return gars.selected_mean(rows, korder, m)
      if kind == "pass2":
        return gars.bulyan_pass2(rows, order, F, m)
      if kind in ("trmean", "phocas", "meamed"):
        return getattr(bm, kind)(rows, F)
      if kind in ("aksel", "cge", "brute", "krum", "bulyan"):   # whole rules, distance pass and ranking included
        gars.invalidate_rank_cache()
        return getattr(bm, kind)(rows, F)
      if kind == "search":   # the factor search on the device: distance pass over honests + [avg, avg + att], then the 16 candidates
        avg_, _, att = bm.stats.stack_stats_async(rows[:h], scale=1.0, attack="empire", direction=True)
        unit = torch.empty_like(avg_)
        bm.stats.multi_fma3([unit], [avg_], [att], 1.0, 1.0)
        return bm.stats.attack_search_device(gars.pairwise_sqdist(rows[:h] + [avg_, unit]), h, F, F, "krum", evals=16)
      if kind in ("mediansearch", "akselsearch", "bulyansearch"):
        # ABI 22: the median's search (bm_order_pair once, then 16 x [cursor, bm_colwise_eval on (lo, hi, candidate)]), resp. a
        # search in the generic form (candidate written, rule, bm_sqdist2) — factor and all 16 (abscissa, objective) pairs
        avg_, _, att = bm.stats.stack_stats_async(rows[:h], scale=1.0, attack="empire", direction=True)
        searcher = searchers[kind]
        found = searcher._search_factor(rows[:h], avg_, att)
        return found.clone() if isinstance(found, torch.Tensor) else torch.tensor([found])
      if kind == "stats":
        avg, norm, dev_, mx = bm.compute_avg_dev_max(rows[:h])
        return torch.cat([avg, torch.tensor([norm, dev_, mx], device=dev)])
      return bm.median(rows)
    for q in range(len(kinds)):
      kind = kinds[(q + it) % len(kinds)]
      outs = [run(kind) for _ in range(3)]
      st = stats[kind]
      st["launch_sets"] += 1
      same01, same12, same02 = (bool(torch.equal(outs[a], outs[b])) for a, b in ((0, 1), (1, 2), (0, 2)))
      if same01 and same12:
        continue
      st["transient"] += 1
      odd = 0 if same12 else (1 if same02 else (2 if same01 else -1))
      st["first_launch_wrong"] += int(odd == 0)
      rep = {"rank": args.rank, "it": it, "scenario": "R", "kind": kind, "odd_launch": odd, "position": q}
      if odd >= 0:
        good = outs[(odd + 1) % 3]
        wrong = (outs[odd] != good) & ~(outs[odd].isnan() & good.isnan())
        wrong = wrong.flatten()
        st["wrong_words"] += int(wrong.sum())
        wid = wrong.nonzero().flatten()
        rep.update(n_wrong=int(wrong.sum()), first=int(wid[0]), last=int(wid[-1]), cols=wid[:40].tolist())
        if kind == "pass2":
          history = [pool[(it - 1) % K] + [pool[(it - 1) % K][h]] * (F - 1) if it >= 1 else None,
                     pool[(it - 2) % K] + [pool[(it - 2) % K][h]] * (F - 1) if it >= 2 else None]
          rep["order"] = order[:MMAX].tolist()
          rep["culprits"] = culprits(torch, rows, order, wrong, outs[odd], history, limit=4)
          rep["inputs"] = ranked_now(rows, order, wid[:40].tolist())
          rep["got"] = [float(outs[odd][j]) for j in wid[:40].tolist()]
          rep["good"] = [float(good[j]) for j in wid[:40].tolist()]
          rep["row_ptr_mod_128"] = sorted({r.data_ptr() % 128 for r in devrows})
      reports.append(rep)
    # ---- scenario S: one step, then the rule again on the step's own buffers ----
    if args.steps:
      sampled = [g.to(dev) for g in pool[(it + 1) % K][:h]]
      defense = step.run(sampled, params, origin)
      step.floats()
      srows = list(step.buffers) + [step.last_byzantine] * F
      gars.invalidate_rank_cache()
      again = bm.bulyan(srows, F)
      st = stats["step"]
      st["launch_sets"] += 1
      if not torch.equal(defense, again):
        third = bm.bulyan(srows, F)
        wrong = (defense != again)
        st["transient"] += 1
        st["first_launch_wrong"] += int(bool(torch.equal(again, third)))
        st["wrong_words"] += int(wrong.sum())
        wid = wrong.nonzero().flatten()
        sorder = torch.tensor(gars.bulyan_ranking(srows, F), device=dev)
        rep = {"rank": args.rank, "it": it, "scenario": "S", "n_wrong": int(wrong.sum()), "first": int(wid[0]), "last": int(wid[-1]),
               "cols": wid[:40].tolist(), "again_equals_third": bool(torch.equal(again, third)), "order": sorder[:MMAX].tolist(),
               "max_abs": float((defense - again).abs().max())}
        rep["inputs"] = ranked_now(srows, sorder, wid[:40].tolist())
        rep["got"] = [float(defense[j]) for j in wid[:40].tolist()]
        rep["good"] = [float(again[j]) for j in wid[:40].tolist()]
        reports.append(rep)
  torch.cuda.synchronize()
  print(json.dumps({"rank": args.rank, "seconds": round(time.time() - t0, 1), "stats": stats, "reports": reports[:8],
                    "n_reports": len(reports)}), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--procs", type=int, default=4)
  ap.add_argument("--iters", type=int, default=100)
  ap.add_argument("--d", type=int, default=200003)
  ap.add_argument("--hold-gb", type=float, default=6.0)
  ap.add_argument("--steps", type=int, default=1)
  ap.add_argument("--kinds", default="gram,mean,pass2,median")
  ap.add_argument("--lib", default="", help="another build of libbm_gar.so for the children (BM_GAR_LIB)")
  ap.add_argument("--rank", type=int, default=-1)
  args = ap.parse_args()
  if args.rank >= 0:
    return child(args)
  held = None
  if args.hold_gb > 0:  # what the pytest parent is when the multi-rank file runs behind the rest of the suite
    import torch
    held = [torch.full((1 << 28,), 1.0, device="cuda:0") for _ in range(int(args.hold_gb))]
    side = [torch.cuda.Stream() for _ in range(3)]
    for s in side:
      with torch.cuda.stream(s):
        held[0][:1024].add_(1.0)
    torch.cuda.synchronize()
  cmd = [sys.executable, os.path.abspath(__file__), "--iters", str(args.iters), "--d", str(args.d), "--steps", str(args.steps),
         "--kinds", args.kinds]
  env = dict(os.environ)
  if args.lib:
    env["BM_GAR_LIB"] = os.path.abspath(args.lib)
  procs = [subprocess.Popen(cmd + ["--rank", str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
           for r in range(args.procs)]
  total = {}
  for p in procs:
    out, err = p.communicate()
    sys.stdout.write(out)
    if p.returncode != 0:
      sys.stdout.write(json.dumps({"child_failed": p.returncode, "stderr": err[-2000:]}) + "\n")
    for line in out.splitlines():
      try:
        rec = json.loads(line)
      except ValueError:
        continue
      for kind, st in rec.get("stats", {}).items():
        agg = total.setdefault(kind, {})
        for key, val in st.items():
          agg[key] = agg.get(key, 0) + val
  print(json.dumps({"summary": total, "procs": args.procs, "iters": args.iters, "parent_holds_gb": args.hold_gb, "lib": args.lib or "default"}))


if __name__ == "__main__":
  main()

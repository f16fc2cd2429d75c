"""One aggregation step of the simulation loop on the HIP path (mirror of attack.py:757-878).

Given the sampled honest gradients of a step it performs, without leaving the GPU:
  0. gradient clipping             g_i <- g_i * clip/||g_i|| where ||g_i|| > clip   attack.py:776-779,791-794
  1. momentum, by placement        worker: buf_i <- mu*buf_i + (1-damp)*g_i            attack.py:800-804
                                   server: hon_i  = (1-damp)*g_i + mu*M                attack.py:805-808
                                   update / none: hon_i = g_i                          attack.py:809-810
  2. the "empire" / "little" attack  byz = avg_h + factor*dir, repeated f_real times   attacks/identical.py:63-86,129-141
                                   factor fixed, or searched within `attack_evals` evaluations of
                                   |GAR(honests + [avg_h + t*dir]*f) - avg_h|^2         attacks/identical.py:67-77
  3. the aggregation rule          defense = GAR(honests + [byz]*f, f)                 attack.py:821
  4. the momentum of the update    server: M <- defense; update: M <- mu*M + (1-damp)*defense   attack.py:832-839
  5. the study statistics          sampled / honest / attack stacks, defense norm and max, six cosines,
                                   previous-step cosine and curvature, l2 distance from the origin   attack.py:828-868
With worker momentum, steps 0-2 and the sampled/honest statistics of step 5 are ONE kernel
(bm_momentum_stats): every sampled gradient and every momentum buffer is read once.

The step shards along the coordinates like the rules do (sharded.ShardedAggregator): every rank
passes its slice of every tensor; scalars are exchanged in at most three small collectives per
step (row norms if clipping, the n x n squared distances if the rule needs them, and one packed
vector of every statistic), never a d-sized one.  With one rank no collective is issued.
The model update itself (optimizer step) belongs to the caller: `update_gradient()` returns what
attack.py hands to `model.update`.  Everything is asynchronous on the current stream until
`floats()` fetches the scalars (one sync) — except for the factor search, sequential by nature: one
synchronisation per evaluation, or ONE for the whole search when the rule is krum, brute or average
(linesearch.py: every evaluation is then a function of (h+2)^2 scalars of one distance pass).
"""

import collections
import math

import torch

__all__ = ["AggregationStep"]

_RULES = ("krum", "bulyan", "median", "trmean", "phocas", "meamed", "aksel", "brute", "average", "cge")
_NEEDS_F = {"krum", "bulyan", "trmean", "phocas", "meamed", "aksel", "brute", "cge"}
MAX_PAST = 4096  # past sampled averages kept for the curvature term (each is one d-vector of device memory)


class AggregationStep:
  def __init__(self, nb_workers, nb_decl_byz, nb_real_byz, gar="krum", gar_args=None, momentum=0.99,
               dampening=0.99, momentum_at="worker", attack="empire", attack_factor=1.1, nb_past=25,
               gradient_clip=None, aggregator=None, single_call=True, attack_evals=None, attack_negative=False,
               line_search="auto"):
    """aggregator: a sharded.ShardedAggregator (default: one over the default process group, or a
    single-rank one when torch.distributed is not initialised).
    single_call: with the HIP backend, worker-side momentum and a rule the C entry point knows
    (krum, bulyan, median, trmean, phocas, meamed), run() is ONE call into libbm_gar.so (bm_step_worker),
    collectives included; False keeps the kernel-by-kernel Python sequence (same kernels, same results).
    attack_evals: None = the fixed `attack_factor`; a positive integer E = the reference's `factor:-E` (its
    default is -16): the factor is searched each step with tools.line_maximize's exploration
    (identical.py:67-77), `attack_negative` being the attack's `negative` argument during the search.
    line_search: "auto" evaluates the search from scalars when the rule allows it (krum, brute, average) — on the
    device for krum / average (bm_attack_line_search_device) — and otherwise keeps the exploration's cursor in device
    memory (bm_search_device_next: median, trmean, phocas, meamed, aksel, cge ...): the step then has no synchronisation
    but `floats()`; Bulyan (its candidates are ranked on the host) and Brute keep the host's cursor.  "host": the same
    forms with scalars and cursor on the host (one synchronisation per evaluation).  "generic" always runs the rule on
    the device once per evaluation like the reference does, the cursor on the host."""
    if gar not in _RULES:
      raise ValueError(f"unknown aggregation rule {gar!r}")
    if momentum_at not in ("worker", "server", "update"):
      raise ValueError(f"momentum_at must be 'worker', 'server' or 'update', got {momentum_at!r}")
    if attack not in ("empire", "little"):
      raise ValueError(f"unknown attack {attack!r} (empire: factor, little: factor, use a negative one for negative:True)")
    if attack_evals is not None and (not isinstance(attack_evals, int) or attack_evals < 1):
      raise ValueError(f"attack_evals must be a positive number of evaluations, got {attack_evals!r}")
    if line_search not in ("auto", "host", "generic"):
      raise ValueError(f"line_search must be 'auto', 'host' or 'generic', got {line_search!r}")
    if not 0 <= nb_past <= MAX_PAST:
      raise ValueError(f"nb_past must be within 0..{MAX_PAST}")
    if aggregator is None:
      from .sharded import ShardedAggregator
      aggregator = ShardedAggregator()
    self.agg = aggregator
    self.ops = aggregator.backend
    self.n = nb_workers
    self.f_decl = nb_decl_byz
    self.f_real = nb_real_byz
    self.h = nb_workers - nb_real_byz
    self.gar = gar
    self.gar_args = dict(gar_args or {})
    self.mu = momentum
    self.damp = dampening
    self.momentum_at = momentum_at
    self.attack = attack
    self.factor = attack_factor
    self.clip = gradient_clip
    self.attack_evals = attack_evals
    self.attack_negative = bool(attack_negative)
    self.line_search = line_search
    self._factor_now = attack_factor  # last_factor / last_search (below): a number and a list, or what the device
    self._search_now = None           # search left in device memory (fetched when somebody asks)
    self.last_byzantine = None        # the Byzantine vector of the last step (aliased f_real times by the rule)
    self.buffers = None        # worker placement: one momentum buffer per honest worker (attack.py:676)
    self.server_momentum = None  # server / update placements: grad_momentum_server (attack.py:678)
    self.pasts = collections.deque(maxlen=max(nb_past, 1))  # past sampled averages, newest first (attack.py:868)
    self.nb_past = nb_past
    self._prev_s2 = None       # device fp64[1]: (this rank's part of) the squared norm of pasts[0]
    # C = sum_i mu^i * pasts[i]: the curvature term mu * sum_i mu^i <s, pasts[i]> (attack.py:863-865) is
    # mu * <s, C>, ONE dot product instead of nb_past of them; C is updated in place each step
    # (C <- s + mu * (C - mu^(P-1) * oldest) once the deque is full), 6 row passes instead of nb_past
    self._curv = None
    self._pending = None
    self._update = None
    self._prev_stats = None    # single-call form: the previous step's reduced statistics (slot 0 = ||avg_s||^2)
    extra_args = set(self.gar_args) - {"m"}
    self.single_call = bool(single_call and attack_evals is None and momentum_at == "worker"
                            and hasattr(self.ops, "step_worker")
                            and gar in ("krum", "bulyan", "median", "trmean", "phocas", "meamed") and not extra_args
                            and (not self.agg.collective or self.agg.native is not None))

  # ------------------------------------------------------------------------ #

  @property
  def last_factor(self):
    """The factor of the last step (the searched one with attack_evals).  After a device search this is the read that
    synchronises; a step that never looks costs nothing."""
    if isinstance(self._factor_now, torch.Tensor):
      self._settle_search()
    return self._factor_now

  @last_factor.setter
  def last_factor(self, value):
    self._factor_now = value

  @property
  def last_search(self):
    """[(x, objective)] of the last search, in evaluation order."""
    if isinstance(self._search_now, torch.Tensor):
      self._settle_search()
    return self._search_now

  @last_search.setter
  def last_search(self, value):
    self._search_now = value

  def _settle_search(self):
    values = self._fetch(self._search_now).tolist()
    self._factor_now = values[0]
    self._search_now = [(values[1 + 2 * i], values[2 + 2 * i]) for i in range((len(values) - 1) // 2)]

  @staticmethod
  def _new_rows(count, like, zero=False):
    """`count` vectors shaped like `like`, one allocation each.  (Rows of one allocation at a skewed stride,
    layout.alloc_rows, help the column kernels; for the write-heavy first pass of the step they measured 0.3-5 %
    slower than separate allocations in every A/B, profiles/r03_h / r03_i, so the step keeps separate ones.)"""
    return [torch.zeros_like(like) if zero else torch.empty_like(like) for _ in range(count)]

  def _aggregate(self, gradients):
    agg, f = self.agg, self.f_decl
    if self.gar == "median":
      return agg.median(gradients)
    if self.gar == "average":
      return agg.average(gradients)
    if self.gar == "brute":
      # checked HERE, before anybody can apply the defense vector (one 4-byte synchronisation per step for this rule
      # alone): no admissible subset raises like brute.py:68, a device search that ran out of its node budget is
      # repeated on the host
      return agg.brute(gradients, f, check=True, **self.gar_args)
    return getattr(agg, self.gar)(gradients, f, **self.gar_args)

  def _fetch(self, matrix):
    """A small device matrix on the host, through a pinned buffer this step keeps (one asynchronous copy + one stream
    synchronisation, no allocation per search).  Measured next to `.cpu()` into pageable memory: 0.019 against 0.017 ms
    for the 15 KB of a C3 search (bench.py, `attack_search_c3_krum.legs`) — the copy is not where a slow search loses
    its time (that was a stalled host, profiles/r06_search_each.txt); the buffer only keeps the path free of the
    runtime's staging allocations."""
    if not matrix.is_cuda:
      return matrix.contiguous()
    key = (tuple(matrix.shape), matrix.dtype)
    if getattr(self, "_pinned", None) is None or self._pinned[0] != key:
      self._pinned = (key, torch.empty(matrix.shape, dtype=matrix.dtype, pin_memory=True))
    host = self._pinned[1]
    host.copy_(matrix, non_blocking=True)
    torch.cuda.current_stream(matrix.device).synchronize()
    return host

  def _search_factor(self, honests, h_avg, direction):
    """attacks/identical.py:67-77: the factor maximising |GAR(honests + [avg + t*dir]*f_real) - avg|^2 under
    the evaluation budget.  Like the reference, `negative` flips the sign of the candidates DURING the
    search only; the factor returned (and then applied) is the positive abscissa the search settled on."""
    from . import linesearch
    ops, agg, h, k = self.ops, self.agg, self.h, self.f_real
    if self.line_search in ("auto", "host") and self.gar in linesearch.ANALYTIC_RULES and h + 2 <= 64 and \
       not (set(self.gar_args) - {"m"}):
      unit = torch.empty_like(h_avg)
      ops.multi_fma3([unit], [h_avg], [direction], 1.0, 1.0)   # avg + dir: the candidate of factor 1
      sq = agg.global_sqdist(list(honests) + [h_avg, unit])
      if self.line_search == "auto" and self.gar in getattr(ops, "device_search_rules", ()):
        # every candidate evaluated where the distances are: no copy, no synchronisation; the factor stays on the
        # device (a tensor: multi_fma3 reads it there) and last_factor / last_search fetch it when asked
        found = ops.attack_search_device(sq, h, k, self.f_decl, self.gar, evals=self.attack_evals,
                                         negative=self.attack_negative, m=self.gar_args.get("m"))
        self.last_search = found
        return found
      ext = self._fetch(sq)  # the search's only synchronisation
      factor, self.last_search = linesearch.attack_line_search(
        ext, h, k, self.f_decl, self.gar, evals=self.attack_evals, negative=self.attack_negative,
        m=self.gar_args.get("m"))
      return factor

    rule = lambda cand, t: self._aggregate(list(honests) + [cand] * k)  # noqa: E731
    n = h + k
    bulyan_objective = None
    host_ranked = self.gar == "brute"  # (its checked call reads a status: a synchronisation per evaluation anyway)
    if self.line_search in ("auto", "host") and self.gar == "bulyan" and k >= 1 and h + 2 <= 64 and hasattr(ops, "bulyan_pass2") \
       and not (set(self.gar_args) - {"m"}):
      # Bulyan's second pass needs the vectors, its ranking does not: the distances among honests + [avg + t*dir] * k
      # are functions of the inner products of ONE distance pass over honests + [avg, avg + dir] (as for krum above),
      # so every candidate is ranked on the host and costs pass 2 alone (m + 1 row passes instead of n + m + 1)
      m = self.gar_args.get("m") or n - self.f_decl - 2
      unit = torch.empty_like(h_avg)
      ops.multi_fma3([unit], [h_avg], [direction], 1.0, 1.0)
      sq_dev = agg.global_sqdist(list(honests) + [h_avg, unit])
      # ranked where the matrix is, from the factor the device cursor left there (bm_attack_ranking_device: no copy, no
      # synchronisation) — or on the host from the number the host's cursor proposed (one copy of the matrix per search)
      device_ranked = self.line_search == "auto" and hasattr(ops, "attack_ranking_device") and sq_dev.is_cuda
      ext = None if device_ranked else self._fetch(sq_dev)
      host_ranked = not device_ranked

      def ranking(t):
        if isinstance(t, torch.Tensor):
          return ops.attack_ranking_device(sq_dev, h, k, self.f_decl, "bulyan", t, m)
        order = linesearch.attack_ranking(ext, h, k, self.f_decl, "bulyan", t, m)
        return ops.index_tensor(order + [0] * (64 - n), h_avg)

      def rule(cand, t):  # noqa: F811
        return ops.bulyan_pass2(list(honests) + [cand] * k, ranking(t), self.f_decl, m)

      if hasattr(ops, "bulyan_pass2_eval") and ops.bulyan_pass2_eval_supported(n, self.f_decl, m, h_avg.shape[0]):
        # ... and pass 2 has an evaluate-only form for the shapes of the reference's experiments: the candidate in
        # registers, the objective accumulated in the same kernel, nothing written (bm_bulyan_pass2_eval)
        def bulyan_objective(t):
          return ops.bulyan_pass2_eval(honests, k, ranking(t), self.f_decl, m, h_avg, direction, t)
    if self.line_search in ("auto", "host") and self.gar == "median" and k >= 1:
      # The lower median of the h honest values and k copies of ONE value b is monotone in b, equals b while b lies
      # between two order statistics of the honest values and stays at them outside: median(honests + [b] * k) =
      # middle of (b, lo, hi) per coordinate, with lo / hi the medians of the honest values and k copies of -inf /
      # +inf.  Two passes over the honest rows for the whole search, then every candidate is the median of THREE
      # rows (4 row passes instead of n + 1): the same value of the rule at every coordinate — it returns one of its
      # inputs, no arithmetic — hence the same objective, bit for bit, as evaluating the rule on the n rows.
      if hasattr(ops, "order_pair") and ops.order_pair_supported(h):
        lo, hi = ops.order_pair(honests, (n - 1) // 2 - k, (n - 1) // 2)   # both in one pass over the honest rows
      else:
        lo = agg.median(list(honests) + [torch.full_like(h_avg, -math.inf)] * k)
        hi = agg.median(list(honests) + [torch.full_like(h_avg, math.inf)] * k)
      rule = lambda cand, t: agg.median([cand, lo, hi])  # noqa: E731

    fused_eval = (self.line_search in ("auto", "host") and k >= 1 and not self.gar_args and hasattr(ops, "colwise_eval")
                  and ops.colwise_eval_supported(self.gar, n))
    eval_rows, eval_copies, eval_f = honests, k, self.f_decl
    if (self.line_search in ("auto", "host") and self.gar == "median" and k >= 1 and hasattr(ops, "colwise_eval")
        and ops.colwise_eval_supported("median", 3)):
      # ... and the middle of (lo, hi, candidate) has an evaluate-only instance: 4 row passes, nothing written
      fused_eval, eval_rows, eval_copies, eval_f = True, [lo, hi], 1, 0

    def evaluate(t):
      """The objective of candidate t as a device fp64[1] tensor; t a number or the device cursor's tensor."""
      if bulyan_objective is not None:
        sq = bulyan_objective(t)
      elif fused_eval:
        # trmean / phocas / meamed: candidate, rule and objective in ONE pass over the honest rows, nothing written
        # (bm_colwise_eval: h + 2 row passes instead of h + 5 read and 2 written); the same value at every column
        sq = ops.colwise_eval(self.gar, eval_rows, eval_copies, eval_f, h_avg, direction, t)
      else:
        cand = torch.empty_like(h_avg)
        ops.multi_fma3([cand], [h_avg], [direction], 1.0, t)
        out = rule(cand, t)
        # aggregated.sub_(grad_avg); dot with itself: one pass over the two vectors where the backend has it
        sq = ops.sqdist2(out, h_avg) if hasattr(ops, "sqdist2") else ops.pairwise_sqdist([out, h_avg])[0, 1].reshape(1)
      agg.all_reduce_sum(sq)
      return sq

    if self.line_search == "auto" and not host_ranked and hasattr(ops, "device_search") and h_avg.is_cuda:
      # the cursor of the exploration in device memory: every evaluation reads its factor there and leaves its objective
      # there — the host queues the whole search and waits for none of it (Bulyan's candidates are ranked by one
      # workgroup from that factor; Brute's checked call synchronises by itself and keeps the host's cursor)
      cursor = ops.device_search(h_avg.device, self.attack_evals, self.attack_negative)
      y = None
      for _ in range(self.attack_evals):
        y = evaluate(cursor.next(y))
      found = cursor.finish(y)
      self.last_search = found
      return found

    def scape(x):
      return evaluate(-x if self.attack_negative else x).item()

    factor, self.last_search = linesearch.line_maximize(scape, evals=self.attack_evals)
    return factor

  def nesterov_lookahead(self, params, lr, worker=None):
    """params <- params - mu*lr*momentum in place (attack.py:760-767): the parameter shift before
    the gradients of a Nesterov step are computed.  worker: index of the worker momentum buffer
    (worker placement), else the server momentum."""
    # (before the first step every momentum is zero, attack.py:676-678: the shift is then the identity)
    mom = (self.buffers[worker] if self.buffers is not None else None) if worker is not None else self.server_momentum
    if mom is not None:
      self.ops.multi_fma3([params], [params], [mom], 1.0, -(self.mu * lr))
    return params

  def run(self, grad_sampleds, params=None, origin=None):
    """grad_sampleds: list of >= h flat fp32 GPU tensors (this rank's slice of the step's sampled
    gradients).  params/origin: optional flat parameter vectors for `l2_origin` (attack.py:830).
    Returns the aggregated gradient (slice); statistics stay on the device until floats()."""
    ops, agg, h = self.ops, self.agg, self.h
    sampled = list(grad_sampleds)
    if getattr(grad_sampleds, "d_total", None) is not None:  # sharded.Shards: keep the stated total length
      from .sharded import Shards
      sampled = Shards(sampled, d_total=grad_sampleds.d_total)
    ks = len(sampled)
    if ks < h:
      raise ValueError(f"{ks} sampled gradients for {h} honest workers")
    omd = 1.0 - self.damp
    if self.single_call:
      return self._run_single_call(sampled, ks, omd, params, origin)
    # 0. clipping factors (device scalars; the all-reduce makes them global under sharding)
    factors = None
    if self.clip is not None:
      sq = ops.row_sqnorms(sampled)
      agg.all_reduce_sum(sq)
      factors = ops.clip_factors_from_sq(sq, ks, self.clip)
    # 1.+2. momentum, attack vector, sampled/honest statistics
    fused_defense, fused_sq = None, None
    if self.momentum_at == "worker":
      if self.buffers is None:
        self.buffers = self._new_rows(h, sampled[0], zero=True)
      fused_rule = (self.attack_evals is None and self.f_real >= 1 and not self.gar_args
                    and self.gar in ("median", "trmean", "phocas", "meamed") and hasattr(ops, "momentum_stats_colwise"))
      if fused_rule:  # first pass + coordinate-wise rule in one call (one kernel for median / trmean at h = 20)
        s_avg, h_avg, byz, fused_defense, out6 = ops.momentum_stats_colwise(
          sampled, self.buffers, self.mu, omd, factors, self.factor, self.attack, self.gar, self.f_decl, self.f_real)
      elif (self.attack_evals is None and self.f_real >= 1 and self.gar in ("krum", "bulyan")
            and not (set(self.gar_args) - {"m"}) and hasattr(ops, "momentum_stats_sqdist")):
        # first pass + the distance pass of the rule in one call (one kernel at h = 20 for long gradients)
        s_avg, h_avg, byz, fused_sq, out6 = ops.momentum_stats_sqdist(
          sampled, self.buffers, self.mu, omd, factors, self.factor, self.attack, self.f_real,
          d_total=agg._total_of(sampled))
      elif self.attack_evals is None:
        s_avg, h_avg, byz, out6 = ops.momentum_stats(sampled, self.buffers, self.mu, omd, factors, self.factor, self.attack)
      else:  # the attack direction alone; the Byzantine vector follows the factor search
        s_avg, h_avg, byz, out6 = ops.momentum_stats(sampled, self.buffers, self.mu, omd, factors, 1.0, self.attack,
                                                     direction=True)
      honests = self.buffers
      s_out3, h_out3 = out6[:3], out6[3:]
    else:
      if factors is not None:
        ops.multi_scale(sampled, factors)  # in place, like the reference's grad.mul_
      if self.momentum_at == "server" and self.server_momentum is not None:
        honests = self._new_rows(h, sampled[0])
        ops.multi_fma3(honests, sampled[:h], [self.server_momentum] * h, omd, self.mu)
      elif self.momentum_at == "server":
        # first step: grad_momentum_server is zero (attack.py:678), hon_i = (1-damp)*g_i
        honests = self._new_rows(h, sampled[0])
        zero = torch.zeros_like(sampled[0])
        ops.multi_fma3(honests, sampled[:h], [zero] * h, omd, self.mu)
      else:
        honests = sampled[:h]
      # momentum at the update with every sampled gradient honest: the rule, or its distance pass, is fed from the pass
      # that forms the statistics and the Byzantine vector (one pass over the rows at h = 20 / 14)
      plain_update = (self.momentum_at == "update" and ks == h and self.attack_evals is None and self.f_real >= 1)
      if plain_update and not self.gar_args and self.gar in ("median", "trmean", "phocas", "meamed") \
         and hasattr(ops, "stack_stats_colwise"):
        h_avg, byz, fused_defense, o6 = ops.stack_stats_colwise(honests, self.factor, self.attack, self.gar, self.f_decl,
                                                                self.f_real)
        h_out3 = o6[3:]
      elif plain_update and self.gar in ("krum", "bulyan") and not (set(self.gar_args) - {"m"}) \
          and hasattr(ops, "stack_stats_sqdist"):
        h_avg, byz, fused_sq, o6 = ops.stack_stats_sqdist(honests, self.factor, self.attack, self.f_real,
                                                          d_total=agg._total_of(sampled))
        h_out3 = o6[3:]
      elif self.attack_evals is None:
        h_avg, h_out3, byz = ops.stack_stats(honests, scale=self.factor, attack=self.attack)
      else:
        h_avg, h_out3, byz = ops.stack_stats(honests, scale=1.0, attack=self.attack, direction=True)
      if self.momentum_at == "update" and ks == h:
        # the honest stack IS the sampled stack (attack.py:809-810): one pass gives both sets of statistics
        s_avg, s_out3 = h_avg, h_out3
      else:
        s_avg, s_out3 = ops.stack_stats(sampled)
    if self.attack_evals is not None and self.f_real > 0:
      direction = byz
      factor = self._search_factor(honests, h_avg, direction)  # a number, or the device search's tensor ([0]: the factor)
      self.last_factor = factor
      byz = torch.empty_like(h_avg)  # grad_att.mul_(factor); byz_grad = grad_avg.add_(grad_att)  (identical.py:82-84)
      ops.multi_fma3([byz], [h_avg], [direction], 1.0, factor)
    attacks = [byz] * self.f_real
    self.last_byzantine = byz if self.f_real > 0 else None  # the Byzantine vector of this step (callers, tests)
    # 3. aggregation
    if fused_defense is not None:
      defense = fused_defense
    elif fused_sq is not None:
      defense = agg.rule_from_sq(self.gar, honests + attacks, fused_sq, self.f_decl, self.gar_args.get("m"))
    else:
      defense = self._aggregate(honests + attacks)
    # 4. momentum of the update
    if self.momentum_at == "server":
      self.server_momentum = defense          # no clone, as attack.py:835
      self._update = defense
    elif self.momentum_at == "update":
      # M <- mu * M + (1 - damp) * defense (attack.py:836-838) rides along with the study block below, which reads the
      # defense vector anyway (bm_study_stats_update): no pass of its own
      if self.server_momentum is None:
        self.server_momentum = torch.zeros_like(defense)
      self._update = self.server_momentum
    else:
      self._update = defense
    # 5. remaining statistics, ONE pass over the vectors (bm_study_stats): attack stack, defense vector, the Gram
    #    matrix behind the cosines, the dots with the past, l2 from the origin, and the curvature combination
    #    for the next step.  grad_pasts.appendleft(PastGrad(sampled_grad_avg, sampled_norm_avg)) (attack.py:868)
    #    happens every step, whether or not the scalars are fetched.
    count = len(self.pasts) if self.nb_past > 0 else 0
    mode = 0
    if self.nb_past > 0:
      if self._curv is None:
        self._curv = torch.empty_like(s_avg)  # written by the first step (C <- s)
      mode = 1 if count == 0 else (3 if count == self.nb_past else 2)  # 3: the oldest entry leaves the deque
    has_l2 = params is not None and origin is not None
    study = ops.study_stats(s_avg, h_avg, defense, byz if self.f_real > 0 else None, self.f_real,
                            past_newest=self.pasts[0] if count > 0 else None, curv=self._curv,
                            past_oldest=self.pasts[-1] if mode == 3 else None, curv_mode=mode, mu=self.mu,
                            oldest_weight=-(self.mu ** (self.nb_past - 1)) if self.nb_past > 0 else 0.0,
                            params=params if has_l2 else None, origin=origin if has_l2 else None,
                            **(dict(update_momentum=self.server_momentum, update_mu=self.mu, update_omd=omd)
                               if self.momentum_at == "update" else {}))
    self._pending = dict(s=s_out3, h=h_out3, study=study, npast=2 if count > 0 else 0,
                         prev_s2=self._prev_s2 if count > 0 else None, has_l2=has_l2, ks=ks, floats=None)
    if self.nb_past > 0:
      self.pasts.appendleft(s_avg)
      self._prev_s2 = s_out3[:1]
    return defense

  def _run_single_call(self, sampled, ks, omd, params, origin):
    h = self.h
    if self.buffers is None:
      self.buffers = self._new_rows(h, sampled[0], zero=True)
    count = len(self.pasts) if self.nb_past > 0 else 0
    if self.nb_past > 0 and self._curv is None:
      self._curv = torch.empty_like(sampled[0])  # written by the first step (C <- s)
    full = count == self.nb_past and count > 0
    defense, s_avg, h_avg, byz, stats = self.ops.step_worker(
      self.agg.native, sampled, self.buffers, self.n, self.f_decl, self.f_real, self.gar, self.gar_args.get("m"),
      self.mu, omd, self.clip, self.attack, self.factor, self.nb_past, count,
      self.pasts[0] if count > 0 else None, self._curv, self.pasts[-1] if full else None, params, origin,
      d_total=self.agg._total_of(sampled))
    self._update = defense
    self.last_byzantine = byz
    self._pending = dict(packed=stats, prev=self._prev_stats if count > 0 else None, npast=2 if count > 0 else 0,
                         has_attack=self.f_real > 0, has_l2=params is not None and origin is not None, ks=ks,
                         floats=None)
    if self.nb_past > 0:
      self.pasts.appendleft(s_avg)
      self._prev_stats = stats
    return defense

  def update_gradient(self):
    """What attack.py:832-839 passes to model.update(): the defense gradient (worker / server
    placements) or the updated server momentum (update placement)."""
    return self._update

  # ------------------------------------------------------------------------ #

  def _exchange(self, pend):
    """One packed exchange of every scalar of the step: sums and maxima, all ranks."""
    st = pend["study"]  # layout: include/bm_gar.h (bm_study_stats)
    sums = [pend["s"][:2], pend["h"][:2], st[:20], st[22:23]]
    maxes = [pend["s"][2:], pend["h"][2:], st[20:22]]
    if pend["prev_s2"] is not None:
      sums.append(pend["prev_s2"])
    # ONE copy to the host, ONE synchronisation: sums and maxima leave the device together (two `tolist()` were two
    # copies with a host round trip between them: ~30 us of a 0.95 ms step, profiles/r05_c_full_kernel_trace.csv), and
    # the whole tensors travel — three or four of them in one concatenation — to be taken apart on the host (eight
    # slices were two batched-copy launches)
    if not self.agg.collective:
      parts = [pend["s"], pend["h"], st] + ([pend["prev_s2"]] if pend["prev_s2"] is not None else [])
      flat = torch.cat(parts).tolist()
      s3, h3, stl = flat[0:3], flat[3:6], flat[6:6 + int(st.numel())]
      host_sums = s3[:2] + h3[:2] + stl[:20] + stl[22:23] + (flat[6 + int(st.numel()):] if pend["prev_s2"] is not None else [])
      return host_sums, s3[2:] + h3[2:] + stl[20:22]
    sums, maxes = self.agg.exchange(torch.cat(sums), torch.cat(maxes))
    ns = int(sums.numel())
    flat = torch.cat([sums, maxes]).tolist()
    return flat[:ns], flat[ns:]

  def _floats_from_packed(self, pend):
    """Decode the statistics vector of bm_step_worker (layout: include/bm_gar.h)."""
    vec = pend["packed"] if pend["prev"] is None else torch.cat([pend["packed"], pend["prev"][:1]])
    v = vec.tolist()  # the only synchronisation
    nan = math.nan
    att, k_s, k_h, k_a = pend["has_attack"], pend["ks"], self.h, self.f_real

    def dev(x, k):
      return math.sqrt(x / (k - 1)) if k >= 2 else nan

    def cos(i, j):
      if not att and 3 in (i, j):
        return nan
      return v[8 + 4 * i + j] / math.sqrt(v[8 + 5 * i]) / math.sqrt(v[8 + 5 * j])

    res = {
      "l2_origin": math.sqrt(v[7]) if pend["has_l2"] else nan,
      "sampled_norm_avg": math.sqrt(v[0]), "sampled_norm_dev": dev(v[1], k_s), "sampled_norm_max": v[26],
      "honest_norm_avg": math.sqrt(v[2]), "honest_norm_dev": dev(v[3], k_h), "honest_norm_max": v[27],
      "attack_norm_avg": math.sqrt(v[5]) if att else nan, "attack_norm_dev": dev(v[6], k_a) if att else nan,
      "attack_norm_max": v[29] if att else nan,
      "defense_norm_avg": math.sqrt(v[4]), "defense_norm_max": v[28],
      "cosin_splhon": cos(0, 1), "cosin_spldef": cos(0, 2), "cosin_hondef": cos(1, 2),
      "cosin_splatt": cos(0, 3), "cosin_honatt": cos(1, 3), "cosin_attdef": cos(3, 2),
    }
    if pend["npast"] > 0:
      res["cosin_sampled"] = v[24] / math.sqrt(v[0]) / math.sqrt(v[32])
      res["curv_sampled"] = self.mu * v[25]
    else:
      res["cosin_sampled"] = nan
      res["curv_sampled"] = nan
    return res

  def floats(self):
    """Python floats of the study row (attack.py:828-868) for the last run(); synchronises once.
    Idempotent: a second call returns the same dictionary without touching the device."""
    pend = self._pending
    if pend is None:
      raise RuntimeError("floats() needs a run() first")
    if pend["floats"] is not None:
      return pend["floats"]
    if self.gar == "brute":
      self.agg.check_brute()  # (a step replayed from a HIP graph could not check inside run(): here at the latest)
    if "packed" in pend:
      pend["floats"] = self._floats_from_packed(pend)
      return pend["floats"]
    sums, maxes = self._exchange(pend)
    s2, sd, h2, hd = sums[0:4]
    st = sums[4:24]          # Gram 4 x 4 | <s, past>, <s, C> | sum avg_a^2, sum_i |a_i - avg_a|^2
    att = self.f_real > 0
    nc = 4 if att else 3
    g = [[st[4 * a + b] for b in range(4)] for a in range(4)]
    ex = st[16:18]
    d2 = g[2][2]
    a2, ad = (st[18], st[19]) if att else (math.nan, math.nan)
    l2 = math.sqrt(sums[24]) if pend["has_l2"] else math.nan
    prev_norm = math.sqrt(sums[25]) if pend["prev_s2"] is not None else math.nan
    smax, hmax = maxes[0], maxes[1]
    amax = maxes[2] if att else math.nan
    dmax = maxes[3]
    k_s, k_h, k_a = pend["ks"], self.h, self.f_real

    def dev(v, k):
      return math.sqrt(v / (k - 1)) if k >= 2 else math.nan

    def cos(i, j):
      if i >= nc or j >= nc:
        return math.nan
      return g[i][j] / math.sqrt(g[i][i]) / math.sqrt(g[j][j])

    res = {
      "l2_origin": l2,
      "sampled_norm_avg": math.sqrt(s2), "sampled_norm_dev": dev(sd, k_s), "sampled_norm_max": smax,
      "honest_norm_avg": math.sqrt(h2), "honest_norm_dev": dev(hd, k_h), "honest_norm_max": hmax,
      "attack_norm_avg": math.sqrt(a2) if att else math.nan,
      "attack_norm_dev": dev(ad, k_a) if att else math.nan, "attack_norm_max": amax,
      "defense_norm_avg": math.sqrt(d2), "defense_norm_max": dmax,
      "cosin_splhon": cos(0, 1), "cosin_spldef": cos(0, 2), "cosin_hondef": cos(1, 2),
      "cosin_splatt": cos(0, 3), "cosin_honatt": cos(1, 3), "cosin_attdef": cos(3, 2),
    }
    if pend["npast"] > 0:
      res["cosin_sampled"] = ex[0] / math.sqrt(s2) / prev_norm
      res["curv_sampled"] = self.mu * ex[1]
    else:
      res["cosin_sampled"] = math.nan
      res["curv_sampled"] = math.nan
    pend["floats"] = res
    return res

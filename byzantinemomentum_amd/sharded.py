"""Dimension-sharded aggregation over several GPUs (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU aggregation; this is the MI355X scaling axis of
the path.  Every rank holds the SAME n workers but only its slice [lo, hi) of the d coordinates:

  * coordinate-wise rules (median, trmean, phocas, meamed), Bulyan pass 2, selected means and
    momentum are independent per coordinate  -> no communication at all;
  * distance-based selection needs ONE exchange: each rank computes partial SQUARED distances over
    its slice, a single all-reduce(sum) of an n x n fp64 buffer (<= 32 KB, latency-bound on xGMI —
    never a d-sized collective) gives every rank the same matrix, and every rank runs the same
    deterministic score/rank kernel -> identical selections without further traffic;
  * Aksel: all-reduce of the n partial squared distances to the (local) median slices;
  * statistics: ONE all-gather of the packed per-rank scalars (sums and maxima together), reduced
    locally in rank order (`exchange`);
  * the full aggregated vector, when a consumer needs it on every rank, is ONE all-gather of the
    d/P slices (the only bandwidth collective: 7 peers on 7 links in parallel).

The compute legs are injected (`backend`): the product uses the HIP backend below; the CPU/gloo
tests of tests/test_sharded_gloo.py inject an oracle-backed one to exercise partitioning and
collectives without a GPU.  With world_size 1 no collective is ever issued.
"""

import ctypes
import math
import warnings

import torch
import torch.distributed as dist

from . import _lib

__all__ = ["shard_bounds", "owned_workers", "Shards", "ShardedAggregator", "HipBackend", "NativeComm"]


class Shards(list):
  """The n gradients restricted to this rank's coordinate slice, which KNOW the length of the whole vectors
  (`d_total`): what `ShardedAggregator.to_dim_sharded` and `shard_rows` return.  The distance-based rules read the
  total from it, so that an aggregator serves any sequence of vector lengths (per-layer aggregation, several
  models) without a collective to find the total out and without guessing."""

  def __init__(self, rows=(), d_total=None):
    super().__init__(rows)
    self.d_total = None if d_total is None else int(d_total)


def shard_bounds(d, world_size, rank, align=64):
  """Contiguous slice [lo, hi) of rank `rank`: ceil(d/P) rounded up to `align` coordinates
  (256 B, keeps every shard 16-byte aligned for the float4 kernels). Trailing ranks may be empty."""
  per = -(-d // world_size)
  per = -(-per // align) * align
  lo = min(rank * per, d)
  hi = min(lo + per, d)
  return lo, hi


def owned_workers(n, world_size, rank):
  """Workers whose gradients rank `rank` produces in the worker-parallel layout: rank, rank+P, ...
  (round robin, so that a worker count that P does not divide still spreads evenly)."""
  return list(range(rank, n, world_size))


class NativeComm:
  """RCCL communicator owned by libbm_gar.so (bm_comm_*): lets the sharded rules run as ONE C call
  per aggregation with the all-reduce issued from C on the caller's stream.  Bootstrapped through the
  existing torch.distributed group (rank 0's unique id is broadcast as a Python object)."""

  def __init__(self, group=None, vote=None):
    """vote(ok) -> bool: "did this step succeed on EVERY rank?" (a collective of the torch group).  With it, no rank
    can be left alone in a collective: the unique id is only broadcast once every rank knows rank 0 made one, and a
    failure of bm_comm_init on one rank is learnt by all (the caller's last vote).  Without it (vote=None) a failure
    simply raises on the rank it happens on."""
    lib = _lib.load()
    if not lib.bm_comm_available():
      raise RuntimeError("RCCL could not be bound by libbm_gar.so")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ident, failure = [None], None
    if rank == 0:
      try:
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.bm_comm_unique_id(buf), "bm_comm_unique_id")
        ident[0] = buf.raw
      except Exception as err:  # noqa: BLE001
        failure = err
    # rank 0 must not skip the broadcast its peers are about to enter: first everyone learns whether there is an id
    if vote is not None:
      if not vote(failure is None):
        raise RuntimeError(f"bm_comm_unique_id failed on rank 0: {failure}" if failure is not None else
                           "bm_comm_unique_id failed on rank 0")
    elif failure is not None:
      raise failure
    dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    handle = ctypes.c_void_p()
    _lib.check(lib.bm_comm_init(ctypes.byref(handle), world, rank, ctypes.c_char_p(ident[0])), "bm_comm_init")
    self.handle = handle
    self.world_size = world

  def close(self):
    if self.handle is not None and self.handle.value:
      _lib.load().bm_comm_destroy(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # noqa: BLE001  (interpreter shutdown)
      pass


class HipBackend:
  """Compute legs on the local MI355X (libbm_gar.so).  No CPU fallback."""

  def __init__(self):
    from . import gars, stats
    self.gars = gars
    self.stats = stats

  # -- aggregation rules ------------------------------------------------------ #

  def pairwise_sqdist(self, gradients, d_total=None):
    return self.gars.pairwise_sqdist(gradients, d_total)

  def rank(self, sq, n, f, m, mode):
    order, _ = self.gars.rank_from_sqdist(sq, n, f, m, mode)
    return order

  def selected_mean(self, gradients, order, m):
    return self.gars.selected_mean(gradients, order, m)

  def bulyan_pass2(self, gradients, order, f, m):
    return self.gars.bulyan_pass2(gradients, order, f, m)

  def colwise(self, rule, gradients, f):
    return getattr(self.gars, rule)(gradients, f=f) if rule != "median" else self.gars.median(gradients)

  def aksel_sqdist(self, gradients):
    return self.gars.aksel_sqdist(gradients)[:len(gradients)]

  def argsort(self, keys, n):
    return self.gars.stable_argsort(keys, n)

  def brute_select(self, dist_host, n, f):
    return self.gars.brute_select_host(dist_host, n, f)

  def brute_select_device(self, sq, n, f):
    """(sel, status) on the device; status -1 = no subset of n - f rows has a finite diameter (brute.py:68)."""
    return self.gars.brute_select_device(sq, n, f)

  def sharded_rule(self, name, comm, gradients, f, m, d_total=None):
    """Multi-Krum / Bulyan of the local slice in one C call (bm_sharded_krum / bm_sharded_bulyan);
    comm: NativeComm or None (one rank); d_total: length of the whole vectors (default: this shard's)."""
    gars = self.gars
    n, d, device = gars._validate(gradients)
    lib = _lib.load()
    out = torch.empty(d, dtype=torch.float32, device=device)
    nbytes = lib.bm_sharded_workspace_bytes(n, d)
    ws = gars._Scratch.get(device, "ws_sharded", nbytes=int(nbytes))
    fn = lib.bm_sharded_krum if name == "krum" else lib.bm_sharded_bulyan
    with torch.cuda.device(device):
      _lib.check(fn(comm.handle if comm is not None else None, _lib.pointer_table(gradients), n, d,
                    int(d_total) if d_total is not None else d, f, m, gars._ptr(out), None, gars._ptr(ws), gars._stream(device)), "bm_sharded_" + name)
    return out

  def index_tensor(self, indices, like):
    return torch.tensor(indices, dtype=torch.int32, device=like.device)

  def sum_over_ranks(self, agg, value):
    """A host integer summed over the ranks (shard lengths: once per shape, not per call)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    return int(agg.all_reduce_sum(t).item())

  # -- step statistics and momentum -------------------------------------------- #

  def stack_stats(self, samples, scale=None, attack="empire", want_avg=True, direction=False):
    return self.stats.stack_stats_async(samples, scale=scale, attack=attack, want_avg=want_avg, direction=direction)

  def momentum_stats(self, sampled, buffers, mu, omd, factors, scale, attack, direction=False):
    return self.stats.momentum_stats(sampled, buffers, mu, omd, factors, scale, attack, direction)

  def momentum_stats_colwise(self, *args, **kwargs):
    return self.stats.momentum_stats_colwise(*args, **kwargs)

  def stack_stats_colwise(self, *args, **kwargs):
    return self.stats.stack_stats_colwise(*args, **kwargs)

  def stack_stats_sqdist(self, *args, **kwargs):
    return self.stats.stack_stats_sqdist(*args, **kwargs)

  def momentum_stats_sqdist(self, *args, **kwargs):
    return self.stats.momentum_stats_sqdist(*args, **kwargs)

  def multi_fma3(self, outs, ps, qs, a, b, p_scale=None):
    return self.stats.multi_fma3(outs, ps, qs, a, b, p_scale)

  @property
  def device_search_rules(self):
    """Rules whose factor search runs on the device (step.AggregationStep, line_search="auto")."""
    return self.stats.DEVICE_SEARCH_RULES

  def attack_search_device(self, *args, **kwargs):
    return self.stats.attack_search_device(*args, **kwargs)

  def device_search(self, device, evals, negative=False):
    """The exploration's cursor in device memory (stats.DeviceSearch): the host queues every evaluation of a search."""
    return self.stats.DeviceSearch(device, evals, negative)

  def multi_scale(self, ys, factors):
    return self.stats.multi_scale(ys, factors)

  def row_sqnorms(self, gradients):
    return self.stats.row_sqnorms(gradients).contiguous()

  def clip_factors_from_sq(self, sq, k, clip):
    return self.stats.clip_factors_from_sq(sq, k, clip)

  def study_stats(self, *args, **kwargs):
    return self.stats.study_stats(*args, **kwargs)

  def colwise_eval_supported(self, rule, n):
    return self.stats.colwise_eval_supported(rule, n)

  def colwise_eval(self, *args, **kwargs):
    return self.stats.colwise_eval(*args, **kwargs)

  def sqdist2(self, a, b):
    return self.stats.sqdist2(a, b)

  def attack_ranking_device(self, *args, **kwargs):
    return self.stats.attack_ranking_device(*args, **kwargs)

  def bulyan_pass2_eval_supported(self, n, f, m, d=0):
    return self.stats.bulyan_pass2_eval_supported(n, f, m, d)

  def bulyan_pass2_eval(self, *args, **kwargs):
    return self.stats.bulyan_pass2_eval(*args, **kwargs)

  def order_pair_supported(self, h):
    return self.stats.order_pair_supported(h)

  def order_pair(self, rows, il, ih):
    return self.stats.order_pair(rows, il, ih)

  def study_dots(self, core, extra):
    return self.stats.study_dots(core, extra)

  def step_worker(self, comm, *args, **kwargs):
    """One whole step with worker-side momentum in one C call (bm_step_worker)."""
    return self.stats.step_worker(comm, *args, **kwargs)


class ShardedAggregator:
  """Aggregation rules over gradients whose coordinates are sharded across the ranks of `group`."""

  def __init__(self, backend=None, group=None, force_collectives=False, native_comm="auto", local_only=False):
    """native_comm: "auto" (default) gives the HIP backend its own RCCL communicator when there is
    more than one rank, so that Multi-Krum / Bulyan are single C calls; False keeps every collective
    in torch.distributed; True insists (raises if RCCL cannot be bound).
    local_only: a single-rank aggregator whatever the state of torch.distributed (the whole vectors are here:
    no collective is ever issued) — e.g. the unsharded reference computation inside a multi-rank job."""
    self.backend = backend if backend is not None else HipBackend()
    self.group = group
    initialised = dist.is_available() and dist.is_initialized() and not local_only
    self.world_size = dist.get_world_size(group) if initialised else 1
    self.rank = dist.get_rank(group) if initialised else 0
    # force_collectives: issue the all-reduce / all-gather calls even with one rank (used to
    # exercise the RCCL path on a single-GPU box); never set in production with world_size 1
    self.collective = self.world_size > 1 or (force_collectives and initialised)
    self.native = None
    self.single_call = isinstance(self.backend, HipBackend) and native_comm is not False
    if self.single_call and self.collective:
      self.native, failure = self._create_native(group)
      if self.native is None:
        if native_comm is True:
          raise RuntimeError(f"libbm_gar RCCL communicator unavailable on at least one rank ({failure})")
        warnings.warn(f"libbm_gar RCCL communicator unavailable ({failure}); using torch.distributed collectives")
        self.single_call = False
    self.brute_status = None
    self._total = None        # (length of the whole vectors, this rank's shard length when it was determined)

  def shard_rows(self, rows, d=None):
    """This rank's slice (shard_bounds) of n whole gradients held on every rank, as `Shards` carrying the total."""
    d = int(rows[0].shape[0]) if d is None else int(d)
    lo, hi = shard_bounds(d, self.world_size, self.rank)
    return Shards([r[lo:hi] for r in rows], d_total=d)

  def _total_of(self, local, d_total=None):
    """Total length for a distance pass over `local`: stated, else carried by the shards, else `total_length`."""
    if d_total is None:
      d_total = getattr(local, "d_total", None)
    return self.total_length(local[0].numel(), d_total)

  def total_length(self, d_local, d_total=None):
    """Length of the WHOLE vectors (all shards), the number every rank must hand to the distance pass so that a
    short or empty trailing shard plans it exactly like its peers (bm_gar.h, bm_sharded_krum).

    Stated by the caller (`d_total`, or carried by `Shards`: to_dim_sharded / shard_rows), or, for plain lists,
    determined ONCE per aggregator: the first call sums the shard lengths over
    the ranks (one tiny all-reduce, which every rank enters because every rank makes its first call), later calls
    return that number without any communication.  An aggregator therefore serves ONE vector length; a rank that
    notices another shard length raises instead of guessing — deciding locally whether to repeat the collective could
    leave ranks whose shard length did not change outside of it (a hang).  For another length pass `d_total`, call
    `reset_total_length()` on every rank, or use another aggregator."""
    if d_total is not None:
      return int(d_total)
    if not self.collective:
      return int(d_local)
    if self._total is None:
      self._total = (int(self.backend.sum_over_ranks(self, int(d_local))), int(d_local))
    elif self._total[1] != int(d_local):
      raise ValueError(f"this ShardedAggregator determined the total length {self._total[0]} from shards of "
                       f"{self._total[1]} coordinates and is now given a shard of {d_local}: pass d_total=, or call "
                       f"reset_total_length() on every rank")
    return self._total[0]

  def reset_total_length(self):
    """Forget the total length (collective-free; the next distance pass determines it again, on every rank)."""
    self._total = None

  def _create_native(self, group):
    """The library's own communicator, on EVERY rank or on none: a rank that fell back alone would issue
    torch.distributed collectives while its peers wait inside RCCL calls of the other communicator (a hang,
    not a degradation).  Every step of the bootstrap that can fail locally is followed by a vote (MIN over the
    ranks of a success flag, through the torch group that is known to work); the unique id is only
    broadcast, and bm_comm_init only entered, when every rank can go on."""
    lib = _lib.load()

    def everyone(ok):
      flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32,
                          device=torch.device("cuda", torch.cuda.current_device()))
      dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
      return flag.item() >= 1.0

    if not everyone(bool(lib.bm_comm_available())):
      return None, "RCCL could not be bound by libbm_gar.so"
    comm, failure = None, None
    try:
      comm = NativeComm(group, vote=everyone)  # (votes once inside: "rank 0 has an id", before the id is broadcast)
    except Exception as err:  # noqa: BLE001
      failure = err
    if not everyone(comm is not None):
      if comm is not None:
        comm.close()
      return None, failure if failure is not None else "bm_comm_init failed on another rank"
    return comm, None

  # -- collectives (never called when world_size == 1) ----------------------- #
  # Every exchange of this class goes through the three primitives below (a subclass may route them elsewhere:
  # tests/test_gpu_zz_multirank.py stages them through host tensors to run several ranks on ONE GPU over gloo).

  def _all_reduce(self, tensor, op=None):
    if self.collective:
      dist.all_reduce(tensor, op=(op or dist.ReduceOp.SUM), group=self.group)
    return tensor

  def _all_gather_into(self, everyone, mine):
    dist.all_gather_into_tensor(everyone, mine, group=self.group)

  def _all_to_all(self, recv, send):
    dist.all_to_all_single(recv, send, group=self.group)

  def all_reduce_sum(self, tensor):
    """In-place sum over the ranks of a small device tensor (no-op with one rank)."""
    return self._all_reduce(tensor)

  def exchange(self, sums, maxes):
    """ONE collective for every scalar of a step: each rank contributes [sums | maxes] (fp64), an
    all-gather hands every rank all contributions, which it reduces itself in rank order — sums and
    (NaN-propagating) maxima from the same message, bitwise identical on every rank."""
    if not self.collective:
      return sums, maxes
    ns = sums.shape[0]
    mine = torch.cat([sums, maxes]).contiguous()
    everyone = torch.empty(self.world_size * mine.shape[0], dtype=mine.dtype, device=mine.device)
    self._all_gather_into(everyone, mine)
    everyone = everyone.view(self.world_size, -1)
    total = everyone[0, :ns].clone()
    for r in range(1, self.world_size):
      total += everyone[r, :ns]
    return total, everyone[:, ns:].max(dim=0).values

  def all_gather_output(self, local_out, d):
    """Full d-vector on every rank from the per-rank slices (shard_bounds layout)."""
    if not self.collective:
      return local_out
    lo0, hi0 = shard_bounds(d, self.world_size, 0)
    per = hi0 - lo0
    padded = torch.zeros(per, dtype=local_out.dtype, device=local_out.device)
    padded[:local_out.shape[0]] = local_out
    full = torch.empty(per * self.world_size, dtype=local_out.dtype, device=local_out.device)
    self._all_gather_into(full, padded)
    return full[:d]

  def to_dim_sharded(self, my_gradients, n, d, dtype=torch.float32, device=None):
    """Worker-major -> dimension-major in ONE all-to-all (the step before the
    path when the honest gradients are PRODUCED in parallel, experiments/model.py:333-366 run once per
    worker).  `my_gradients`: the full-length gradients of owned_workers(n, P, rank), in that order.
    Returns the n gradients restricted to this rank's coordinate slice (shard_bounds), in worker order, as `Shards`
    (they carry d, so the distance-based rules need neither a d_total argument nor a collective to learn it):
    views into one receive buffer, 256-byte aligned, ready for the rules.  Each rank sends (P-1)/P of
    what it produced — n/P * d * 4 bytes spread over the P-1 peers' links at once — never more."""
    world, rank = self.world_size, self.rank
    mine = owned_workers(n, world, rank)
    if len(my_gradients) != len(mine):
      raise ValueError(f"rank {rank} must pass the gradients of workers {mine}")
    if not self.collective:  # (one rank with forced collectives still goes through the exchange: that is how a
      return Shards(my_gradients, d_total=d)  #  single-GPU box exercises the RCCL all-to-all)
    per = -(-(-(-d // world)) // 64) * 64    # padded shard length: ceil(d / P) rounded up to 64 coordinates (256 B)
    n_max = -(-n // world)
    # a rank that owns no worker (n < P) still takes part in the exchange with an all-zero send buffer:
    # dtype / device come from its gradients when it has some, else from the arguments
    if my_gradients:
      dtype, device = my_gradients[0].dtype, my_gradients[0].device
    elif device is None:
      device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    send = torch.zeros((world, n_max, per), dtype=dtype, device=device)
    for j, g in enumerate(my_gradients):
      full = g.shape[0] // per
      send[:full, j, :] = g[:full * per].view(full, per)
      if full < world and g.shape[0] > full * per:
        send[full, j, :g.shape[0] - full * per] = g[full * per:]
    recv = torch.empty_like(send)
    self._all_to_all(recv.view(-1), send.view(-1))
    lo, hi = shard_bounds(d, world, rank)
    return Shards([recv[i % world, i // world, :hi - lo] for i in range(n)], d_total=d)

  # -- rules ------------------------------------------------------------------ #

  def median(self, local):
    return self.backend.colwise("median", local, 0)

  def trmean(self, local, f):
    return self.backend.colwise("trmean", local, f)

  def phocas(self, local, f):
    return self.backend.colwise("phocas", local, f)

  def meamed(self, local, f):
    return self.backend.colwise("meamed", local, f)

  def global_sqdist(self, local, d_total=None):
    """All-reduced n x n squared-distance matrix (every rank gets the same bits: the sum runs over
    the same P partial matrices in the collective's fixed order)."""
    # the precision plan follows the length of the whole vectors, the same number on every rank (total_length)
    sq = self.backend.pairwise_sqdist(local, d_total=self._total_of(local, d_total))  # fresh, reduced in place
    self._all_reduce(sq)
    return sq

  def krum(self, local, f, m=None, d_total=None):
    n = len(local)
    if m is None:
      m = n - f - 2
    if self.single_call:
      return self.backend.sharded_rule("krum", self.native, local, f, m, self._total_of(local, d_total))
    order = self.backend.rank(self.global_sqdist(local, d_total), n, f, m, _lib.RANK_KRUM)
    return self.backend.selected_mean(local, order, m)

  def bulyan(self, local, f, m=None, d_total=None):
    n = len(local)
    if m is None:
      m = n - f - 2
    if self.single_call:
      return self.backend.sharded_rule("bulyan", self.native, local, f, m, self._total_of(local, d_total))
    order = self.backend.rank(self.global_sqdist(local, d_total), n, f, m, _lib.RANK_BULYAN)
    return self.backend.bulyan_pass2(local, order, f, m)

  def rule_from_sq(self, name, local, local_sq, f, m=None):
    """Multi-Krum / Bulyan when the squared distances of the local shard are already known (the first pass of a step
    produced them, stats.momentum_stats_sqdist): all-reduce, rank, average / pass 2."""
    n = len(local)
    if m is None:
      m = n - f - 2
    sq = self._all_reduce(local_sq)  # in place: a fresh tensor per call
    order = self.backend.rank(sq, n, f, m, _lib.RANK_KRUM if name == "krum" else _lib.RANK_BULYAN)
    if name == "krum":
      return self.backend.selected_mean(local, order, m)
    return self.backend.bulyan_pass2(local, order, f, m)

  def aksel(self, local, f, mode="mid"):
    n = len(local)
    count = (n + 1) // 2 if mode == "mid" else n - f
    sq = self.backend.aksel_sqdist(local)
    self._all_reduce(sq)
    return self.backend.selected_mean(local, self.backend.argsort(sq, n), count)

  def brute(self, local, f, d_total=None, check=False):
    """Brute rule: all-reduced distances, then the (deterministic) subset search on every rank — on the device with
    the HIP backend (same bits in, same selection out on every rank, no host round trip).  The search's status is
    kept in `self.brute_status` (device int32[1]) — per aggregator, not per process.  Unchecked, a search without a
    usable answer shows in the result (status -1: non-finite where a bad gradient is; -2, the node budget: NaN
    everywhere).  check=True reads the status here (one 4-byte synchronisation; every rank holds the same status, the
    search being a function of the all-reduced matrix): -1 raises like the reference (brute.py:68), -2 repeats the
    search on the host, which has no budget — AggregationStep does this before it hands out the defense vector."""
    n = len(local)
    sq = self.global_sqdist(local, d_total)
    if hasattr(self.backend, "brute_select_device"):
      sel, self.brute_status = self.backend.brute_select_device(sq, n, f)
      if check and not (sq.is_cuda and torch.cuda.is_current_stream_capturing()):
        code = int(self.brute_status.item())
        if code == -2:
          host = self.backend.brute_select(sq.sqrt().cpu().contiguous(), n, f)
          sel = self.backend.index_tensor(host, local[0])
          self.brute_status = None
        elif code != 0:
          from . import gars
          raise RuntimeError(gars.BRUTE_NO_SUBSET)
      return self.backend.selected_mean(local, sel, n - f)
    self.brute_status = None
    sel = self.backend.brute_select(sq.sqrt().cpu().contiguous(), n, f)
    return self.backend.selected_mean(local, self.backend.index_tensor(sel, local[0]), n - f)

  def check_brute(self):
    """Raise when the latest unchecked brute() of THIS aggregator had no usable answer (-1: the reference's assertion,
    brute.py:68; -2: the node budget); syncs."""
    status = getattr(self, "brute_status", None)
    if status is not None:
      from . import gars
      gars.brute_check(status)

  def average(self, local):
    n = len(local)
    return self.backend.selected_mean(local, self.backend.index_tensor(list(range(n)), local[0]), n)

  def cge(self, local, f):
    n = len(local)
    sq = self.backend.row_sqnorms(local)
    self._all_reduce(sq)
    return self.backend.selected_mean(local, self.backend.argsort(sq, n), n - f)

  def compute_avg_dev_max(self, local_samples):
    """Sharded tools.compute_avg_dev_max: (local slice of the average, norm, deviation, max)."""
    k = len(local_samples)
    if k == 0:
      return None, math.nan, math.nan, math.nan
    avg, out3 = self.backend.stack_stats(local_samples)
    sums, mx = self.exchange(out3[:2], out3[2:])  # one collective (none with a single rank)
    norm2, dev2 = sums.tolist()
    amax = mx.item()
    dev = math.sqrt(dev2 / (k - 1)) if k >= 2 else math.nan
    return avg, math.sqrt(norm2), dev, amax

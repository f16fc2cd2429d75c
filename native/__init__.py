"""`native` — the module name the reference already probes for (`try: import native`,
aggregators/krum.py:22-26, bulyan.py:22-26, median.py:22-26, brute.py:23-27).

With /root/repo on PYTHONPATH the UNMODIFIED reference registers, at `import aggregators`:
    native-krum    -> native.krum.aggregate(gradients, f, m)      (krum.py:82-96,159-166)
    native-bulyan  -> native.bulyan.aggregate(gradients, f, m)    (bulyan.py:86-100,137-144)
    native-median  -> native.median.aggregate(gradients)          (median.py:41-49,80-87)
    native-brute   -> native.brute.aggregate(gradients, f)        (brute.py:82-91,149-156)
and this module registers the remaining rules itself through the reference's own registry
(aggregators/__init__.py:71-86): native-trmean, native-phocas, native-meamed, native-aksel,
native-average, native-cge — including the `influence` hooks (attack acceptation ratio).

Everything executes in the HIP kernels of byzantinemomentum_amd; gradients must be on a GPU
(`--device cuda:0` / `--device-gar cuda:0`). There is no CPU fallback.
"""

import sys

import byzantinemomentum_amd as _bm
from byzantinemomentum_amd import gars as _gars

__all__ = ["krum", "bulyan", "median", "brute"]


class _Rule:
  """Object with an `aggregate` attribute, as the reference's call sites expect."""

  def __init__(self, name, aggregate, doc):
    self.__name__ = name
    self.aggregate = aggregate
    self.__doc__ = doc

  def __repr__(self):
    return f"<native rule {self.__name__!r} (MI355X HIP)>"


def _krum_aggregate(gradients, f, m=None):
  _late_bind()
  return _gars.krum(gradients, f, m)


def _bulyan_aggregate(gradients, f, m=None):
  _late_bind()
  return _gars.bulyan(gradients, f, m)


def _median_aggregate(gradients):
  return _gars.median(gradients)


def _brute_aggregate(gradients, f):
  _late_bind()
  # check=True: the reference asserts that a subset with a finite diameter exists (brute.py:68); so does the plugin
  return _gars.brute(gradients, f, check=True)


krum = _Rule("krum", _krum_aggregate, "Multi-Krum, see byzantinemomentum_amd.gars.krum")
bulyan = _Rule("bulyan", _bulyan_aggregate, "Bulyan over Multi-Krum, see byzantinemomentum_amd.gars.bulyan")
median = _Rule("median", _median_aggregate, "Coordinate-wise median, see byzantinemomentum_amd.gars.median")
brute = _Rule("brute", _brute_aggregate, "Brute rule, see byzantinemomentum_amd.gars.brute")


# ---------------------------------------------------------------------------- #
# Attack acceptation ratios: |selected ∩ attacks| / |selected|, from the selected INDICES
# (gradients = honests + attacks, so index >= len(honests) <=> `gradient is attack`,
# aggregators/krum.py:144-150, brute.py:132-140, aksel.py:97-105, cge.py:88-96).

def _ratio(selected, nb_honests):
  if len(selected) == 0:
    return 0.
  return sum(1 for i in selected if i >= nb_honests) / len(selected)


def _influence_krum(honests, attacks, f, m=None, **kwargs):
  return _ratio(_gars.krum_selection(honests + attacks, f, m), len(honests))


def _influence_brute(honests, attacks, f, **kwargs):
  return _ratio(_gars.brute_selection(honests + attacks, f), len(honests))


def _influence_aksel(honests, attacks, f, mode="mid", **kwargs):
  return _ratio(_gars.aksel_selection(honests + attacks, f, mode), len(honests))


def _influence_cge(honests, attacks, f, **kwargs):
  n = len(honests) + len(attacks)
  return _ratio(_gars.cge_selection(honests + attacks, f)[:n - f].tolist(), len(honests))


def _influence_average(honests, attacks, **kwargs):
  return len(attacks) / (len(honests) + len(attacks))


# ---------------------------------------------------------------------------- #
# Registration of the rules the reference has no `native` call site for

def _reference_check(module, fallback):
  """Use the reference's own `check` of the rule, looked up lazily (`native` is imported from inside
  the first rule module that probes for it, before aggregators/trmean.py, aksel.py... have run).  Only
  when that module does not exist at all (this package used with another registry) the generic
  validator below is used."""
  def check(**kwargs):
    mod = sys.modules.get(f"aggregators.{module}")
    fn = getattr(mod, "check", None)
    return (fn or fallback)(**kwargs)
  return check


def _generic_check(min_rows_per_f=0, modes=None):
  """Validator for registries other than the reference's: None when the arguments can be served,
  otherwise a message (the contract of aggregators/__init__.py:23-27)."""
  def check(gradients=None, f=None, mode=None, **kwargs):
    if not isinstance(gradients, list) or not gradients:
      return "gradients must be a non-empty list of tensors"
    if min_rows_per_f:
      if not isinstance(f, int) or f < 1:
        return f"f must be a positive integer (got {f!r})"
      if len(gradients) < min_rows_per_f * f + 1:
        return f"{len(gradients)} gradients cannot tolerate f = {f}: at least {min_rows_per_f * f + 1} are needed"
    if modes is not None and mode is not None and mode not in modes:
      return f"mode must be one of {modes} (got {mode!r})"
    return None
  return check


_check_list = _generic_check()
_check_f_half = _generic_check(min_rows_per_f=2)
_check_aksel = _generic_check(min_rows_per_f=2, modes=("mid", "n-f"))


_registered = False


def _register_extra():
  """Called at import: if we are being imported from inside `aggregators` (its `register` and
  `gars` exist before the rule modules run, aggregators/__init__.py:71,89-93), add our rules."""
  global _registered
  agg = sys.modules.get("aggregators")
  register = getattr(agg, "register", None)
  if _registered or register is None or not hasattr(agg, "gars"):
    return
  _registered = True
  register("native-trmean", _gars.trmean, _reference_check("trmean", _check_f_half))
  register("native-phocas", _gars.phocas, _reference_check("trmean", _check_f_half))
  register("native-meamed", _gars.meamed, _reference_check("trmean", _check_f_half))
  register("native-aksel", _gars.aksel, _reference_check("aksel", _check_aksel), influence=_influence_aksel)
  register("native-average", _gars.average, _reference_check("average", _check_list),
           influence=_influence_average)
  register("native-cge", _gars.cge, _reference_check("cge", _check_list), influence=_influence_cge)


_bound = False


def _late_bind():
  """The reference registers native-krum/-brute WITHOUT an influence function (krum.py:164,
  brute.py:154), which turns the "Attack acceptation ratio" column into NaN.  The GAR objects are
  plain functions carrying attributes (aggregators/__init__.py:61-69), so attach ours on first use."""
  global _bound
  if _bound:
    return
  agg = sys.modules.get("aggregators")
  table = getattr(agg, "gars", None)
  if not isinstance(table, dict):
    return
  _bound = True
  for name, fn in (("native-krum", _influence_krum), ("native-brute", _influence_brute)):
    rule = table.get(name)
    if rule is not None and getattr(rule, "influence", None) is None:
      rule.influence = fn


_register_extra()

"""Import the REAL reference to validate the oracle, to generate the golden fixtures and — staged —
as the CPU reference on the GPU box.  TEST INFRASTRUCTURE: the product never imports this.

Two places hold a checkout: /root/reference (the build container, read-only) and the git-ignored
oracle/_ref/reference that `scripts/stage_reference.sh` fills from it (`__graft_entry__.build()` runs
the recipe whenever /root/reference is present).  The staged copy rides the gpurun snapshot, so the
`-m gpu` tests that drive the unmodified attack.py, the full-size fp32 parity tests and bench.py's
`cpu_baseline` leg (kind "reference") find the reference's own code on the GPU box; none of them reads
/root/reference at run time.

The reference's `tools` package wraps sys.stdout/sys.stderr and replaces sys.excepthook at import
(tools/__init__.py:215-246); we restore them so pytest's capture keeps working.
"""

import os
import sys

STAGED_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference")


def _find():
  env = os.environ.get("BM_REFERENCE_DIR")
  for cand in ([env] if env else []) + ["/root/reference", STAGED_DIR]:
    if os.path.isfile(os.path.join(cand, "aggregators", "__init__.py")):
      return cand
  return env or "/root/reference"


REFERENCE_DIR = _find()


def available():
  return os.path.isfile(os.path.join(REFERENCE_DIR, "aggregators", "__init__.py"))


_cache = {}


def load(with_native=False):
  """Return (aggregators, tools) modules of the reference.

  with_native=False hides our `native` package so that only the reference's own PyTorch rules are
  registered (the pure oracle); with_native=True lets the reference discover it (drop-in test).
  """
  key = bool(with_native)
  if key in _cache:
    return _cache[key]
  if not available():
    raise RuntimeError(f"reference checkout not found at {REFERENCE_DIR}")
  saved = (sys.stdout, sys.stderr, sys.excepthook)
  saved_path = list(sys.path)
  saved_native = sys.modules.get("native")
  for name in [m for m in sys.modules if m == "aggregators" or m.startswith("aggregators.") or m == "tools"
               or m.startswith("tools.")]:
    del sys.modules[name]
  try:
    sys.path.insert(0, REFERENCE_DIR)
    if not with_native:
      sys.modules["native"] = None  # `import native` raises ImportError -> reference falls back
    else:
      sys.modules.pop("native", None)
    import aggregators
    import tools
  finally:
    sys.stdout, sys.stderr, sys.excepthook = saved
    sys.path[:] = saved_path
    if not with_native:
      if saved_native is not None:
        sys.modules["native"] = saved_native
      else:
        sys.modules.pop("native", None)
  _cache[key] = (aggregators, tools)
  return _cache[key]

"""Import the REAL reference (read-only checkout) to validate the oracle and to generate the
golden fixtures.  Only usable where /root/reference exists (the build container); the GPU box
has no such checkout, so nothing under `-m gpu`, smoke() or bench.py may call this.

The reference's `tools` package wraps sys.stdout/sys.stderr and replaces sys.excepthook at import
(tools/__init__.py:215-246); we restore them so pytest's capture keeps working.
"""

import os
import sys

REFERENCE_DIR = os.environ.get("BM_REFERENCE_DIR", "/root/reference")


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

"""Moved to oracle/step_oracle.py (bench.py's cpu_baseline leg times one step of it); kept as the import the tests use."""

from oracle.step_oracle import *  # noqa: F401,F403
from oracle.step_oracle import COS_KEYS, FLOAT_KEYS, RULES, ReferenceLoop, assert_floats_close  # noqa: F401

"""Host-side timeline (BP_TRACE=1) of one single-proof verification: python benchmarks/verify_trace.py"""
import os, sys
os.environ["BP_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import config_bench as cb  # noqa
from bulletproofs_amd import Engine
eng = Engine(0)
_, _, h = eng.derive_generators(1024)
cb.range_config(eng, eng.lib, "config1", 64, [(7238283, 827361)], 1, 2, h)

"""-m gpu: the superseded GSTS kernel generations of the -DSN_EXPERIMENTAL library (A/B material, not the product).

They must stay correct to be useful as baselines, so they get the same block checks as the production chain.  The
experimental library is selected per PROCESS (SN_EXPERIMENTAL=1 at import of shiftnet_amd.lib), so the checks run in a
child interpreter; the production tests never see that library.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
sys.path[:0] = [%r, %r, %r]
import test_gpu_parity as TP
from shiftnet_amd import lib as L
from shiftnet_amd.engine import Plan
from shiftnet_amd.engine_experimental import ExperimentalEngine
from shiftnet_amd.spec import VARIANTS
from shiftnet_amd.weights import synth_state_dict
assert L.EXPERIMENTAL and L.LIB_PATH.endswith("_exp.so")
for name in ("gshift_deblur2", "gshift_denoise2", "gshift_deblur1"):
    sd = synth_state_dict(name)
    eng = ExperimentalEngine(Plan(VARIANTS[name], sd, TP.DEV))
    for v in (0, 1, 3):
        eng.gsts_v = v
        TP._gsts_pieces((eng, sd), name, "_v%%d" %% v)
        print("ok", name, v, flush=True)
# fused CAB (sn_cab_fused): every storage width it supports, ragged and sub-tile maps, second residual
cache = {}
def engines(name):
    if name not in cache:
        sd = synth_state_dict(name)
        cache[name] = (ExperimentalEngine(Plan(VARIANTS[name], sd, TP.DEV)), sd)
    return cache[name]
for name, pre, c in (("gshift_deblur2", "stage1.concat.", 14), ("gshift_deblur2", "orb1.encoder_level2.1.", 18),
                     ("gshift_deblur1", "stage1.concat.", 24), ("gshift_deblur1", "orb1.encoder_level2.0.", 36),
                     ("gshift_deblur1", "orb1.encoder_level3.0.", 48)):
    for hw in ((20, 44), (13, 70), (2, 3)):
        TP.test_cab.__wrapped__(name, pre, c, hw, engines) if hasattr(TP.test_cab, "__wrapped__") else TP.test_cab(name, pre, c, hw, engines)
    print("okcab", name, c, flush=True)
"""


def test_experimental_chains_match_the_oracle():
    lib = os.path.join(ROOT, "shift-net_amd", "lib", "libshiftnet_hip_exp.so")
    if not os.path.exists(lib):
        pytest.skip("experimental library not built (python shift-net_amd/build.py --experimental)")
    code = CHILD % (ROOT, os.path.join(ROOT, "shift-net_amd"), os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SN_EXPERIMENTAL="1"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok ") == 9 and r.stdout.count("okcab ") == 5

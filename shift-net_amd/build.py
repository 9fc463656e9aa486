#!/usr/bin/env python3
"""Build libshiftnet_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libshiftnet_hip.so")
SOURCES = ["sn_conv.hip", "sn_conv3p.hip", "sn_cabf.hip", "sn_gsts.hip", "sn_gsts2.hip", "sn_gsts3.hip", "sn_phase1r.hip", "sn_f32.hip", "sn_io.hip"]


def needs_build(out: str = OUT) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "shiftnet_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, out: str = OUT, flags=()) -> str:
    """out / flags: one-off variant libraries for A/B runs (tools/); the product is OUT with no flags."""
    if not force and not needs_build(out):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
           "-o", out, *flags] + [os.path.join(CSRC, s) for s in SOURCES]
    for f in os.environ.get("SN_HIPCC_FLAGS", "").split():      # extra compiler flags for one-off builds (e.g. --save-temps)
        cmd.insert(1, f)
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed")
    if verbose:
        sys.stderr.write(r.stderr)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))

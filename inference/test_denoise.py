#!/usr/bin/env python3
"""Drop-in for the reference's ``inference/test_denoise.py`` (same flags, windows, metrics and log lines), running
``basicsr.models.archs.gshift_denoise1.GShiftNet`` on the MI355X HIP kernels.  Logic: shift-net_amd/shiftnet_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basicsr import _paths  # noqa: E402,F401
from shiftnet_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main("gshift_denoise1")

"""Drop-in for the reference's ``basicsr/models/archs/gshift_denoise1.py``: same module path, ``make_model(opt)`` and
``GShiftNet`` API (plugin hook: image_restoration1_model.py:22-25), executed by the MI355X HIP kernels."""
from basicsr import _paths  # noqa: F401  (puts shift-net_amd/ on sys.path)
from shiftnet_amd.arch import CLASSES, make_model as _make_model

GShiftNet = CLASSES["gshift_denoise1"]


def make_model(opt):
    return _make_model("gshift_denoise1", opt)

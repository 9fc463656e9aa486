"""MI355X-native Shift-Net inference path (GSTS + conv encoder-decoder) behind the reference's arch-class API."""
from .spec import VARIANTS  # noqa: F401

import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shift-net_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

"""xcltk_amd - MI355X-native drop-in for the counting hot path of hxj5/xcltk
(`xcltk basefc`, and step 3 of `xcltk baf`): host code in Python, compute in hand-written
HIP behind a C-ABI (include/xck.h)."""
from .config import VERSION, ENGINE

__version__ = VERSION
__all__ = ["__version__", "ENGINE"]

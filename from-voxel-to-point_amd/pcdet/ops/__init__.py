import os as _os
import sys as _sys

# fv2p_native (ctypes binding of libfv2p_ops.so) sits next to the pcdet package.
_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

"""ctypes binding of libfv2p_ops.so (the C ABI declared in include/fv2p_ops.h).

The prototypes are parsed from the header itself so the header stays the single source of
truth; `declared_symbols()` is what the CPU test-suite checks against the built library.
There is no fallback of any kind: if the shared object is missing, `lib()` raises.
PyTorch is used only as the owner of device memory and streams.
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "fv2p_ops.h")
# FV2P_LIB_DIR: another build of the same library (tools/ and the profile scripts point it at lib/dev, the -DFV2P_DEV=1 build
# that reads tuning overrides from the environment; the release library in lib/ reads none)
LIB_DIR = os.environ.get("FV2P_LIB_DIR") or os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libfv2p_ops.so")

_SCALARS = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
    "double": ctypes.c_double, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64,
    "fv2p_stream_t": ctypes.c_void_p,
}
_ELEM = {"float": ctypes.c_float, "int": ctypes.c_int, "int64_t": ctypes.c_int64, "double": ctypes.c_double}


class Proto:
    def __init__(self, name, restype, params):
        self.name, self.restype, self.params = name, restype, params  # params: list of (kind, ctype, n)


def _parse_header(path=HEADER):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(fv2p_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        if ret.endswith("*"):
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _SCALARS[ret.split()[-1]]
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                arr = re.match(r"(?:const )?(\w+) \w+\[(\d+)\]$", a)
                if arr:  # small host-side array, e.g. `const float voxel_size[3]`
                    params.append(("hostarr", _ELEM[arr.group(1)], int(arr.group(2))))
                elif "*" in a:
                    params.append(("ptr", ctypes.c_void_p, 0))
                else:
                    t = a.split()[-2] if len(a.split()) > 1 else a
                    params.append(("scalar", _SCALARS[t], 0))
        protos[name] = Proto(name, restype, params)
    return protos


_PROTOS = None
_LIB = None
_LOCK = threading.Lock()


def declared_symbols():
    global _PROTOS
    if _PROTOS is None:
        _PROTOS = _parse_header()
    return _PROTOS


class Fv2pError(RuntimeError):
    pass


def lib():
    """Loads the HIP library (after torch, so both share torch's libamdhip64) — no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        if not os.path.exists(LIB_PATH):
            raise Fv2pError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the pcdet.ops hot path.")
        import torch  # noqa: F401  (loads libamdhip64.so.7 first so our DT_NEEDED resolves to the same runtime)
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, p in declared_symbols().items():
            fn = getattr(handle, name)  # AttributeError here == header/library mismatch
            fn.restype = p.restype
            fn.argtypes = [ctypes.POINTER(c) if k == "hostarr" else c for k, c, _ in p.params]
        if handle.fv2p_abi_version() != 1:
            raise Fv2pError("libfv2p_ops ABI version mismatch")
        _LIB = handle
    return _LIB


_EXT = False


def torch_ext():
    """The compiled autograd binding lib/fv2p_torch.so (csrc_torch/fv2p_torch.cpp), or None.

    It is an optional second front end of the same library: sparse conv and BatchNorm1d(+ReLU) as C++
    torch::autograd::Functions (no Python in their backward).  FV2P_TORCH_EXT=0 keeps everything on ctypes."""
    global _EXT
    if _EXT is False:
        _EXT = None
        path = os.path.join(LIB_DIR, "fv2p_torch.so")
        if os.environ.get("FV2P_TORCH_EXT", "1") != "0" and os.path.exists(path):
            lib()  # libfv2p_ops.so first (RTLD_GLOBAL), the extension links against it
            import importlib.util
            spec = importlib.util.spec_from_file_location("fv2p_torch", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != 1:
                raise Fv2pError("fv2p_torch.so / libfv2p_ops.so ABI version mismatch")
            _EXT = mod
    return _EXT


def last_error():
    return lib().fv2p_last_error().decode()


_BOUND = {}


def _bind(name):
    l = lib()
    p = declared_symbols()[name]
    fn = getattr(l, name)
    kinds = tuple(k for k, _, _ in p.params)
    arrs = tuple((c * n) if k == "hostarr" else None for k, c, n in p.params)
    ent = (fn, kinds, arrs, p.restype is ctypes.c_int)
    _BOUND[name] = ent
    return ent


def call(name, *args):
    """Calls an entry point, converting tensors to device pointers; raises Fv2pError on a negative return code.
    (Hot: kept allocation-light — one tuple build and one foreign call.)"""
    ent = _BOUND.get(name) or _bind(name)
    fn, kinds, arrs, is_int = ent
    if len(args) != len(kinds):
        raise TypeError(f"{name} expects {len(kinds)} arguments, got {len(args)}")
    conv = [None] * len(args)
    for i, a in enumerate(args):
        k = kinds[i]
        if k == "scalar":
            conv[i] = a
        elif k == "ptr":
            conv[i] = a if (a is None or isinstance(a, int)) else a.data_ptr()
        else:
            conv[i] = arrs[i](*a)
    rc = fn(*conv)
    if is_int and rc < 0:
        raise Fv2pError(f"{name} failed ({rc}): {last_error()}")
    return rc


def ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on the current device (the C accessors: torch.cuda.current_stream()
    builds a Stream object and costs ~7 us, this is ~0.3 us — it is called once per library call)."""
    c = _torch()._C
    return c._cuda_getCurrentRawStream(c._cuda_getDevice())


_TORCH = None


def _torch():
    global _TORCH
    if _TORCH is None:
        import torch
        _TORCH = torch
    return _TORCH


class _NoGuard(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def device_guard(device):
    """`torch.cuda.device(device)` only when `device` is not already current (the context manager costs ~10 us)."""
    t = _torch()
    idx = device.index
    if idx is None or idx == t._C._cuda_getDevice():
        return _NOGUARD
    return t.cuda.device(device)


_WS = {}


def workspace(nbytes, device):
    """Grow-only per-device scratch buffer (torch caching allocator owns the memory).

    All library calls are issued on the caller's current stream, so reuse across consecutive
    calls is ordered by the stream itself."""
    torch = _torch()
    idx = device.index if device.index is not None else torch._C._cuda_getDevice()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes) * (2 if nbytes < (1 << 28) else 1), 1 << 22), dtype=torch.uint8, device=device)   # small requests double (fewer regrowths), large ones are taken as asked
        _WS[key] = buf
    return buf


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Fv2pError("fv2p ops run on the GPU only: got a CPU tensor (no CPU fallback exists)")

"""ctypes binding of libveloxseg_hip.so (C ABI declared in include/veloxseg_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
Prototypes are parsed from the public header so Python and C cannot drift apart.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Optional, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "veloxseg_hip.h")
LIB_PATH = os.path.join(_PKG, "lib", "libveloxseg_hip.so")
ABI_VERSION = 1

_CTYPE = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "unsigned long long": ctypes.c_ulonglong,
}


class VxPwaPlan(ctypes.Structure):
    """Mirror of `struct VxPwaPlan` (include/veloxseg_hip.h)."""
    _fields_ = [("grid", ctypes.c_int * 3), ("n", ctypes.c_int * 3), ("heads", ctypes.c_int), ("nb", ctypes.c_int),
                ("small", (ctypes.c_int * 3) * 4), ("nwin", (ctypes.c_int * 3) * 4), ("woff", ctypes.c_int * 4),
                ("Ntot", ctypes.c_int), ("l", ctypes.c_int)]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """-> {name: (return type, [(ctype string, arg name), ...])} for every `vx_*` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int)\s+(vx_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip()
                arglist.append((ty, mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


def _to_ctype(ty: str):
    if "*" in ty:
        return ctypes.c_void_p
    return _CTYPE[ty]


class _Lib:
    def __init__(self):
        self._dll = None
        self._fns = {}
        self.protos = parse_header()

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"veloxseg_amd: HIP library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback for the VeloxSeg hot path.")
        # the gfx950 packed-fp32 op_sel hazard (veloxseg_amd/_isa_fix.py): a library without a valid stamp -- not scanned since it was (re)built or copied in -- is
        # disassembled and checked ONCE (about 10 s) before any of its kernels can run; a hazardous library raises.  VELOXSEG_SKIP_ISA_CHECK=1 skips (debugging only).
        if os.environ.get("VELOXSEG_SKIP_ISA_CHECK") != "1":
            from . import _isa_fix
            if not _isa_fix.stamp_ok(LIB_PATH):
                _isa_fix.check_library(LIB_PATH, verbose=False)
        dll = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in self.protos.items():
            fn = getattr(dll, name)          # AttributeError here = header/library drift
            fn.restype = ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int
            fn.argtypes = [_to_ctype(t) for t, _ in args]
        v = dll.vx_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"veloxseg_amd: ABI mismatch: library {v}, python {ABI_VERSION}")
        self._dll = dll
        return dll

    def call(self, name: str, *args):
        fn = self._fns.get(name)
        if fn is None:
            dll = self.load()
            fast = fast_module()
            fn = getattr(fast, name, None) if fast is not None else None      # generated METH_FASTCALL wrapper (same library underneath)
            if fn is None:
                fn = getattr(dll, name)                                       # ctypes binding
            self._fns[name] = fn
        rc = fn(*args)
        if rc != 0:
            msg = self._dll.vx_last_error()
            raise RuntimeError(f"{name} failed (rc={rc}): {msg.decode() if msg else '?'}")


LIB = _Lib()
_FAST = [False, None]


def fast_module(reload: bool = False):
    """veloxseg_amd._vxfast (built by __graft_entry__.build from the generated csrc/_vxfast.c) or None: the ctypes path is then used;
    both are bindings of the same libveloxseg_hip.so, neither is a fallback implementation."""
    if reload:
        _FAST[0] = False
        LIB._fns.clear()
    if not _FAST[0]:
        _FAST[0] = True
        if os.environ.get("VELOXSEG_NO_FASTCALL") != "1":
            try:
                LIB.load()                      # the extension links against the library: make sure it is resident first
                from . import _vxfast
                _FAST[1] = _vxfast
            except ImportError:
                _FAST[1] = None
    return _FAST[1]


def available() -> bool:
    return os.path.exists(LIB_PATH)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """hipStream_t of PyTorch's current stream on the current device (the raw getter is ~30x cheaper than building a Stream object;
    this is called once per kernel launch)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def P(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("veloxseg_amd: tensors must live on an MI355X (cuda) device; there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"veloxseg_amd: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("veloxseg_amd: tensor must be contiguous (NCDHW)")
    return t.data_ptr()


_PROFILE = None   # when a list: (name, int-args key, start event, end event) per call -- bench.py's per-kernel timing pass


def _cpp_profiler():
    from . import functional as VF
    return VF.cpp_module()


def profile_begin():
    """time every C-ABI call with HIP events on its launch stream: the calls made from python (this module) and those made by the C++ operator
    bodies (veloxseg_amd._vxops keeps its own record list)"""
    global _PROFILE
    _PROFILE = []
    m = _cpp_profiler()
    if m is not None:
        m.profile_begin()


def profile_end():
    """-> {(name, key): [launches, total_ms]} measured with HIP events on the launch stream."""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, key, e0, e1 in rec:
        d = out.setdefault((name, key), [0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
    m = _cpp_profiler()
    if m is not None:
        for name, key, ms in m.profile_end():
            d = out.setdefault((name, tuple(key)), [0, 0.0])
            d[0] += 1
            d[1] += ms
    return out


def query(name: str, *args) -> int:
    """entries whose int return value is an answer, not a status (negative = error)"""
    LIB.load()
    fn = LIB._fns.get(name)
    if fn is None:
        fast = fast_module()
        fn = LIB._fns[name] = getattr(fast, name, None) if fast is not None and hasattr(fast, name) else getattr(LIB._dll, name)
    rc = fn(*args)
    if rc < 0:
        msg = LIB._dll.vx_last_error()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg.decode() if msg else '?'}")
    return rc


def call(name: str, *args):
    if _PROFILE is None:
        LIB.call(name, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    LIB.call(name, *args)
    e1.record()
    key = tuple(a for a in args[:-1] if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 24))
    _PROFILE.append((name, key, e0, e1))


def make_plan(grid, n, heads, small, nwin) -> VxPwaPlan:
    pl = VxPwaPlan()
    nb = len(small)
    if nb < 1 or nb > 4:
        raise ValueError(f"PWA supports 1..4 window scales, got {nb}")
    for k in range(3):
        pl.grid[k] = int(grid[k])
        pl.n[k] = int(n[k])
    pl.heads, pl.nb = int(heads), nb
    off = 0
    for i in range(nb):
        for k in range(3):
            pl.small[i][k] = int(small[i][k])
            pl.nwin[i][k] = int(nwin[i][k])
        pl.woff[i] = off
        off += int(nwin[i][0]) * int(nwin[i][1]) * int(nwin[i][2])
    pl.Ntot = off
    pl.l = int(n[0]) * int(n[1]) * int(n[2])
    return pl

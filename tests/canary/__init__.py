"""Guard-zone device allocator of the GPU test suite (see canary_alloc.cpp).  TEST INFRASTRUCTURE: nothing under vidi_amd/ imports it.

    VIDI_CANARY=1 python -m pytest tests -m gpu       every tensor gets poisoned zones, checked when it is freed and after every test
    VIDI_CANARY=2 ...                                  ... and after every C-ABI call (slow; names the call that did it)"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "canary_alloc.cpp")
LIB = os.path.join(HERE, "libcanary_alloc.so")
_handle = None


def build(force: bool = False) -> str:
    """hipcc -shared (host code only: hipMalloc / hipMemset / hipMemcpy); needs no GPU"""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        hipcc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for canary_alloc.cpp:\n" + r.stderr[-3000:])
    return LIB


def install() -> None:
    """make the guard-zone allocator torch's device allocator; must run before the first device allocation of the process"""
    global _handle
    import torch
    path = build()
    alloc = torch.cuda.memory.CUDAPluggableAllocator(path, "canary_malloc", "canary_free")
    torch.cuda.memory.change_current_allocator(alloc)
    _handle = ctypes.CDLL(path)                       # the same loaded object: shares the allocation table with torch's handle
    _handle.canary_check_all.restype = ctypes.c_long
    _handle.canary_check_all.argtypes = [ctypes.c_char_p, ctypes.c_int]
    _handle.canary_live_allocations.restype = ctypes.c_long
    _handle.canary_total_allocations.restype = ctypes.c_long


def installed() -> bool:
    return _handle is not None


def check() -> tuple:
    """(violations since start, first message, live allocations, allocations so far)"""
    buf = ctypes.create_string_buffer(512)
    n = int(_handle.canary_check_all(buf, 512))
    return n, buf.value.decode(), int(_handle.canary_live_allocations()), int(_handle.canary_total_allocations())


def reset() -> None:
    _handle.canary_reset()

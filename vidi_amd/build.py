"""Build libvidi_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/vidi_hip.h); Python binds it with ctypes (vidi_amd/hip.py)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# VIDI_BUILD_OUT: objects and the library go to that directory instead of the tree (tests/test_abi.py exercises the driver's build entry that way,
# without relinking the library underneath the handles the test session has open)
_OUT = os.environ.get("VIDI_BUILD_OUT")
OBJ = os.path.join(_OUT, "obj") if _OUT else os.path.join(CSRC, "build")
LIB = os.path.join(_OUT, "libvidi_hip.so") if _OUT else os.path.join(HERE, "libvidi_hip.so")
SOURCES = ["gemm.hip", "gemv.hip", "gemm_w4_bf16.hip", "gemm_w4_f16.hip", "gemm_w4_modes.hip", "gemm_w4_lnf.hip", "gemm_w4_patch.hip", "gemm_w4n.hip", "gemm_skinny.hip", "gemv_mfma.hip", "attn_self.hip", "attn_self_rm.hip", "attn_cross.hip", "attn_cross_rows.hip", "attn_text.hip", "rowops.hip", "elementwise.hip", "preproc.hip", "probe.hip", "capi.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_tile.h", "gemm_w4.h", "gemm_w4n.h", "gemm_w4_launch.h", "gemm_skinny.h", "gemm_skinny_api.h", "gemv_mfma_api.h", "attn_text_decode.h", os.path.join("..", "..", "include", "vidi_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]      # (+ the resource-usage remark, see _compile)
# MFMA results in architectural VGPRs: without this hipcc parks the attention accumulators in AGPRs and copies them
# to VGPRs and back around every softmax step (attn_self: 2192 v_accvgpr moves, 204 registers -> 0 moves, 150)
# -fno-honor-nans (encoder attention only): the softmax never produces a NaN (masked scores are -inf, every tile holds a valid key), and
# without the flag every fmaxf canonicalises its operands first (v_max_f32 x, x, x: 10 of the loop's ~130 vector instructions per tile):
# +1.3 % on the kernel, bit-identical (profiles/r3_ab_attn_nonans.jsonl)
EXTRA_FLAGS = {"attn_self.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "attn_self_rm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"],
               "attn_cross.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "gemv_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    for k in sorted(EXTRA_FLAGS):
        if any(p.endswith(k) for p in paths):
            h.update(" ".join(EXTRA_FLAGS[k]).encode())
    return h.hexdigest()


GEMM_FILES = ["gemm.hip", "gemm_w4_bf16.hip", "gemm_w4_f16.hip", "gemm_w4_modes.hip", "gemm_w4_lnf.hip", "gemm_w4_patch.hip", "gemm_w4n.hip", "gemm_tile.h", "gemm_w4.h", "gemm_w4n.h", "gemm_w4_launch.h", "common.h", "kernels.h"]


def source_digest(which: str = "all") -> str:
    """Short digest of the kernel sources + compile flags: identifies a build's code independently of when it was compiled.
    `gemm`: only the files the GEMM family is built from — profiles/traffic.json (PMC passes) is keyed by it (bench.py)."""
    files = GEMM_FILES if which == "gemm" else sorted(set(SOURCES) | {h for h in HEADERS if not h.startswith("..")})
    return _digest([os.path.join(CSRC, f) for f in files])[:16]


def _compile(src: str, force: bool) -> str:
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    dig = _digest([os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS])
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    # -Rpass-analysis=kernel-resource-usage: the per-kernel register / scratch report is kept next to the object
    # (csrc/build/<src>.resources.txt); tests/test_build_resources.py fails if a hot kernel spills (a spill in the GEMM body
    # once cost 11 % end to end without any functional symptom)
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    with open(obj.replace(".o", ".resources.txt"), "w") as f:
        f.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[vidi_amd.build] {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

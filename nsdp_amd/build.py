"""Builds libnsdp_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch extension
machinery, no hipify: plain `hipcc --offload-arch=gfx950` per translation unit, then one shared link.

    python -m nsdp_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "libnsdp_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "nsdp_hip.h")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -munsafe-fp-atomics: fp32 atomicAdd lowers to the hardware global_atomic_add_f32 instead of a CAS loop
# (all buffers are ordinary coarse-grained device allocations)
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-munsafe-fp-atomics"]
# geometry kernels must keep one rounding per fp32 op (bit-exact distances): contraction off
EXACT = ["-ffp-contract=off"]
FAST = ["-ffp-contract=fast"]
PER_FILE = {
    "fps.hip": EXACT,
    "knn.hip": EXACT,
    "pointnet2_ops.hip": EXACT,
    # torch's single-tensor Adam rounds once per operation: no fused multiply-adds in the optimizer kernel
    "adam.hip": EXACT,
    # -O3 turns the uniform base pointers of this file's hand-placed `global_load ... s[base]` asm operands into
    # VGPR copies (rejected by the assembler); -O2 keeps them scalar
    "wgrad_bf16x3.hip": FAST + ["-O2", "-fno-slp-vectorize"],
    # packed fp32 VALU (v_pk_add_f32 / v_pk_mul_f32, what the SLP vectorizer makes of the activation split's adjacent scalar
    # subtractions) costs more issue time beside MFMAs than the two scalar operations it replaces (MI355X_MICROARCH.md): without
    # it the forward / dX GEMMs run 0.5-3 % faster alone and the B = 32 step 0.26 ms (three interleaved rounds), bit-identical
    "gemm_bf16x3.hip": FAST + ["-fno-slp-vectorize"],
    "gemm_bf16x3_g16.hip": FAST + ["-fno-slp-vectorize"],      # (the same kernel template, x3_kernel.h: its G16-layout instantiations)
}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    # (this file too: the compiler flags live here)
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [HEADER, os.path.abspath(__file__)]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime())):
        return obj, False
    cmd = [HIPCC] + COMMON + PER_FILE.get(src, FAST) + ["-c", path, "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, force), sources()))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(SO):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
        if verbose:
            print("linked", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

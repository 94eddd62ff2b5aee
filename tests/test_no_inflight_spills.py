"""The split-precision / fused kernels issue some loads by hand (`asm volatile` global_load / ds_read whose completion
the compiler does not track).  That is only sound if the compiler never spills or copies a destination register
while the load is in flight -- i.e. no scratch store and no VGPR->AGPR parking inside the MFMA regions of those
kernels.  This test compiles the three translation units to gfx950 assembly and checks exactly that, per kernel
variant (the same audit that was used while writing them)."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILES = {  # translation unit -> (optimisation flag used by nsdp_amd/build.py, kernel-name regex)
    "gemm_bf16x3.hip": ("-O3", r"linear_bf16x3_kernel"),
    "wgrad_bf16x3.hip": ("-O2", r"wgrad_bf16x3_kernel"),
    "decoder_fused.hip": ("-O3", r"decoder_fused_fwd_kernel"),
}


def _asm(item):
    src, (opt, _) = item
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", opt, "-std=c++17", "-ffp-contract=fast", "-munsafe-fp-atomics",
                          "-S", "--cuda-device-only", "-o", "-", os.path.join(ROOT, "nsdp_amd", "csrc", src)],
                         capture_output=True, text=True, check=True)
    return src, out.stdout


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")
def test_nothing_is_spilled_inside_the_mfma_regions():
    with ThreadPoolExecutor(max_workers=3) as ex:
        listings = dict(ex.map(_asm, FILES.items()))
    checked = 0
    for src, text in listings.items():
        lines = text.split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + FILES[src][1] + r"\w*:", l)]
        assert starts, src
        for a, b in zip(starts, starts[1:] + [len(lines)]):
            body = lines[a:b]
            if any("s_endpgm" in l for l in body):
                body = body[:max(i for i, l in enumerate(body) if "s_endpgm" in l) + 1]
            mf = [i for i, l in enumerate(body) if "v_mfma" in l]
            region = body[mf[0]:mf[-1]]
            if src == "decoder_fused.hip":
                # the whole kernel is one MFMA chain whose accumulators are initialised from activation vectors
                # (legitimate VGPR -> AGPR moves); what must not exist at all is scratch
                bad = [l.strip() for l in body if "scratch_" in l]
            else:
                bad = [l.strip() for l in region if "scratch_store" in l or "v_accvgpr_write" in l]
            assert not bad, (src, lines[a][:80], bad[:3])
            checked += 1
    assert checked >= 20

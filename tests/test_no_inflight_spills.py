"""The split-precision / fused kernels issue some loads by hand (`asm volatile` global_load / ds_read whose completion
the compiler does not track).  That is only sound if the compiler never spills or copies a destination register
while the load is in flight -- i.e. no scratch store and no VGPR->AGPR parking of a load destination inside the
MFMA regions of those kernels.  This test compiles the three translation units to gfx950 assembly and checks exactly that, per kernel
variant (the same audit that was used while writing them).

A second, performance-only rule for the weight-gradient kernel: no scratch at all.  A reload in its block loop has
to wait for every global load issued before it (vmcnt retires in order), i.e. for the whole prefetch of the block
after next; the 13x13 variant once carried such reloads and ran 1.3x slower for it."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
import sys
sys.path.insert(0, ROOT)
from nsdp_amd import build as nsdp_build  # noqa: E402  (the flags under test are the ones the library is built with)

FILES = {  # translation unit -> kernel-name regex
    "gemm_bf16x3.hip": r"linear_bf16x3_kernel",
    "gemm_bf16x3_g16.hip": r"linear_bf16x3_kernel",
    "wgrad_bf16x3.hip": r"wgrad_bf16x3_(?:rows_)?kernel",
    "decoder_fused.hip": r"decoder_fused_fwd_kernel",
}
NO_SCRATCH = ("decoder_fused.hip", "wgrad_bf16x3.hip")


def _defines(line, reg):
    """Does this instruction write VGPR `reg` (first operand vN or v[a:b])?"""
    parts = line.strip().split(None, 1)
    if len(parts) < 2 or parts[0].startswith(("s_", ";", ".")):
        return False
    dst = parts[1].split(",")[0].strip()
    m = re.fullmatch(r"v(\d+)", dst)
    if m:
        return int(m.group(1)) == reg
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", dst)
    return bool(m) and int(m.group(1)) <= reg <= int(m.group(2))


def _parked_loads(body, lo, hi, hand_issued_only=False):
    """v_accvgpr_write instructions in body[lo:hi] whose source VGPR was last written by a hand-issued load
    (global_load / ds_read): a register the compiler must not copy before the matching s_waitcnt.  Accumulator
    shuffles (v_accvgpr_read -> v_accvgpr_write) are the compiler's own business and are fine.
    ``hand_issued_only``: count only loads inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND) -- outside the MFMA region
    the compiler's OWN loads may be parked (it waits for them first); a parked asm load is how the staged epilogue of
    round 4 once used bias values that had not arrived."""
    in_asm, flag = False, []
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        flag.append(in_asm)
    bad = []
    for i in range(lo, hi):
        m = re.match(r"\s*v_accvgpr_write_b32 a\d+, v(\d+)", body[i])
        if not m:
            continue
        reg = int(m.group(1))
        for j in range(i - 1, -1, -1):
            if _defines(body[j], reg):
                if body[j].strip().startswith(("global_load", "ds_read")) and (flag[j] or not hand_issued_only):
                    # (parked AFTER its wait is fine: look for the s_waitcnt between the load and the copy)
                    waited = any("s_waitcnt" in body[k] and ("lgkmcnt(0)" in body[k] or "vmcnt(0)" in body[k]) for k in range(j + 1, i))
                    if not (hand_issued_only and waited):
                        bad.append(body[i].strip() + "   <- " + body[j].strip())
                break
    return bad


def _asm(item):
    src, _ = item
    flags = [f for f in nsdp_build.COMMON if f not in ("-fPIC", "-Wall")] + nsdp_build.PER_FILE.get(src, nsdp_build.FAST)
    out = subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", "-",
                                            os.path.join(ROOT, "nsdp_amd", "csrc", src)],
                         capture_output=True, text=True, check=True)
    return src, out.stdout


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")
def test_nothing_is_spilled_inside_the_mfma_regions():
    with ThreadPoolExecutor(max_workers=4) as ex:
        listings = dict(ex.map(_asm, FILES.items()))
    checked = 0
    for src, text in listings.items():
        lines = text.split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + FILES[src] + r"\w*:", l)]
        assert starts, src
        for a, b in zip(starts, starts[1:] + [len(lines)]):
            body = lines[a:b]
            ends = [i for i, l in enumerate(body) if l.startswith(".Lfunc_end")]     # (kernels the regex skips may follow)
            if ends:
                body = body[:ends[0]]
            mf = [i for i, l in enumerate(body) if "v_mfma" in l]
            region = body[mf[0]:mf[-1]]
            bad = []
            if src in NO_SCRATCH:
                bad += [l.strip() for l in body if "scratch_" in l]
            if src != "decoder_fused.hip":
                # (the decoder is one MFMA chain whose accumulators are initialised from activation vectors:
                # legitimate VGPR -> AGPR moves)
                bad += [l.strip() for l in region if "scratch_store" in l]
                bad += _parked_loads(body, mf[0], mf[-1])
                # ... and hand-issued loads anywhere else in the kernel (prologue, epilogue)
                bad += _parked_loads(body, 0, mf[0], hand_issued_only=True) + _parked_loads(body, mf[-1], len(body), hand_issued_only=True)
            assert not bad, (src, lines[a][:80], bad[:3])
            checked += 1
    assert checked >= 20


def _regs(line):
    """All VGPR numbers an instruction line mentions (vN and v[a:b])."""
    used = {int(x) for x in re.findall(r"\bv(\d+)\b", line)}
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
        used.update(range(int(a), int(b) + 1))
    return used


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")
def test_bf16_linear_never_touches_a_register_whose_hand_issued_load_is_in_flight():
    """csrc/gemm_bf16.hip prefetches the next tile with `asm volatile` global_load_dwordx4 and makes the registers
    usable with `s_waitcnt vmcnt(0)` asm statements.  Between the two NOTHING may read or write those registers -- not a
    spill, not a v_mov the register allocator inserts at a loop back edge (that is how a register-ring version of the
    weight-gradient kernel produced NaNs: copies of registers whose data had not arrived yet), not an AGPR parking."""
    _, text = _asm(("gemm_bf16.hip", None))
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*linear_bf16_kernel\w*:", l)]
    assert len(starts) >= 16
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        body = lines[a:b]
        ends = [i for i, l in enumerate(body) if l.startswith(".Lfunc_end")]
        if ends:
            body = body[:ends[0]]
        assert not any("scratch_" in l for l in body), lines[a][:80]
        inflight, in_asm, bad = set(), False, []
        for l in body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            if in_asm and t.startswith("global_load_dwordx4"):
                inflight |= _regs(t.split(",")[0])
                continue
            if "s_waitcnt" in t and "vmcnt(0)" in t:
                inflight.clear()
                continue
            # sources only: a k block beyond K is zeroed (a write) in the branch that does not load it, and this scan
            # follows text order, not control flow
            ops = t.split(None, 1)
            srcs = ops[1].split(",", 1)[1] if len(ops) > 1 and "," in ops[1] else ""
            hit = _regs(srcs) & inflight
            dst = ops[1].split(",", 1)[0] if len(ops) > 1 else ""
            if not ops[0].startswith(("s_", "global_store", "ds_write", "buffer_store")):
                inflight -= _regs(dst)          # redefined (the zeroing of an unloaded k block): no longer a load destination
            if hit:
                bad.append(t + "   <- in flight: v" + ", v".join(map(str, sorted(hit))))
        assert not bad, (lines[a][:80], bad[:3])

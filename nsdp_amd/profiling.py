"""Per-kernel timing of the hand-written HIP kernels with HIP events recorded on the launch stream
(C side: nsdp_prof_* in csrc/api.hip), used by bench.py to report the roofline of the dominant kernel."""
from __future__ import annotations

import ctypes

from . import _lib

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0
PEAK_F32_MFMA_TFLOPS = 157.3
# bf16x3 kernels: every fp32 multiply-add is 6 bf16 MFMA products (3-way split of both operands, csrc/gemm_bf16x3.hip);
# the compute roofline for their ALGORITHMIC flops is the dense bf16 MFMA peak / 6
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def _mfma_peak(name):
    if "bf16x3" in name:
        return PEAK_BF16X3_TFLOPS
    if "bf16" in name:                      # bf16-storage kernels: one bf16 MFMA product per multiply-add
        return PEAK_BF16_MFMA_TFLOPS
    return PEAK_F32_MFMA_TFLOPS

_active = False


def available() -> bool:
    return hasattr(_lib.lib(), "nsdp_prof_enable")


def start(only=None):
    """Start timing; ``only`` = iterable of kernel names (nsdp_prof_name) to restrict the HIP events to -- an event
    pair per launch is not free (it keeps consecutive kernels from overlapping head-to-tail)."""
    global _active
    if not available():
        return
    lib = _lib.lib()
    if only is None:
        lib.nsdp_prof_enable(1)
    else:
        lib.nsdp_prof_name.restype = ctypes.c_char_p
        names = [lib.nsdp_prof_name(k).decode() for k in range(lib.nsdp_prof_num_kinds())]
        mask = 0
        for nm in only:
            mask |= 1 << names.index(nm)
        lib.nsdp_prof_enable_kinds(ctypes.c_uint(mask))
    _active = True


def stop():
    """Returns {kernel: {'launches', 'ms', 'flops', 'bytes'}} accumulated since start()."""
    global _active
    if not _active:
        return {}
    lib = _lib.lib()
    lib.nsdp_prof_enable(0)
    _active = False
    n = lib.nsdp_prof_num_kinds()
    lib.nsdp_prof_name.restype = ctypes.c_char_p
    out = {}
    for kind in range(n):
        cnt = ctypes.c_longlong(0)
        ms = ctypes.c_double(0)
        flops = ctypes.c_double(0)
        nbytes = ctypes.c_double(0)
        lib.nsdp_prof_collect(kind, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(flops), ctypes.byref(nbytes))
        if cnt.value:
            out[lib.nsdp_prof_name(kind).decode()] = {"launches": cnt.value, "ms": ms.value,
                                                      "flops": flops.value, "bytes": nbytes.value}
    return out


def summary(prof):
    if not prof:
        return None
    return {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                "tflops": round(v["flops"] / (v["ms"] * 1e9), 2) if v["ms"] > 0 and v["flops"] else None,
                "gbps": round(v["bytes"] / (v["ms"] * 1e6), 1) if v["ms"] > 0 and v["bytes"] else None}
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}


def roofline(prof, prof_isolated=None, pmc_matches=True, pmc_suffix="", replayed=None):
    """Roofline object of the dominant hand-written kernel class.

    Two measurements exist for every kernel, both taken live with HIP events on the launch stream: inside the timed
    region (``prof``), where the weight-gradient kernels run on a second stream and an event pair then also spans the
    time a kernel waits for CUs held by the other stream, and in an extra pass with that overlap switched off
    (``prof_isolated``), where the pair brackets the kernel alone.  The dominant kernel is chosen and
    ``achieved`` / ``frac`` / ``avg_launch_ms`` are reported from the isolated pass (a kernel's own duration is what a
    roofline fraction is about; `rocprofv3 --kernel-trace --stats` of ``NSDP_WGRAD_STREAM=0 python bench.py`` is the
    matching profile); ``in_step`` repeats the figures as timed inside the overlapped region."""
    if not prof and not prof_isolated:
        return None
    prof = prof or {}      # (a graph replay carries no events: only the isolated pass exists then)
    base = prof_isolated if prof_isolated else prof
    name, v = max(base.items(), key=lambda kv: kv[1]["ms"])
    out = _roofline_one(name, v)
    out["measured"] = "isolated pass (no cross-stream overlap)" if prof_isolated else "timed region"
    # the committed PMC passes are of the default workload (B = 32 forward.yaml train step) only
    out.update(_pmc_traffic(name, pmc_suffix) if pmc_matches else {"traffic": None})
    if out.get("traffic"):
        # how busy HBM is while this kernel runs: the bytes the memory-side counters saw (masks, residuals, gathered tables,
        # weight re-reads and store read-modify-writes included -- everything `algorithmic_bytes` leaves out) per second of the
        # kernel's own duration, against the 8 TB/s peak.  `frac` is the contract's figure; this one says how far from the HBM
        # ceiling the launch really is.
        out["frac_hbm_busy_by_pmc"] = round(out["traffic"] / (out["avg_launch_ms"] * 1e6) / PEAK_HBM_GBPS, 4)
    if prof_isolated and name in prof:
        ins = _roofline_one(name, prof[name])
        out["in_step"] = {k: ins[k] for k in ("achieved", "frac", "launches", "avg_launch_ms")}
    # ``replayed`` = {"launches_per_step", "ms_per_step"} of this kernel class inside a REPLAY of the captured step
    # (GraphedStep.timed_replay): the same algorithmic bytes per launch against the duration a launch has beside the other
    # stream's kernels -- what the timed region actually runs (the isolated `frac` is the kernel's own).
    if replayed and replayed.get("launches_per_step"):
        per = replayed["ms_per_step"] / replayed["launches_per_step"]
        iso_per_step = v["launches"] / 2.0 if prof_isolated else None        # (the isolated pass times two steps)
        # (the class sits AT the ridge of its roofline -- 50-52 flop/B against 52: whichever side `bound` names, the replayed
        # figure is the same work against the longer launch)
        if out["bound"] == "hbm":
            rate, peak_r = out["algorithmic_bytes"] / (per * 1e6), PEAK_HBM_GBPS
        else:
            rate, peak_r = v["flops"] / v["launches"] / (per * 1e9), out["peak"]
        out["replayed"] = {"launches_per_step": replayed["launches_per_step"], "avg_launch_ms": round(per, 4),
                           "class_ms_per_step": replayed["ms_per_step"],
                           "same_launch_count_as_isolated": (iso_per_step == replayed["launches_per_step"]) if iso_per_step else None}
        out["replayed"]["achieved"] = round(rate, 1)
        out["frac_replayed"] = out["replayed"]["frac"] = round(rate / peak_r, 4)
    elif replayed:
        out["replayed"] = replayed
    return out


def _pmc_traffic(name, suffix=""):
    """HBM bytes per launch of `name` from the committed rocprofv3 PMC passes of this same command
    (profiles/rN_pmc_hbm.json of the newest round N: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950).  None when the file is absent."""
    import json
    import os
    prof_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    # (the newest round's file: profiles/rN_pmc_hbm{suffix}.json)
    files = [f"r{n}_pmc_hbm{suffix}.json" for n in range(9, 0, -1)]
    path = next((os.path.join(prof_dir, f) for f in files if os.path.exists(os.path.join(prof_dir, f))),
                os.path.join(prof_dir, files[0]))
    try:
        with open(path) as f:
            pmc = json.load(f)
        stem = name.replace("_kernels", "").replace("_kernel", "")
        fetch = [v for k, v in pmc.get("FETCH_SIZE", {}).items() if k.startswith(stem)]
        write = [v for k, v in pmc.get("WRITE_SIZE", {}).items() if k.startswith(stem)]
        n = sum(v[0] for v in fetch)
        if not n:
            return {"traffic": None}
        total = (2.0 * sum(v[1] for v in fetch) + sum(v[1] for v in write)) * 1024.0
        out = {"traffic": round(total / n), "traffic_unit": f"bytes/launch (PMC, profiles/{os.path.basename(path)}; FETCH_SIZE x 2 as the guide prescribes)",
               "traffic_writes": round(sum(v[1] for v in write) * 1024.0 / n)}
        # the same passes with FETCH_SIZE CALIBRATED on the launches whose reads are known exactly, and the excess over SURVEY 8d's
        # X + Y formula attributed by operand (tools/pmc_attribution.py -> profiles/rN_pmc_attribution.json; the bf16x3 class only)
        att = path.replace("_pmc_hbm" + suffix + ".json", "_pmc_attribution.json")
        if not suffix and os.path.exists(att):
            with open(att) as f:
                a = json.load(f)
            if name.startswith(a.get("kernel", "?").replace("_kernel", "")):
                out["traffic_calibrated"] = round(a["read_bytes_per_launch"] + a["write_bytes_per_launch"])
                out["traffic_calibrated_over_algorithmic_by_operand"] = round(
                    (a["read_bytes_per_launch"] + a["write_bytes_per_launch"])
                    / (a["algorithmic_read_bytes_per_launch"] + a["algorithmic_write_bytes_per_launch"]), 3)
                out["traffic_over_survey_8d"] = {k: round(v, 3) for k, v in a["over_survey_8d"].items()}
        return out
    except Exception:
        return {"traffic": None}


def _roofline_one(name, v):
    per_launch_ms = v["ms"] / v["launches"]
    flops_per_launch = v["flops"] / v["launches"]
    bytes_per_launch = v["bytes"] / v["launches"]
    intensity = flops_per_launch / max(bytes_per_launch, 1.0)
    peak = _mfma_peak(name)
    both = {}
    if flops_per_launch > 0 and bytes_per_launch > 0:
        # both sides of the roofline, whichever bounds: the 200-wide bf16x3 layers sit AT the ridge (50 flop/B against 52)
        both = {"frac_hbm": round(bytes_per_launch / (per_launch_ms * 1e6) / PEAK_HBM_GBPS, 4),
                "frac_mfma": round(flops_per_launch / (per_launch_ms * 1e9) / peak, 4),
                "mfma_peak_tflops": round(peak, 1), "flop_per_byte": round(intensity, 1)}
    if intensity > peak * 1e12 / (PEAK_HBM_GBPS * 1e9):
        achieved = flops_per_launch / (per_launch_ms * 1e9)
        return {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                "algorithmic_bytes": round(bytes_per_launch), "launches": v["launches"],
                "avg_launch_ms": round(per_launch_ms, 4), **both}
    achieved = bytes_per_launch / (per_launch_ms * 1e6)
    return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": round(achieved / PEAK_HBM_GBPS, 4), "traffic": None,
            "algorithmic_bytes": round(bytes_per_launch), "launches": v["launches"],
            "avg_launch_ms": round(per_launch_ms, 4), **both}

#!/usr/bin/env python
"""HBM traffic of the bf16x3 GEMM class, attributed: joins the PMC passes (profiles/<tag>_pmc_hbm.json: rocprofv3 --pmc FETCH_SIZE
/ --pmc WRITE_SIZE per kernel template) with the algorithmic bytes per launch BY OPERAND (profiles/<tag>_x3_operands.json,
written by tools/profile_linear_shapes.py --json on the same binary).  FETCH_SIZE on gfx950 under-reports by a factor that
depends on the access pattern (2.0 for whole-KiB row reads; less for the 64-byte segments of the row-major activation DMA), so
it is CALIBRATED here on the launches whose reads are known exactly -- the plain forms, which read X and the weight planes and
nothing else -- per access pattern (row-major / G16), and that factor is applied to the forms that also stream a mask or a
residual.  WRITE_SIZE needs no correction.

    python tools/pmc_attribution.py r6      ->  profiles/r6_pmc_attribution.md
"""
import json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
pmc = json.load(open(os.path.join(root, f"{tag}_pmc_hbm.json")))
ops = json.load(open(os.path.join(root, f"{tag}_x3_operands.json")))["templates"]
KIB = 1024.0


def pm(kind, name):
    v = pmc.get(kind, {}).get(name)
    return (v[0], v[1] * KIB) if v else (0, 0.0)      # launches, bytes


rows = []
for name, o in ops.items():
    nf, fetch = pm("FETCH_SIZE", name)
    nw, write = pm("WRITE_SIZE", name)
    if not nf and not nw:
        continue
    n = o["launches"]
    reads = o["X"] + o["mask"] + o["residual"] + o["other"]
    args = [a.strip() for a in re.search(r"<(.*)>", name).group(1).split(",")]
    lay = int(args[8]) if len(args) > 8 else 0
    plain = o["mask"] == 0 and o["residual"] == 0 and int(args[2]) in (0, 2) and int(args[6]) == 0 and int(args[7]) == 0
    rows.append({"name": name, "n": n, "alg_read": reads / n, "alg_write": o["Y"] / n, "X": o["X"] / n, "mask": o["mask"] / n,
                 "residual": o["residual"] / n, "fetch": fetch / max(nf, 1), "write": write / max(nw, 1), "plain": plain, "x_g16": bool(lay & 1)})
# calibration: bytes really read / bytes FETCH_SIZE reports, on the plain forms, per activation access pattern
cal = {}
for pat in (False, True):
    sel = [r for r in rows if r["plain"] and r["x_g16"] == pat and r["fetch"] > 0]
    if sel:
        cal[pat] = sum(r["alg_read"] * r["n"] for r in sel) / sum(r["fetch"] * r["n"] for r in sel)
cal.setdefault(False, 2.0)
cal.setdefault(True, cal[False])
out = [f"# HBM traffic of `linear_bf16x3_kernel` by form ({tag}): PMC against algorithmic bytes, per launch\n",
       "FETCH_SIZE calibration on the plain forms (reads = X + weight planes, exactly): "
       f"row-major activations x{cal[False]:.3f}, G16 activations x{cal[True]:.3f} (the guide's generic correction is x2: exact for "
       "whole-KiB row reads, an OVER-correction for this kernel's DMA pattern).  `read` = FETCH_SIZE x that factor; WRITE_SIZE as reported.\n",
       "| kernel template | launches/step | alg. read MB (X + mask + residual) | PMC read MB | read / alg. | alg. write MB | PMC write MB | write / alg. |",
       "|---|---:|---:|---:|---:|---:|---:|---:|"]
tot = {"ar": 0.0, "pr": 0.0, "aw": 0.0, "pw": 0.0, "n": 0.0, "mask": 0.0, "res": 0.0, "x": 0.0}
for r in sorted(rows, key=lambda r: -(r["fetch"] + r["write"]) * r["n"]):
    pr = r["fetch"] * cal[r["x_g16"]]
    out.append(f"| `{r['name']}` | {r['n']:.1f} | {r['alg_read'] / 1e6:.1f} ({r['X'] / 1e6:.0f} + {r['mask'] / 1e6:.0f} + {r['residual'] / 1e6:.0f}) | "
               f"{pr / 1e6:.1f} | {pr / max(r['alg_read'], 1):.2f} | {r['alg_write'] / 1e6:.1f} | {r['write'] / 1e6:.1f} | "
               f"{(r['write'] / r['alg_write']) if r['alg_write'] else float('nan'):.2f} |")
    for k, v in (("ar", r["alg_read"]), ("pr", pr), ("aw", r["alg_write"]), ("pw", r["write"]), ("mask", r["mask"]), ("res", r["residual"]), ("x", r["X"])):
        tot[k] += v * r["n"]
    tot["n"] += r["n"]
surv = tot["x"] + tot["aw"]      # SURVEY 8d's formula counts X, Y (and W) only
out += ["", f"Class totals per step ({tot['n']:.0f} launches): algorithmic {1e-9 * (tot['ar'] + tot['aw']):.2f} GB (reads {1e-9 * tot['ar']:.2f} = X {1e-9 * tot['x']:.2f} + "
        f"masks {1e-9 * tot['mask']:.2f} + residuals / output masks {1e-9 * tot['res']:.2f} + weights; writes {1e-9 * tot['aw']:.2f}); "
        f"PMC, calibrated: reads {1e-9 * tot['pr']:.2f} GB, writes {1e-9 * tot['pw']:.2f} GB = {(tot['pr'] + tot['pw']) / (tot['ar'] + tot['aw']):.3f} x the algorithmic bytes "
        f"by operand, {(tot['pr'] + tot['pw']) / surv:.3f} x SURVEY 8d's X + Y formula (of which masks +{tot['mask'] / surv:.3f}, residuals +{tot['res'] / surv:.3f}, "
        f"write amplification +{(tot['pw'] - tot['aw']) / surv:.3f}, read amplification +{(tot['pr'] - tot['ar']) / surv:.3f})."]
open(os.path.join(root, f"{tag}_pmc_attribution.md"), "w").write("\n".join(out) + "\n")
# the class figure bench.py's roofline quotes next to the guide's x2 correction (nsdp_amd/profiling.py::_pmc_traffic)
json.dump({"kernel": "linear_bf16x3_kernel", "launches_per_step": tot["n"], "fetch_factor_row_major": cal[False], "fetch_factor_g16": cal[True],
           "read_bytes_per_launch": tot["pr"] / tot["n"], "write_bytes_per_launch": tot["pw"] / tot["n"],
           "algorithmic_read_bytes_per_launch": tot["ar"] / tot["n"], "algorithmic_write_bytes_per_launch": tot["aw"] / tot["n"],
           "survey_8d_bytes_per_launch": surv / tot["n"],
           "over_survey_8d": {"masks": tot["mask"] / surv, "residuals_and_output_masks": tot["res"] / surv,
                              "write_amplification": (tot["pw"] - tot["aw"]) / surv, "read_amplification": (tot["pr"] - tot["ar"]) / surv}},
          open(os.path.join(root, f"{tag}_pmc_attribution.json"), "w"), indent=1)
print("\n".join(out[-2:]))

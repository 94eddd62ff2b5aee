#!/usr/bin/env python
"""Condenses rocprofv3 CSV output (gpurun_out/) into the small tracked summaries under profiles/.

    python tools/summarize_profile.py <round-tag>   # e.g. r1
"""
import csv
import collections
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:90]


for suffix, title in (("", "python bench.py --eager --reps 1 --steps 5 --warmup 2 --no-cpu-baseline (the eager launcher: the same kernels as the graph replay, which the tracer does not need to see captured)"),
                      ("_iso", "NSDP_WGRAD_STREAM=0 python bench.py --eager --reps 1 --steps 5 --warmup 2 --no-cpu-baseline "
                               "(weight gradients on the main stream: every duration is the kernel alone -- the profile "
                               "that matches bench.py's `roofline`)"),
                      ("_replay", "python bench.py --reps 1 --steps 5 --warmup 2 --no-cpu-baseline (the DEFAULT launcher: the captured "
                                  "step replayed by the multi-stream graph executor, traced as it runs -- durations include the "
                                  "time a kernel shares the chip with the other stream's kernel)"),
                      ("_bf16", "NSDP_WGRAD_STREAM=0 python bench.py --dtype bf16 --workload arbitrary_train --steps 5 "
                                "--warmup 2 --no-cpu-baseline (BASELINE config 3: FlowArbitrary, bf16 storage)")):
    stats = os.path.join(root, "gpurun_out", f"prof_{tag}{suffix}", f"{tag}{suffix}_kernel_stats.csv")
    if not os.path.exists(stats):
        continue
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(out_dir, f"{tag}_kernel_stats{suffix}.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats ({tag}): {title}\n\n")
        # the decoder's kNN runs exactly once per train step: its call count is the number of steps in the trace
        # (set-up steps + warm-up + timed + the isolated roofline pass of bench.py)
        nsteps = next((int(r["Calls"]) for r in rows if "fps_reg_kernel<256" in r["Name"]), 0)
        if suffix == "_bf16":
            nsteps //= 3          # FlowArbitrary runs three encoder passes per step
        per = f" = {total/1e6/nsteps:.1f} ms of kernel time per step" if nsteps else ""
        f.write(f"total kernel time {total/1e6:.1f} ms over {nsteps} train steps{per}, B=32 shapes\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:50]:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                    f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    with open(os.path.join(out_dir, f"{tag}_kernel_stats{suffix}.csv"), "w") as f:   # keep the raw stats CSV too (small)
        f.write(open(stats).read())

for pmc_suffix, pmc_what in (("", "the default workload"), ("_bf16", "--dtype bf16")):
    pmc = {}
    for key, sub, pre in (("FETCH_SIZE", "pmc_fetch" + pmc_suffix, "f"), ("WRITE_SIZE", "pmc_write" + pmc_suffix, "w")):
        path = os.path.join(root, "gpurun_out", sub, f"{pre}_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != key:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] = max(a[2], float(r["Counter_Value"]))
        pmc[key] = agg
    if pmc:
        names = sorted(set().union(*[set(v) for v in pmc.values()]),
                       key=lambda n: -sum(pmc[k][n][1] for k in pmc if n in pmc[k]))
        with open(os.path.join(out_dir, f"{tag}_pmc_hbm{pmc_suffix}.md"), "w") as f:
            f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 (B = 32, {pmc_what})\n\n")
            f.write("Counter unit = KiB as reported.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x\n"
                    "(MI355X_MICROARCH.md, HBM section): the `fetch x2` column applies that correction.\n\n")
            f.write("| kernel | launches | FETCH_SIZE sum KiB | fetch x2 MiB | WRITE_SIZE sum KiB | max launch fetch KiB | max launch write KiB |\n|---|---:|---:|---:|---:|---:|---:|\n")
            for n in names[:40]:
                fe = pmc.get("FETCH_SIZE", {}).get(n, [0, 0, 0])
                wr = pmc.get("WRITE_SIZE", {}).get(n, [0, 0, 0])
                f.write(f"| `{n}` | {fe[0] or wr[0]} | {fe[1]:.0f} | {2*fe[1]/1024:.1f} | {wr[1]:.0f} | {fe[2]:.0f} | {wr[2]:.0f} |\n")
        json.dump({k: {n: v for n, v in agg.items()} for k, agg in pmc.items()},
                  open(os.path.join(out_dir, f"{tag}_pmc_hbm{pmc_suffix}.json"), "w"), indent=0)
# SQ counters (one pass, NSDP_WGRAD_STREAM=0 so that counters are per kernel): matrix-pipe busy fraction and wait split
sq_path = os.path.join(root, "gpurun_out", "pmc_sq", "s_counter_collection.csv")
if os.path.exists(sq_path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(sq_path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            launches[k] += 1
    rows_sq = sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))
    with open(os.path.join(out_dir, f"{tag}_pmc_mfma.md"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                "GRBM_GUI_ACTIVE --kernel-trace (one pass, no other trace domains): NSDP_WGRAD_STREAM=0 python bench.py "
                "--steps 1 --warmup 1 --no-cpu-baseline, B=32\n\n")
        f.write("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs): fraction of all SIMD cycles of the "
                "kernel's lifetime in which the matrix pipe is busy.  A bf16 16x16x32 MFMA is busy 16 cycles and an fp32 "
                "16x16x4 MFMA 32 cycles, so this is the fraction of the dense MFMA peak of the kernel's data type at the "
                "running clock (for the bf16x3 kernels: of 2.5 PFLOP/s of bf16 products, six per fp32 multiply-add).\n"
                "wait_mem = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt), wait_issue = SQ_WAIT_INST_ANY / "
                "SQ_WAVE_CYCLES, active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.\n\n")
        f.write("| kernel | launches | GPU-busy Mcycles | MFMA busy | wait_mem | wait_issue | active |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, v in rows_sq[:32]:
            gui = v.get("GRBM_GUI_ACTIVE", 0.0)
            wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui / 8.0 * 1024.0, 1.0)
            f.write(f"| `{k}` | {launches[k]} | {gui/1e6:.1f} | {100*busy:.1f} % | {100*v.get('SQ_WAIT_ANY',0)/wc:.0f} % | "
                    f"{100*v.get('SQ_WAIT_INST_ANY',0)/wc:.0f} % | {100*v.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f} % |\n")
for name in ("bench_default.json", "bench_b8.json", "bench_arbitrary.json", "bench_dense_inference.json",
             "bench_forward_eval.json", "bench_forward_bf16.json", "bench_arbitrary_bf16.json", "bench_b8_bf16.json",
             "bench_2ranks_gloo.json", "bench_default_eager.json", "bench_b8_eager.json", "bench_forward_bf16_eager.json",
             "bench_arbitrary_bf16_eager.json", "bench_force_reducer_nccl.json", "bench_arbitrary_bf16_net1f32.json",
             "bench_force_reducer_nccl_overlap.json", "bench_arbitrary_bf16_net1dec32.json", "bench_8ranks_gloo.json",
             "linear_shapes.txt", "x3_operands.json"):
    src = os.path.join(root, "gpurun_out", f"{tag}_{name}")       # written by tools/profile_round.sh
    if os.path.exists(src) and open(src).read().strip():
        open(os.path.join(out_dir, f"{tag}_{name}"), "w").write(open(src).read())
print("wrote", sorted(os.listdir(out_dir)))

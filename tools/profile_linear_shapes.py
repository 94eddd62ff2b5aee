"""Which native calls carry the train step?  Every entry point `include/nsdp_hip.h` declares is wrapped at the ctypes boundary
(the list is PARSED FROM THE HEADER: a new entry point cannot escape), one B = 32 forward.yaml train step runs with a
synchronisation behind every call, and the library's own per-launch accounting (csrc/prof.h: HIP events on the launch stream,
flops and ALGORITHMIC bytes as the C side states them in its prof::Scope lines) is read back per call.  Output:

  * per kernel class (the names of bench.py's `kernels` / `roofline`): launches / step, algorithmic GB / step, isolated ms / step
    -- the `linear_bf16x3_kernel` line must equal the bench line's roofline.launches / 2 and launches x algorithmic_bytes;
  * per (entry point, integer arguments, operands present): calls / step, launches / call, us / call, TFLOP/s, algorithmic MB / call.

    NSDP_WGRAD_STREAM=0 python tools/profile_linear_shapes.py [--batch 32] [--workload forward|arbitrary] [--all-classes]
"""
import argparse, collections, ctypes, os, re, sys
os.environ.setdefault("NSDP_WGRAD_STREAM", "0")      # weight gradients on the main stream: every duration is the kernel alone
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsdp_amd import _lib, hip_linear, synth
from nsdp_amd.model import build_model, optimizer_factory

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--workload", default="forward", choices=["forward", "arbitrary"])
ap.add_argument("--all-classes", action="store_true", help="list the calls of every kernel class, not only the dense layers'")
ap.add_argument("--json", default=None, help="also write, per bf16x3 GEMM kernel TEMPLATE (the name rocprofv3 reports): launches / step and "
                                              "the algorithmic bytes per launch split by operand -- what tools/pmc_attribution.py joins with the PMC passes")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = bench.model_config()
if args.workload == "arbitrary":
    cfg["model"]["type"] = "arbitrary"
model, train_fn, *_ = build_model(cfg, device="cpu")
state = synth.procedural_state_dict(model.state_dict(), 2048)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model.to(dev).train()
_, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, args.batch, bench.N_SURF, bench.N_QUERY).items()}

# ---- the header: prototype -> parameter names and which of them are integers / pointers -------------------------------------
text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
protos = {}
for m in re.finditer(r"\b(?:int|size_t|long long|void|const char \*)\s*\b(nsdp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
    name, plist = m.group(1), m.group(2)
    params = []
    for prm in [p.strip() for p in plist.replace("\n", " ").split(",") if p.strip() and p.strip() != "void"]:
        pm = re.match(r"(.*?)(\w+)$", prm)
        ptype, pname = pm.group(1).strip(), pm.group(2)
        params.append((pname, "ptr" if "*" in ptype else ("int" if re.search(r"\b(int|long long|size_t|unsigned)\b", ptype) else "other")))
    protos[name] = params
SKIP = re.compile(r"nsdp_(prof|trace|debug|last_error|abi_version|device_count|graph_exec|.*_supported$|.*_bytes$|.*_ok$|.*_floats$|adam_chunk)")
L = hip_linear.lib()
NK = L.nsdp_prof_num_kinds()
L.nsdp_prof_name.restype = ctypes.c_char_p
KINDS = [L.nsdp_prof_name(i).decode() for i in range(NK)]
calls = collections.OrderedDict()      # key -> {"n": calls, kind: [launches, ms, flops, bytes]}
recording = [False]
templates = collections.OrderedDict()  # rocprof kernel name of a bf16x3 GEMM -> {"launches", operand -> algorithmic bytes (summed)}


def template_name(trace):
    """NSDP_TRACE name of a bf16x3 GEMM launch -> the kernel's C++ template name as rocprofv3 prints it."""
    m = re.match(r"linear_bf16x3<(\d+),(\d+),(\d+),(\d+),(\d+)>(.*)", trace)
    if not m:
        return None
    mt, nt, pre, wv, xreg, rest = int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5]), m[6]
    kbm = 4 if " wres" in rest else 2
    gather = 2 if " gather1" in rest else 1 if " gather" in rest else 0
    lay = {"x": 1, "y": 2, "xy": 3, "r": 4, "xr": 5, "yr": 6, "xyr": 7}.get((re.search(r" g16:(\w+)", rest) or [None, ""])[1], 0)
    tail = 0
    if " k4tail" in rest:
        tail, kbm = -1, 2      # (k_out decides 1 / 2: filled in by the caller)
    return [mt, nt, pre, wv, "true" if xreg else "false", kbm, gather, tail, lay]


def operand_bytes(name, a, params):
    """Algorithmic HBM bytes of one bf16x3 GEMM call by operand (the small L2-resident tables and the weights count as `other`)."""
    v = {p: value(x) for (p, t), x in zip(params, a)}
    M, N, K = int(v.get("M", 0)), int(v.get("N", 0)), int(v.get("K", 0))
    out = {"X": 4.0 * M * K, "mask": 0.0, "residual": 0.0, "other": 4.0 * N * K, "Y": 4.0 * M * N}
    if "h0" in name:
        out["X"] = 16.0 * M
    if "k4tail" in name:
        out["Y"] = 0.0
        out["other"] += 16.0 * M
    if v.get("mask_bits"):
        out["mask"] = float(((M + 15) // 16) * ((K + 31) // 32) * 64)
    elif v.get("mask"):
        out["mask"] = 4.0 * M * K
    for opnd in ("residual", "out_mask", "addend"):
        if v.get(opnd):
            out["residual"] += 4.0 * M * N
    if v.get("bits_out"):
        out["Y"] += float(((M + 15) // 16) * ((N + 31) // 32) * 64)
    return out, v


def value(a):
    v = getattr(a, "value", a)
    return v


def wrap(name, fn, params):
    def wrapper(*a):
        if not recording[0]:
            return fn(*a)
        torch.cuda.synchronize()
        L.nsdp_prof_enable(1)              # (re-enabling clears the records)
        x3 = name.startswith("nsdp_linear_bf16x3")
        if x3:
            L.nsdp_trace_enable(1)
        rc = fn(*a)
        torch.cuda.synchronize()
        if x3:
            L.nsdp_trace_enable(0)
            nb = L.nsdp_trace_read(None, 0)
            tb = ctypes.create_string_buffer(nb)
            L.nsdp_trace_read(tb, nb)
            ob, v = operand_bytes(name, a, params)
            for tr in tb.value.decode().split("\n"):
                tn = template_name(tr)
                if tn is None:
                    continue
                if tn[7] == -1:
                    tn[7] = 2 if int(v.get("k_out", 4)) == 3 else 1
                key = "linear_bf16x3_kernel<" + ", ".join(str(t) for t in tn) + ">"
                ent = templates.setdefault(key, {"launches": 0, "X": 0.0, "mask": 0.0, "residual": 0.0, "other": 0.0, "Y": 0.0})
                ent["launches"] += 1
                for kk, bb in ob.items():
                    ent[kk] += bb
        ints = tuple((p, int(value(x))) for (p, t), x in zip(params, a) if t == "int" and p not in ("accumulate", "workspace_bytes", "ws_bytes"))
        ptrs = tuple(p for (p, t), x in zip(params, a) if t == "ptr" and value(x) and p in
                     ("bias", "residual", "mask", "out_mask", "addend", "gq", "db", "b0", "a_g", "qsub", "desc_out"))
        ent = calls.setdefault((name, ints, ptrs), {"n": 0})
        ent["n"] += 1
        for k in range(NK):
            n, ms, fl, by = ctypes.c_longlong(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
            L.nsdp_prof_collect(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
            if n.value:
                acc = ent.setdefault(KINDS[k], [0, 0.0, 0.0, 0.0])
                acc[0] += n.value; acc[1] += ms.value; acc[2] += fl.value; acc[3] += by.value
        return rc
    wrapper.restype = getattr(fn, "restype", None)
    return wrapper


wrapped = 0
for name, params in protos.items():
    if SKIP.match(name) or not hasattr(L, name):
        continue
    setattr(L, name, wrap(name, getattr(L, name), params))
    wrapped += 1

STEPS = 2
for i in range(1 + STEPS):
    recording[0] = i >= 1
    train_fn.tensor_step(model, opt, data, cfg)
    torch.cuda.synchronize()
recording[0] = False
L.nsdp_prof_enable(0)

print(f"# {wrapped} of {len(protos)} entry points of include/nsdp_hip.h wrapped; {len(calls)} distinct (entry point, shape, operands) "
      f"in {STEPS} {args.workload} train steps at B = {args.batch}; per-step figures below")
tot = collections.OrderedDict()
for key, ent in calls.items():
    for kind, v in ent.items():
        if kind == "n":
            continue
        t = tot.setdefault(kind, [0, 0.0, 0.0, 0.0])
        for j in range(4):
            t[j] += v[j]
print(f"{'class':28s} {'launches/step':>13s} {'algorithmic GB/step':>20s} {'isolated ms/step':>17s} {'GB/s':>7s} {'of 8 TB/s':>9s} {'TFLOP/s':>8s}")
for kind, (n, ms, fl, by) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{kind:28s} {n / STEPS:13.1f} {by / STEPS / 1e9:20.3f} {ms / STEPS:17.3f} {by / ms / 1e6 if ms else 0:7.0f} {by / ms / 8e9 if ms else 0:9.3f} "
          f"{fl / ms / 1e9 if ms else 0:8.1f}")
dense = ("linear_bf16x3_kernel", "wgrad_bf16x3_kernel", "linear_nt_kernel", "linear_wgrad_kernel", "linear_bf16_kernel", "wgrad_bf16_kernel")
print(f"\n{'entry point':34s} {'shape / flags (nonzero)':66s} {'operands':26s} {'calls':>5s} {'launches':>8s} {'us/call':>9s} {'ms/step':>8s} {'TF':>6s} {'alg MB':>9s}")
rows = []
for (name, ints, ptrs), ent in calls.items():
    kinds = [k for k in ent if k != "n"]
    if not kinds or (not args.all_classes and not any(k in dense for k in kinds)):
        continue
    n_l = sum(ent[k][0] for k in kinds); ms = sum(ent[k][1] for k in kinds); fl = sum(ent[k][2] for k in kinds); by = sum(ent[k][3] for k in kinds)
    shape = " ".join(f"{p}={v}" for p, v in ints if v != 0 or p in ("M", "N", "K"))      # (zero flags are not printed)
    rows.append((ms / STEPS, name.replace("nsdp_", ""), shape, ",".join(ptrs), ent["n"] / STEPS, n_l / ent["n"], 1e3 * ms / ent["n"], fl / ms / 1e9 if ms else 0, by / ent["n"] / 1e6))
if args.json:
    import json
    json.dump({"steps": STEPS, "batch": args.batch, "workload": args.workload,
               "templates": {k: {kk: (vv / STEPS) for kk, vv in v.items()} for k, v in templates.items()}},
              open(args.json, "w"), indent=1)
for ms_step, name, shape, ptrs, n, nl, us, tf, mb in sorted(rows, reverse=True):
    print(f"{name[:34]:34s} {shape[:66]:66s} {ptrs[:26]:26s} {n:5.1f} {nl:8.1f} {us:9.1f} {ms_step:8.3f} {tf:6.1f} {mb:9.2f}")

"""Where do the torch elementwise kernels of a train step come from?  One eager B = 32 step of bench.py's workload under
torch.profiler with Python stacks; device time of every aten op that launches a torch kernel (add, mul, copy_, fill_ ...),
grouped by the innermost nsdp_amd source line on its stack (autograd-engine accumulations have no Python frame: 'engine').
    python tools/trace_aten_ops.py [--batch 32] [--top 40]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    import torch
    from torch.profiler import ProfilerActivity, profile
    import bench
    from nsdp_amd import synth
    from nsdp_amd.model import build_model, optimizer_factory
    from nsdp_amd.model.utils import compute_l2_error
    dev = torch.device("cuda:0")
    cfg = bench.model_config()
    model, _, _, _ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), 2048)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.to(dev).train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1, "weight_decay": 0.0},
                               model.parameters())
    data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, args.batch, bench.N_SURF, bench.N_QUERY).items()}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"])
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    if os.environ.get("TRACE_DEBUG"):
        print(prof.key_averages().table(sort_by="self_device_time_total", row_limit=25, max_name_column_width=60))
    by_site = collections.defaultdict(lambda: [0.0, 0])
    by_op = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        t = ev.self_device_time_total
        if t <= 0:
            continue
        # no Python frame in the backward pass: name the autograd node the op runs under instead
        par, node = ev.cpu_parent, ""
        while par is not None:
            if "evaluate_function" in par.name or par.name.endswith("Backward") or "Backward" in par.name:
                node = par.name.replace("autograd::engine::evaluate_function: ", "")
            par = par.cpu_parent
        site = (node or "forward") + "  " + str(ev.input_shapes)[:70]
        for fr in ev.stack or ():
            if "nsdp_amd" in fr and "torch/" not in fr:
                site = fr.split("nsdp_amd/")[-1]
                break
        by_site[(ev.name, site)][0] += t
        by_site[(ev.name, site)][1] += 1
        by_op[ev.name][0] += t
        by_op[ev.name][1] += 1
    if os.environ.get("TRACE_SEQUENCE"):
        # the backward pass as the engine ran it: every node, and the accumulations (aten::add*) it performed
        tops = sorted((e for e in prof.events() if "evaluate_function" in e.name), key=lambda e: e.time_range.start)
        for i, e in enumerate(tops):
            kids, stack = [], list(e.cpu_children)
            while stack:
                c = stack.pop()
                if c.name in ("aten::add", "aten::add_") and c.self_device_time_total > 20:
                    kids.append(f"{c.name}{str(c.input_shapes[0])} {c.self_device_time_total:.0f}us")
                stack.extend(c.cpu_children)
            print(f"  {i:4d} {e.name.replace('autograd::engine::evaluate_function: ', ''):32s} {' | '.join(kids)}")
    print("per op (device us, calls):")
    for k, (t, c) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:20]:
        print(f"  {k:32s} {t:9.1f} {c:5d}")
    print("per site:")
    for (name, site), (t, c) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:args.top]:
        print(f"  {t:8.1f} us {c:4d}x  {name:24s} {site}")


if __name__ == "__main__":
    main()

"""Timeline of a replayed train step from a rocprofv3 kernel trace: per queue (= HIP stream) busy time, the overlap between
queues, idle gaps of the busiest queue, and what the other queue ran during the last millisecond before each optimizer step.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --reps 1 --steps 5 --warmup 2 --no-cpu-baseline
    python tools/timeline.py DIR/**/t_kernel_trace.csv [--steps 5]"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--dump-ms", type=float, default=0.0, help="list every kernel that starts in the first DUMP_MS ms of the step")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    rows.sort()
    # the timed steps: the last `steps` occurrences of the fused Adam kernel chain end a step; cut at the first multi_tensor launch
    # of each optimizer step (a gap of > 5 ms between multi_tensor kernels separates steps)
    opt = [r for r in rows if "multi_tensor_apply" in r[3] or "adam_multi_kernel" in r[3]]
    step_ends, last = [], None
    for r in opt:
        if last is None or r[0] - last > 5_000_000:
            step_ends.append(r[0])
        last = r[0]
    step_ends = step_ends[-(args.steps + 1):]
    print(f"{len(rows)} dispatches, {len(step_ends) - 1} whole steps, step times (ms):",
          [round((b - a) / 1e6, 2) for a, b in zip(step_ends, step_ends[1:])])
    a, b = step_ends[-2], step_ends[-1]
    step = [r for r in rows if a <= r[0] < b]
    queues = collections.defaultdict(list)
    for r in step:
        queues[r[2]].append(r)
    print("last step: per queue  busy ms / kernels / first start / last end (ms from step start)")
    for q, rs in sorted(queues.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
        busy = sum(r[1] - r[0] for r in rs) / 1e6
        print(f"  queue {q}: {busy:7.2f} ms  {len(rs):5d}  {(rs[0][0] - a) / 1e6:7.2f}  {(max(r[1] for r in rs) - a) / 1e6:7.2f}")
    # union busy time and overlap
    ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
    depth, t_prev, any_busy, both = 0, a, 0, 0
    idle_gaps = []
    for t, d in ev:
        if depth >= 1:
            any_busy += t - t_prev
        elif t - t_prev > 20_000:
            idle_gaps.append((t_prev, t))
        if depth >= 2:
            both += t - t_prev
        depth += d
        t_prev = t
    print(f"  some kernel running {any_busy / 1e6:.2f} ms, two or more {both / 1e6:.2f} ms, step {(b - a) / 1e6:.2f} ms")
    print(f"  gaps > 20 us with NO kernel anywhere: {len(idle_gaps)}, total {sum(y - x for x, y in idle_gaps) / 1e6:.2f} ms")
    for x, y in sorted(idle_gaps, key=lambda g: g[0] - g[1])[:8]:
        before = [r for r in step if r[1] <= x]
        after = [r for r in step if r[0] >= y]
        print(f"    {(y - x) / 1e3:7.1f} us at {(x - a) / 1e6:6.2f} ms  after {before[-1][3][:50] if before else '-'}  before {after[0][3][:50] if after else '-'}")
    # per queue: kernels stretched relative to their own median (victims of the other queue)
    main_q = max(queues, key=lambda q: len(queues[q]))
    print(f"main queue = {main_q}; the other queues' kernels by total time:")
    other = collections.defaultdict(lambda: [0, 0])
    for q, rs in queues.items():
        if q == main_q:
            continue
        for r in rs:
            other[(q, r[3][:70])][0] += r[1] - r[0]
            other[(q, r[3][:70])][1] += 1
    for (q, n), (t, c) in sorted(other.items(), key=lambda kv: -kv[1][0])[:args.top]:
        print(f"    q{q} {t / 1e6:6.2f} ms {c:4d}x {n}")
    # busy share of the main queue / of the others per millisecond of the step (where does a queue sit idle?)
    nb = int((b - a) / 1e6) + 1
    share = {True: [0.0] * nb, False: [0.0] * nb}
    for r in step:
        t0, t1 = r[0], min(r[1], b)
        while t0 < t1:
            i = int((t0 - a) / 1e6)
            edge = min(t1, a + (i + 1) * 1_000_000)
            share[r[2] == main_q][i] += (edge - t0) / 1e6
            t0 = edge
    print("busy share per ms of the step (main queue | other queues):")
    for i in range(nb):
        print(f"    {i:3d} ms  {share[True][i]:5.2f} | {share[False][i]:5.2f}")
    if args.dump_ms > 0:
        print(f"kernels starting in the first {args.dump_ms} ms (start ms, dur us, queue):")
        for r in step:
            if r[0] - a < args.dump_ms * 1e6:
                print(f"    {(r[0] - a) / 1e6:7.3f} {(r[1] - r[0]) / 1e3:8.1f} q{r[2]} {r[3][:90]}")
    # when does each queue finish relative to the step end
    print("tail: last 12 kernels of the step (start ms, dur us, queue)")
    for r in sorted(step, key=lambda r: r[1])[-12:]:
        print(f"    {(r[0] - a) / 1e6:7.3f} {(r[1] - r[0]) / 1e3:8.1f} q{r[2]} {r[3][:80]}")


if __name__ == "__main__":
    main()

"""CPUs this process may really use: the scheduler affinity AND the cgroup CPU quota.

A 256-thread box whose container has a 16-CPU quota (cgroup `cpu.max` = "1600000 100000") looks like 256 CPUs to
`os.cpu_count()`; OpenMP / PyTorch then start 128-256 worker threads whose spin-waiting burns the quota, the cgroup is
throttled for the rest of every 100 ms period, and the one thread that matters -- the one enqueueing GPU work -- stalls with
it: measured as train steps of 60-90 ms instead of 45 ms on otherwise idle boxes (the kernels' own durations unchanged).
Entry points that own the process (bench.py, the smoke test, the test session) cap the CPU thread pools with this."""
import os


def cpu_budget() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),):
        try:
            quota, period = parse(open(path).read())
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError, IndexError):
            pass
    try:      # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cap_thread_pools(limit: int = 16) -> int:
    """Set OMP / MKL thread counts (environment, for pools not yet created) and PyTorch's intra-op pool (if torch is already
    imported) to min(limit, cpu_budget()); returns the number."""
    import sys
    n = min(limit, cpu_budget())
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ.setdefault(var, str(n))
    torch = sys.modules.get("torch")
    if torch is not None:
        torch.set_num_threads(min(n, int(os.environ.get("OMP_NUM_THREADS", n))))
    return n

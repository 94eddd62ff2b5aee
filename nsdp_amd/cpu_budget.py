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


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(pci_bus_id: str):
    """CPUs of the NUMA node a GPU hangs off (sysfs `local_cpulist` of its PCI function), or None when unknown."""
    try:
        with open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/local_cpulist") as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def pin_rank(local_rank: int, local_world: int, pci_bus_id: str | None = None):
    """One process per GPU, several per node: give rank r its own slice of the CPUs this container may use, taken from the
    NUMA node of its GPU when sysfs tells (the thread that enqueues ~1000 launches per step must not migrate across
    sockets or share a core with another rank's enqueue thread).  Ranks whose GPUs share a NUMA node split that node's
    CPUs by local rank.  Returns the sorted CPU list the process is now bound to (unchanged affinity on failure)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    pool = allowed
    near = gpu_local_cpus(pci_bus_id) if pci_bus_id else None
    if near:
        both = [c for c in allowed if c in near]
        if len(both) >= max(1, len(allowed) // max(1, 2 * local_world)):   # (a sane intersection, else ignore sysfs)
            pool = both
    n = max(1, local_world)
    per = max(1, len(pool) // n)
    mine = pool[(local_rank % n) * per:(local_rank % n) * per + per] or pool
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return allowed
    return mine

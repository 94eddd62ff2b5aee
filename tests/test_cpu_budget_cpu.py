"""nsdp_amd.cpu_budget: the CPUs a process may really use (affinity AND cgroup quota), and the thread-pool cap."""
import os


def test_cpu_budget_is_bounded_by_affinity_and_positive():
    from nsdp_amd.cpu_budget import cpu_budget
    n = cpu_budget()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    try:      # where a cgroup-v2 quota exists, it is honoured
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            assert n <= max(1, int(quota) // int(period))
    except OSError:
        pass


def test_cap_thread_pools_limits_torch_and_environment(monkeypatch):
    import torch
    from nsdp_amd.cpu_budget import cap_thread_pools, cpu_budget
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        monkeypatch.delenv(var, raising=False)
    before = torch.get_num_threads()
    try:
        n = cap_thread_pools(4)
        assert n == min(4, cpu_budget())
        assert torch.get_num_threads() <= 4 and os.environ["OMP_NUM_THREADS"] == str(n)
    finally:
        torch.set_num_threads(before)

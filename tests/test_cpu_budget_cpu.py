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


def test_pin_rank_gives_each_local_rank_its_own_cpus():
    import os
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); from nsdp_amd.cpu_budget import pin_rank;"
            "a = sorted(os.sched_getaffinity(0)); m0 = pin_rank(int(sys.argv[1]), 2, '0000:ff:1f.0');"
            "print(len(a), m0, sorted(os.sched_getaffinity(0)) == m0)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, timeout=60).stdout.strip()
            for r in (0, 1)]
    n0, rest0 = outs[0].split(" ", 1)
    assert rest0.endswith("True") and outs[1].endswith("True"), outs
    if int(n0) >= 2:
        m0 = eval(outs[0].split(" ", 1)[1].rsplit(" ", 1)[0])
        m1 = eval(outs[1].split(" ", 1)[1].rsplit(" ", 1)[0])
        assert not set(m0) & set(m1) and len(m0) == len(m1) == int(n0) // 2

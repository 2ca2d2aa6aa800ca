"""bench.py's helpers, split by concern (round-4 verdict, task 10): common.py small shared pieces, roofline.py byte models +
live PMC passes, batch.py the resident pattern batch and the timed loop, extras.py the secondary measurements of the N = 1
run, multigpu.py what only N > 1 runs, cpu_baseline.py the reference / oracle timed beside the GPU.  bench.py keeps the
contract: arguments, the timed K steps, the one JSON line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_quota():
    """CPUs the container may use on average (cgroup v2 cpu.max / v1 cfs quota), or None: the GPU box shows 256 hardware threads
    but runs this process under a quota of 16 -- host-side stages that use 128 threads are bursts, and back-to-back bursts are
    throttled (the 40-75 ms outliers of the host-pointer path)"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:      # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:      # noqa: BLE001
        return None


def source_hash():
    """hash of the kernel / host sources: ties a committed PMC file to the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "femto_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

"""Host CPU budget for the CPU oracle / cpu_baseline leg (TEST INFRASTRUCTURE).

The GPU boxes expose 256 logical CPUs but run the job under a cgroup quota (16 CPUs at
the time of writing); letting torch start one thread per logical CPU makes the CPU
reference ~100x slower through throttling.  Use the quota when there is one."""
from __future__ import annotations

import math
import os


def effective_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.floor(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)

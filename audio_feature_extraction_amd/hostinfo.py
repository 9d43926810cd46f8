"""What the host lets this process use.  A GPU box hands a 1-GPU job a share of its host (e.g. 16 of 256 hardware
threads): thread and process pools sized by ``os.cpu_count()`` then measure the scheduler, not the code."""
from __future__ import annotations

import os


def usable_cpus() -> int:
    """Cores this process can actually run on: the affinity mask, cut down to the cgroup's CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read().split()[0]) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n

#!/usr/bin/env python3
"""Do the control group's throttle counters (benchlib/hostinfo.py: cpu.stat at the cgroup level whose cpu.max binds) move when they
should?  Busy processes for a moment: 1, 8, then 4 x the CPU quota (throttled in every 100-ms period), then a sleep.
  python3 scripts/cpustat_diag.py"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib import hostinfo  # noqa: E402


def busy(sec):
    t = time.time()
    while time.time() - t < sec:
        pass


def region(label, n, sec):
    with hostinfo.HostRegion() as h:
        ps = [mp.Process(target=busy, args=(sec,)) for _ in range(n)]
        [p.start() for p in ps]
        [p.join() for p in ps]
    print(label, n, sec, h.record(), flush=True)


if __name__ == "__main__":
    quota = hostinfo.cpu_quota_cores()
    print("quota dir", hostinfo._quota_dir(), "quota", quota, "cpu.stat", hostinfo.cpu_stat())
    region("one process", 1, 0.5)
    region("8 processes", 8, 0.5)
    region("4 x quota processes", 4 * quota, 1.0)
    with hostinfo.HostRegion() as h:
        time.sleep(0.5)
    print("sleep", h.record())

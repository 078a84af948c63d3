"""What the host gives this process: the CPU quota of its control group, and how often the scheduler throttled the group while a
region ran (cpu.stat: a process that exhausts its quota is stopped as a whole -- the thread that launches kernels included -- for
the rest of the 100-ms accounting period, DESIGN.md sections 7 and 8)."""
import os


def _cgroup_dir():
    """The process's own control group in the mounted hierarchy (cgroup v2: '0::<path>' in /proc/self/cgroup)."""
    rel = "/"
    try:
        for line in open("/proc/self/cgroup"):
            if line.startswith("0::"):
                rel = line[3:].strip() or "/"
    except OSError:
        pass
    d = os.path.join("/sys/fs/cgroup", rel.lstrip("/"))
    return d if os.path.isdir(d) else "/sys/fs/cgroup"


def _quota_dir():
    """The control group whose CPU quota binds this process: walking up from its own group, the first level whose cpu.max holds a
    quota (counters of a level without one never move), else the mounted root."""
    d = _cgroup_dir()
    while True:
        try:
            first = open(os.path.join(d, "cpu.max")).read().split()
            if first and first[0] != "max":
                return d
        except OSError:
            pass
        if d.rstrip("/") in ("/sys/fs/cgroup", ""):
            return "/sys/fs/cgroup"
        d = os.path.dirname(d.rstrip("/"))


def cpu_stat():
    """{nr_periods, nr_throttled, throttled_usec} of the control group whose quota binds the process, or {}."""
    for path in (os.path.join(_quota_dir(), "cpu.stat"), "/sys/fs/cgroup/cpu/cpu.stat"):         # v2, then the v1 controller
        try:
            kv = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        if "nr_throttled" in kv:
            usec = int(kv["throttled_usec"]) if "throttled_usec" in kv else int(kv.get("throttled_time", 0)) // 1000
            return {"nr_periods": int(kv.get("nr_periods", 0)), "nr_throttled": int(kv["nr_throttled"]), "throttled_usec": usec,
                    "file": path}
    return {}


def cpu_quota_cores():
    """CPUs the process may really use (advntr_host_threads: hardware threads cut down to affinity and the cgroup quota)."""
    from advntr_amd import _lib
    return int(_lib.load().advntr_host_threads())


class HostRegion(object):
    """Throttle counters around a region: `with HostRegion() as h: ...; h.record()`."""

    def __enter__(self):
        import time
        self.before = cpu_stat()
        self.t0, self.c0 = time.perf_counter(), time.process_time()
        return self

    def __exit__(self, *exc):
        import time
        self.wall_s, self.cpu_s = time.perf_counter() - self.t0, time.process_time() - self.c0
        self.after = cpu_stat()
        return False

    def record(self):
        b, a = self.before, getattr(self, "after", None) or cpu_stat()
        rec = {"cpu_quota_cores": cpu_quota_cores(), "host_threads_available": os.cpu_count(), "cpu_stat_file": (a or {}).get("file"),
               # CPU seconds this PROCESS (all its threads) burned in the region, and the region's wall time: a rank that sleeps in its
               # device waits (ADVNTR_BLOCKING_SYNC) shows a small fraction of a core, one that spins a whole one
               "process_cpu_s": getattr(self, "cpu_s", None), "wall_s": getattr(self, "wall_s", None)}
        if a and b:
            rec.update({"nr_periods_delta": a["nr_periods"] - b["nr_periods"], "nr_throttled_delta": a["nr_throttled"] - b["nr_throttled"],
                        "throttled_usec_delta": a["throttled_usec"] - b["throttled_usec"]})
        else:
            rec.update({"nr_periods_delta": None, "nr_throttled_delta": None, "throttled_usec_delta": None,
                        "note": "no cpu.stat with throttle counters in this control group"})
        return rec

    @staticmethod
    def since(region):
        """The record of a region that was entered by hand and ends now."""
        region.__exit__(None, None, None)
        return region.record()

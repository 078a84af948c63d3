"""Sustained engine clock / power while the Viterbi kernel runs back to back (rocm-smi polled from a thread).
Usage: python scripts/clock_probe.py [seconds]"""
import subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12
loc = workloads.ref150()
reads = workloads.make_reads(np.random.default_rng(1), loc, 100000, 150)
bases, off = _lib.encode_reads(reads)
B = _lib.DeviceBatch([loc.model.device_model()], bases, off, np.zeros(len(reads), np.int32))
B.run()
stop = False
samples = []


def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             stdin=subprocess.DEVNULL, timeout=20).stdout.decode()
        samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power (W)" in l])
        time.sleep(0.5)


t = threading.Thread(target=poll)
t.start()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    B.run()
    n += 1
stop = True
t.join()
print("launches", n, "ms per launch incl. sync", 1e3 * (time.time() - t0) / n)
for s in samples:
    print(s)

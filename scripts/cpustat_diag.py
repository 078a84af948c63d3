import time, threading, sys
sys.path.insert(0, '.')
from benchlib import hostinfo
def busy(sec):
    t=time.time()
    while time.time()-t<sec: pass
def region(label, nthreads, sec):
    with hostinfo.HostRegion() as h:
        ts=[threading.Thread(target=busy,args=(sec,)) for _ in range(nthreads)]
        # threads in Python hold the GIL: use processes instead
        import multiprocessing as mp
        ps=[mp.Process(target=busy,args=(sec,)) for _ in range(nthreads)]
        [p.start() for p in ps]; [p.join() for p in ps]
    print(label, nthreads, sec, h.record(), flush=True)
print(open('/sys/fs/cgroup/cpu.stat').read())
region('one process', 1, 0.5)
region('8 processes', 8, 0.5)
region('64 processes', 64, 1.0)
with hostinfo.HostRegion() as h: time.sleep(0.5)
print('sleep', h.record())
print(open('/sys/fs/cgroup/cpu.stat').read())

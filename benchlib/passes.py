"""Consecutive passes over a resident device batch, one or two of them in flight."""
import time

SECOND_QUEUE = 128        # ADVNTR_FLAG_SECOND_QUEUE (include/advntr_hip.h)


class Passes(object):
    """Consecutive passes over ONE resident batch, one or two of them queued at a time.  With two, the passes alternate between
    two copies of the device batch -- same models, same reads, scratch, result arrays and stream of their own: pass k + 1 is
    queued behind nothing but its own copy's previous pass and starts while the last workgroups of pass k drain (the dynamic
    dequeue of a launch ends on single sweeps: 2-4 % of a launch, most of what separates an 8-rank share from an eighth of the
    whole set).  Every pass scores every read; the copies hold identical results."""

    def __init__(self, make, in_flight):
        # make(extra_flags) -> device batch; the second copy's stream is of a class of its own (ADVNTR_FLAG_SECOND_QUEUE): two
        # streams of one class can land on the same hardware queue, where their kernels would run strictly one after the other
        self.batches = [make(SECOND_QUEUE if i else 0) for i in range(max(1, int(in_flight)))]
        self.k = 0

    def run(self, reserve=0):
        b = self.batches[self.k % len(self.batches)]
        self.k += 1
        if reserve:
            b.reserve_next(reserve)
        b.run()
        return b

    def sync(self):
        for b in self.batches:
            b.sync()

    def ms_per_pass(self, steps, warm=2, reserve=0):
        for _ in range(warm):
            self.run(reserve)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run(reserve)
        self.sync()
        return (time.perf_counter() - t0) / steps * 1e3

    def close(self):
        for b in self.batches:
            b.close()


def passes_of(batch):
    """One pass at a time over an existing device batch."""
    one = Passes(lambda extra: None, 0)
    one.batches = [batch]
    return one


def two_in_flight_ms(batch, make, steps, reserve=0):
    """ms per pass with two passes in flight: `batch` and a second copy of it made here (and given back)."""
    twin = make(SECOND_QUEUE)
    try:
        both = passes_of(batch)
        both.batches.append(twin)
        return both.ms_per_pass(steps, reserve=reserve)
    finally:
        twin.close()

"""Helper of tests/test_rccl_peers.py (run as a child process with ADVNTR_RCCL_LIB = tests/native/libfake_rccl.so): several
ranks of the result gather inside ONE process on ONE GPU -- a host thread per rank, each with its own rendezvous object,
RcclComm (csrc/abi_comm.h through the C ABI) and device batch.  What no 1-GPU lease could execute before: the Send branch of
comm_gatherv_post, one Recv per peer, a peer with no reads, a root that holds no reads, a root other than rank 0, the gather
of pass i in flight while pass i + 1 runs.  Prints one JSON line."""
import ctypes
import json
import os
import sys
import tempfile
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def blob_of(rank, world):
    return b"" if rank == world - 1 else ("rows of rank %d" % rank).encode() * (rank + 1)


def scenario(world, root, counts, seed):
    from advntr_amd import _lib, comm, workloads
    locus = workloads.s300()
    dm = locus.model.device_model()
    rdzv_dir = tempfile.mkdtemp(prefix="advntr_fake_rccl_")
    os.chmod(rdzv_dir, 0o700)
    out, errors = {}, []

    def rank_main(rank):
        try:
            c = comm.RcclComm(comm.FileRendezvous(rank, world, rdzv_dir, timeout=120.0), init_timeout=120.0)
            reads = workloads.make_reads(np.random.default_rng(seed + rank), locus, counts[rank], 150)
            bases, off = _lib.encode_reads(reads) if reads else (np.zeros(0, np.uint8), np.zeros(1, np.int64))
            # two copies of the rank's device batch, as bench.py's strong-scaling lines alternate them (class Passes)
            batch = _lib.DeviceBatch([dm], bases, off, np.zeros(counts[rank], np.int32))
            twin = _lib.DeviceBatch([dm], bases, off, np.zeros(counts[rank], np.int32), flags=_lib.FLAG_SECOND_QUEUE)
            batch.run()
            mine = batch.fetch()
            seen = c.allgather_i64(batch.n_reads)
            assert seen == list(counts), (seen, counts)
            # pass 1's gather (first copy) is in flight while pass 2 runs on the other copy; pass 2's gather is queued while
            # pass 3 is on the first copy again, and fetched
            c.gather_results_start(batch, seen, root=root)
            twin.run()
            c.gather_results_finish(fetch=False)
            c.gather_results_start(twin, seen, root=root)
            batch.run()
            logp, summ = c.gather_results_finish(fetch=True)
            batch.sync()
            twin.close()
            blobs = c.gather_bytes(blob_of(rank, world), root)
            c.barrier()
            top = c.allreduce_max(10.0 + rank)
            out[rank] = {"mine": mine, "gathered": (logp, summ), "blobs": blobs, "max": top, "gather_ms": c.last_gather_ms()}
            batch.close()
            c.close()
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            import traceback
            errors.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    if errors or any(t.is_alive() for t in threads):
        raise RuntimeError("; ".join(errors) or "a rank thread did not finish")
    want_logp = np.concatenate([out[r]["mine"][0] for r in range(world)])
    want_summ = np.concatenate([out[r]["mine"][1] for r in range(world)])
    got_logp, got_summ = out[root]["gathered"]
    ok = bool(np.array_equal(got_logp, want_logp) and np.array_equal(got_summ, want_summ) and len(got_logp) == sum(counts))
    ok = ok and all(out[r]["gathered"] == (None, None) for r in range(world) if r != root)
    ok = ok and out[root]["blobs"] == [blob_of(r, world) for r in range(world)]
    ok = ok and all(out[r]["blobs"] is None for r in range(world) if r != root)
    ok = ok and all(out[r]["max"] == 10.0 + world - 1 for r in range(world))
    return {"world": world, "root": root, "counts": list(counts), "ok": ok, "records": int(len(got_logp)),
            "with_repeats": int((want_summ[:, 0] > 0).sum()), "gather_ms_root": out[root]["gather_ms"]}


def main():
    lib_path = os.environ.get("ADVNTR_RCCL_LIB", "")
    assert lib_path.endswith("libfake_rccl.so"), "run with ADVNTR_RCCL_LIB = tests/native/libfake_rccl.so"
    import __graft_entry__ as entry
    entry.build()
    from advntr_amd import _lib
    _lib.require_gpu()
    fake = ctypes.CDLL(lib_path)            # the same object abi_comm.h dlopen'ed (one copy per process): its counters
    res = []

    def stats():
        s = (ctypes.c_int64 * 8)()
        fake.fake_rccl_stats(s)
        return list(s)
    before = stats()
    for world, root, counts, seed in ((3, 1, (700, 400, 0), 11), (2, 0, (0, 500), 21), (4, 3, (64, 1, 0, 333), 31)):
        r = scenario(world, root, counts, seed)
        after = stats()
        r["fake_rccl"] = dict(zip(("sends", "recvs", "groups", "bytes_sent", "allgathers", "allreduces", "comms"),
                                  [a - b for a, b in zip(after, before)]))
        before = after
        res.append(r)
    print(json.dumps({"scenarios": res}))


if __name__ == "__main__":
    main()

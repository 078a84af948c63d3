"""The multi-rank legs of csrc/abi_comm.h on a 1-GPU box.  RCCL refuses two ranks on one device, so no lease of this pool
could ever run the Send branch of the gather, a Recv per peer, a zero-count peer or a root other than 0.  tests/native/
fake_rccl.cpp (test infrastructure; compiled against <rccl/rccl.h>'s own prototypes) is loaded in RCCL's place through
ADVNTR_RCCL_LIB and carries matched ncclSend / ncclRecv out as device-to-device copies between the ranks -- host threads of one
process -- with RCCL's blocking semantics for communicator creation and collectives."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "fake_rccl.cpp")
LIB = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")


def build_fake():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall",
                               "-o", LIB, SRC])
    return LIB


def test_fake_rccl_builds_against_the_rccl_header_and_resolves_every_entry_point(tmp_path):
    """CPU: the stand-in compiles against <rccl/rccl.h> (the prototypes abi_comm.h derives its pointers from) and
    advntr_comm_available() accepts it: every entry point abi_comm.h looks up is there.  A library without them is refused
    by name, before any rank could enter a collective."""
    lib = build_fake()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import __graft_entry__ as e; e.build()\n"
            "from advntr_amd import _lib\n"
            "rc = _lib.load().advntr_comm_available()\n"
            "print('available', rc, _lib.last_error() if rc else '')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ADVNTR_RCCL_LIB=lib), stdout=subprocess.PIPE,
                         check=True, timeout=600).stdout.decode()
    assert "available 0" in out, out
    empty = str(tmp_path / "libempty.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-x", "c", "-o", empty, "/dev/null"])
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ADVNTR_RCCL_LIB=empty), stdout=subprocess.PIPE,
                         check=True, timeout=600).stdout.decode()
    assert "available 0" not in out and "lacks ncclGetUniqueId" in out, out


@pytest.mark.gpu
def test_gather_with_peers_in_one_process():
    """GPU: three scenarios -- (world 3, root 1, a peer without reads), (world 2, root 0 holding no reads itself),
    (world 4, root 3, a one-read rank and an empty one) -- each rank a thread with its own communicator and batch: the
    gathered arrays on the root are the ranks' own results in rank order, nothing arrives elsewhere, the byte gather and the
    small collectives agree, and the stand-in saw exactly the sends / receives the ragged counts call for."""
    from rccl_peers_check import blob_of
    lib = build_fake()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_peers_check.py")],
                         env=dict(os.environ, ADVNTR_RCCL_LIB=lib), stdout=subprocess.PIPE, check=True, timeout=900).stdout.decode()
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)["scenarios"]
    assert len(res) == 3 and all(r["ok"] for r in res), res
    for r in res:
        world, root, counts = r["world"], r["root"], r["counts"]
        assert r["records"] == sum(counts)
        senders = sum(1 for k in range(world) if k != root and counts[k])
        blob_senders = [k for k in range(world) if k != root and blob_of(k, world)]
        f = r["fake_rccl"]
        # two result gathers x two arrays (log-probabilities, summaries) per sending peer + the byte gather's non-empty blobs
        assert f["sends"] == f["recvs"] == 2 * 2 * senders + len(blob_senders), (r, senders, blob_senders)
        assert f["bytes_sent"] == 2 * 40 * sum(counts[k] for k in range(world) if k != root) + \
            sum(len(blob_of(k, world)) for k in blob_senders), r
        assert f["comms"] == world

"""One process per GPU: RCCL communicator bootstrap without torch in the compute process.

The launcher contract (python -m torch.distributed.run --nproc-per-node N ...) provides RANK, LOCAL_RANK,
WORLD_SIZE, MASTER_ADDR and MASTER_PORT.  MASTER_PORT itself is owned by the launcher's store, so the
128-byte ncclUniqueId travels through a file on the node (single-node data parallelism): rank 0
writes it atomically, the other ranks poll for it.  Everything after that is RCCL on the ctx stream.
"""
import ctypes as C
import os
import time

import numpy as np

from ._lib import call


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


_T_IMPORT = time.time()
STALE_SLACK_S = 60.0     # ranks of one launch start within this many seconds of each other


def rendezvous_path():
    """GHM_RDZV_FILE if set; else a file keyed on what every worker of ONE job shares under any launcher:
    MASTER_ADDR, MASTER_PORT (unique per job on a node), WORLD_SIZE and the launcher's run id.  Under
    torch.distributed.run the launcher's pid is added (all its workers share it as parent)."""
    if os.environ.get("GHM_RDZV_FILE"):
        return os.environ["GHM_RDZV_FILE"]
    base = os.environ.get("GHM_RDZV_DIR", "/tmp")
    run_id = os.environ.get("TORCHELASTIC_RUN_ID")
    tag = "%s_%s_%s_%s" % (os.environ.get("MASTER_ADDR", "local").replace("/", "_"), os.environ.get("MASTER_PORT", "0"),
                           os.environ.get("WORLD_SIZE", "1"), run_id or "norunid")
    if run_id is not None:
        tag += "_%d" % os.getppid()
    return os.path.join(base, "ghm_rdzv_%s.uid" % tag)


def exchange_unique_id(rank, world, make_id, path=None, timeout=300.0):
    """rank 0: make_id() -> 128 bytes, published through ``path``; other ranks: wait for it.  A file left behind
    by a crashed earlier job with the same key is recognised by its age (older than this process by more than
    STALE_SLACK_S) and ignored until rank 0 replaces it."""
    path = path or rendezvous_path()
    if rank == 0:
        uid = bytes(make_id())
        assert len(uid) == 128
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            fresh = os.path.getmtime(path) >= _T_IMPORT - STALE_SLACK_S
            with open(path, "rb") as f:
                uid = f.read()
            if fresh and len(uid) == 128:
                return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no ncclUniqueId at %s after %.0fs" % (rank, path, timeout))
        time.sleep(0.01)


def shard_batch(arrays, rank, world):
    """Split a global batch evenly along axis 0; rank r gets rows [r*b, (r+1)*b)."""
    out = []
    for a in arrays:
        n = a.shape[0]
        if n % world:
            raise ValueError("global batch %d not divisible by world size %d" % (n, world))
        b = n // world
        out.append(a[rank * b:(rank + 1) * b])
    return out


class Comm:
    """RCCL communicator bound to a Device's stream."""

    def __init__(self, dev, rank, world, path=None):
        self.dev, self.rank, self.world = dev, rank, world
        self._path = path or rendezvous_path()

        def make_id():
            buf = (C.c_uint8 * 128)()
            call("ghm_comm_unique_id", C.byref(buf))
            return bytes(buf)

        uid = exchange_unique_id(rank, world, make_id, self._path)
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        call("ghm_comm_init", dev.h, rank, world, C.byref(buf))
        self._scratch = dev.zeros((1, 8, 1, 1))
        self.barrier()
        if rank == 0:
            try:
                os.remove(self._path)
            except OSError:
                pass

    def barrier(self):
        call("ghm_allreduce_sum", self.dev.h, C.c_void_p(self._scratch.ptr), 1)
        self.dev.sync()

    def max_scalar(self, v):
        self._scratch.set(np.array([v] + [0.0] * 7, np.float32))
        call("ghm_allreduce_max", self.dev.h, C.c_void_p(self._scratch.ptr), 8)
        return float(self._scratch.numpy().ravel()[0])

    def close(self):
        call("ghm_comm_destroy", self.dev.h)

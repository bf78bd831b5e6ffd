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


def _write_atomic(path, data):
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return None


NONCE = 16


def exchange_unique_id(rank, world, make_id, path=None, timeout=300.0):
    """rank 0: make_id() -> 128 bytes, published through ``path``; other ranks: wait for it.

    No clocks are compared: a file left behind by a crashed earlier job with the same key cannot be taken for this
    launch's, whenever the ranks start relative to each other.  Every other rank r announces itself with a random
    nonce in ``path.h<r>`` (and re-creates that file should rank 0's start-up sweep remove it); rank 0 first sweeps
    everything under the key, then publishes {unique id, the nonces it has seen} and republishes when an
    announcement changes; rank r accepts only a publication that carries ITS nonce and acknowledges it in
    ``path.a<r>``; rank 0 returns once every acknowledgement matches what it published."""
    path = path or rendezvous_path()
    t0 = time.time()

    def expired(what):
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: %s at %s after %.0fs" % (rank, what, path, timeout))

    if rank == 0:
        for stale in [path] + ["%s.%s%d" % (path, k, r) for k in "ha" for r in range(1, world)]:
            try:
                os.remove(stale)
            except OSError:
                pass
        uid = bytes(make_id())
        assert len(uid) == 128
        published = None
        while True:
            nonces = [_read("%s.h%d" % (path, r)) for r in range(1, world)]
            if all(n is not None and len(n) == NONCE for n in nonces):
                if nonces != published:
                    _write_atomic(path, uid + b"".join(nonces))
                    published = nonces
                if all(_read("%s.a%d" % (path, r)) == published[r - 1] for r in range(1, world)):
                    return uid
            expired("ranks missing at the rendezvous")
            time.sleep(0.005)
    nonce = os.urandom(NONCE)
    hello = "%s.h%d" % (path, rank)
    while True:
        if _read(hello) != nonce:
            _write_atomic(hello, nonce)
        pub = _read(path)
        if pub is not None and len(pub) == 128 + NONCE * (world - 1) and \
                pub[128 + NONCE * (rank - 1):128 + NONCE * rank] == nonce:
            _write_atomic("%s.a%d" % (path, rank), nonce)
            return pub[:128]
        expired("no ncclUniqueId for this launch")
        time.sleep(0.005)


def cleanup_rendezvous(path, world):
    for f in [path] + ["%s.%s%d" % (path, k, r) for k in "ha" for r in range(1, world)]:
        try:
            os.remove(f)
        except OSError:
            pass


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

    def __init__(self, dev, rank, world, path=None, channels=None):
        """``channels=(min, max)`` (or GHM_RCCL_CHANNELS="min,max"): RCCL's channel count = the CUs its kernels occupy beside
        the three compute streams of the step (NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS, read by RCCL when the communicator
        is created) -- unmeasured on hardware, plumbed so that the first multi-GPU run can sweep it"""
        self.dev, self.rank, self.world = dev, rank, world
        self._path = path or rendezvous_path()
        if channels is None and os.environ.get("GHM_RCCL_CHANNELS"):
            channels = tuple(int(v) for v in os.environ["GHM_RCCL_CHANNELS"].split(","))
        if channels is not None:
            os.environ["NCCL_MIN_NCHANNELS"], os.environ["NCCL_MAX_NCHANNELS"] = str(int(channels[0])), str(int(channels[1]))
        self.channels = channels

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
            cleanup_rendezvous(self._path, world)

    def nranks(self):
        """the rank count RCCL reports for this communicator (ncclCommCount)"""
        n = C.c_int32()
        call("ghm_comm_count", self.dev.h, C.byref(n))
        return n.value

    def barrier(self):
        call("ghm_allreduce_sum", self.dev.h, C.c_void_p(self._scratch.ptr), 1)
        self.dev.sync()

    def max_scalar(self, v):
        self._scratch.set(np.array([v] + [0.0] * 7, np.float32))
        call("ghm_allreduce_max", self.dev.h, C.c_void_p(self._scratch.ptr), 8)
        return float(self._scratch.numpy().ravel()[0])

    def close(self):
        call("ghm_comm_destroy", self.dev.h)

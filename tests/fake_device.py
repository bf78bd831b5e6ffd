"""Host-only stand-ins for gan_heightmaps_amd.device.{Device, Ops}: fake HBM addresses and a recorder of the
op calls, so the lowering / placement / program-emission logic can be tested without a GPU.  No arithmetic."""
import numpy as np

from gan_heightmaps_amd.device import DevTensor


class FakeDevice:
    def __init__(self, index=0):
        self.h = None
        self.index = index
        self._next = 1 << 20
        self.bytes_allocated = 0
        self.uploads = 0
        self._vals = {}             # ptr -> host copy of small tensors made by tensor() (d2h returns them)
        self.ls_state = None

    def alloc(self, nbytes):
        p = self._next
        self._next += (int(nbytes) + 255) // 256 * 256 + 256
        self.bytes_allocated += nbytes
        return p

    def free(self, ptr):
        pass

    def empty(self, shape):
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape))
        return DevTensor(self, self.alloc(4 * n), shape if len(shape) in (2, 4) else (1, n, 1, 1))

    zeros = empty

    def tensor(self, arr):
        arr = np.asarray(arr, np.float32)
        shape = arr.shape if arr.ndim in (2, 4) else (1, arr.size, 1, 1)
        t = self.empty(shape)
        if arr.size <= 64:
            self._vals[t.ptr] = arr.copy().ravel()
        return t

    def set_loss_scale_state(self, state):
        self.ls_state = state

    def h2d(self, ptr, arr):
        self.uploads += 1

    def d2h(self, arr, ptr, nbytes):
        arr[...] = 0
        if ptr in self._vals:
            arr.ravel()[:self._vals[ptr].size] = self._vals[ptr]

    def memset_zero(self, ptr, nbytes):
        pass

    def sync(self):
        pass

    def wait_for(self, other):
        pass

    # ---- the asynchronous input pipeline's device surface: events and uploads are logged, nothing is asynchronous ----
    class pinned_array:
        def __init__(self, shape, dtype=np.float32):
            self.shape, self.array, self.ptr = tuple(shape), np.zeros(shape, dtype), 1

        def close(self):
            self.array = None

    def event_create(self):
        self._nev = getattr(self, '_nev', 0) + 1
        return ("ev", id(self), self._nev)

    def event_record(self, ev):
        FakeDevice.pipe_log.append(("record", ev[2], self.name()))

    def event_wait(self, ev):
        FakeDevice.pipe_log.append(("wait", ev[2], self.name()))

    @staticmethod
    def event_sync(ev):
        FakeDevice.pipe_log.append(("host_sync", ev[2]))

    @staticmethod
    def event_destroy(ev):
        pass

    def h2d_async(self, ptr, pinned):
        FakeDevice.pipe_log.append(("h2d_async", int(ptr), float(np.asarray(pinned.array).ravel()[0]), self.name()))

    def d2d(self, dst, src, nbytes):
        FakeDevice.pipe_log.append(("d2d", int(dst), int(src), int(nbytes), self.name()))

    def name(self):
        return "dev%x" % (id(self) & 0xffff)

    def close(self):
        pass


FakeDevice.pipe_log = []


class RecordingOps:
    def __init__(self, dev):
        self.dev = dev
        self.calls = []

    def bn_workspace(self, C):
        return 1024

    def wgrad_workspace(self, d):
        return 1024

    def dgrad_t_supported(self, d):
        return d.stride == 1

    def conv_variant(self, d, kind):
        return "fake<%d>" % kind

    def blconv_supported(self, N, Cc, K, n1, n2):
        return Cc % 32 == 0 and K % 32 == 0 and n1 >= 2 and n2 >= 2

    def blconv_frame_sizes(self, N, Cc, K, n1, n2):
        lp = (2 * max(n1, n2) + 4 + 31) // 32 * 32
        return 2 * (N * 4 * Cc * lp + 64), 2 * 6 * N * K * (lp + 32)

    def wgrad_pooled_split_supported(self, d, dtype):        # (the sparse-instruction weight gradient: GPU tests only)
        return False

    def __getattr__(self, name):
        def rec(*args, **kw):
            self.calls.append((name, args, kw))
        return rec


FakeDevice.ops_class = RecordingOps


class PolicyOps(RecordingOps):
    """RecordingOps whose capability queries answer like libghm.so does for the big layers, so that the lowering's
    reduced-precision and fused conv + pool decisions can be exercised without a GPU: the low-precision kernels serve
    3x3 / 5x5 convs with channels % 16 == 0 and maps >= 32 wide, the fused conv + pool every stride-1 'same' conv with
    a linear / relu / lrelu epilogue (thin layers on the fp32 weights, the rest on the low-precision pack)."""

    def lp_supported(self, d, kind, dtype):
        if dtype == 'f32' or d.kh not in (3, 5) or d.kh != d.kw:
            return False
        red, rows = (d.C, d.K) if kind in (0, 2) else (d.K, d.C)
        if kind == 2:
            return d.Wo % 32 == 0 and d.K >= 32 and d.C * d.kh * d.kw >= 96
        return red % 16 == 0 and rows >= 32 and (d.Wo if kind == 0 else d.W) % 32 == 0

    def lp_weight_bytes(self, d, transposed=False, dtype=None):
        return 256

    def wgrad_lp_workspace(self, d):
        return 1024

    def lp_pack_table(self, items):
        self.calls.append(("lp_pack_table", tuple((int(t[2]), int(t[3]), int(t[4]), bool(t[5])) for t in items), {}))
        return (0, len(items), 1)

    def pool_bwd_sparse_supported(self, d, act):
        return 3 if (d.C == 1 and d.kh == 5 and d.kw == 5 and d.stride == 1 and d.K <= 64 and act in ('linear', 'relu', 'lrelu')) else 0

    def pool_wgrad_sparse_workspace(self, d):
        return 2048

    def conv_pool_supported(self, d, act, dtype='f32'):
        if act not in ('linear', 'relu', 'lrelu') or d.stride != 1 or d.Ho != d.H or d.W % 4:
            return 0
        if d.C <= 4:
            return 1
        return 2 if self.lp_supported(d, 0, dtype) else 1


class PolicyDevice(FakeDevice):
    ops_class = PolicyOps


# ---- a host-memory device: allocations are numpy arrays, and the handful of ops a data-parallel exchange needs do
# ---- arithmetic (all-reduce through torch.distributed/gloo, the optimiser update, memset); every other op is
# ---- recorded only.  Contexts made by ``type(dev)(index)`` share one arena and one host-order event log, like HIP
# ---- streams of one GPU share its memory.  Use ``host_device_class()`` for a fresh arena per test.
def host_device_class():
    import bisect

    class Shared:
        bases, arrays, log, nctx = [], {}, [], 0

    class HostOps:
        def __init__(self, dev):
            self.dev = dev

        def bn_workspace(self, C):
            return 1024

        def wgrad_workspace(self, d):
            return 1024

        def dgrad_t_supported(self, d):
            return d.stride == 1

        def conv_variant(self, d, kind):
            return "host<%d>" % kind

        def transpose_table(self, items):
            return (0, len(items), 0)

        def expand_table(self, items):
            return [int(it[1].ptr) for it in items]

        def upconv_expand_batched(self, table, accumulate=False):
            Shared.log.append((self.dev.name, "upconv_expand_batched") + tuple(table))     # the 5x5 gradients it writes

        def allreduce_sum(self, buf, n):
            import torch
            import torch.distributed as tdist
            v = self.dev.view(buf.ptr, n)
            t = torch.from_numpy(v)
            tdist.all_reduce(t)
            Shared.log.append((self.dev.name, "allreduce_sum", int(buf.ptr), int(n)))

        def allreduce_sum_bf16(self, buf, n, scratch_ptr):
            """the bf16 exchange buffer on the host: contributions rounded to bf16 (nearest even), summed, the sum rounded again
            (RCCL rounds per reduction hop; the program under test is what matters here, not the last bit)"""
            import torch
            import torch.distributed as tdist

            def bf16(a):
                u = a.view(np.uint32).astype(np.uint64)
                r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
                return r.astype(np.uint32).view(np.float32)
            v = self.dev.view(buf.ptr, n)
            t = torch.from_numpy(bf16(v.copy()))
            tdist.all_reduce(t)
            v[:] = bf16(t.numpy())
            Shared.log.append((self.dev.name, "allreduce_sum_bf16", int(buf.ptr), int(n)))

        def reduce_scatter_sum(self, buf, shard):
            # gloo has no reduce-scatter: all-reduce a copy and keep this rank's shard (the other shards stay as they were,
            # like RCCL's in-place form leaves them unspecified)
            import torch
            import torch.distributed as tdist
            world, rank = tdist.get_world_size(), tdist.get_rank()
            v = self.dev.view(buf.ptr, shard * world)
            t = torch.from_numpy(v.copy())
            tdist.all_reduce(t)
            v[rank * shard:(rank + 1) * shard] = t.numpy()[rank * shard:(rank + 1) * shard]
            Shared.log.append((self.dev.name, "reduce_scatter_sum", int(buf.ptr), int(shard * world)))

        def all_gather(self, buf, shard):
            import torch
            import torch.distributed as tdist
            world, rank = tdist.get_world_size(), tdist.get_rank()
            v = self.dev.view(buf.ptr, shard * world)
            parts = [torch.empty(shard, dtype=torch.float32) for _ in range(world)]
            tdist.all_gather(parts, torch.from_numpy(v[rank * shard:(rank + 1) * shard].copy()))
            for r, part in enumerate(parts):
                v[r * shard:(r + 1) * shard] = part.numpy()
            Shared.log.append((self.dev.name, "all_gather", int(buf.ptr), int(shard * world)))

        def rmsprop(self, p, g, acc, n, hyper, rho=0.9, eps=1e-6, grad_scale=1.0):
            pv, gv, av = (self.dev.view(t.ptr, n) for t in (p, g, acc))
            lr = self.dev.view(hyper.ptr, 1)[0]
            gs = gv * np.float32(grad_scale)
            av[:] = np.float32(rho) * av + np.float32(1 - rho) * gs * gs        # lasagne.updates.rmsprop
            pv[:] = pv - lr * gs / np.sqrt(av + np.float32(eps))
            Shared.log.append((self.dev.name, "rmsprop", int(p.ptr), int(n)))

        def __getattr__(self, name):
            def rec(*args, **kw):
                ptrs = tuple(int(a.ptr) for a in args if isinstance(a, DevTensor))
                Shared.log.append((self.dev.name, name) + ptrs)
            return rec

    class HostDevice:
        ops_class = HostOps
        shared = Shared

        def __init__(self, index=0):
            self.h, self.index = None, index
            self.name = "ctx%d" % Shared.nctx
            Shared.nctx += 1
            self.bytes_allocated = 0

        def alloc(self, nbytes):
            n = (int(max(nbytes, 16)) + 3) // 4
            base = (Shared.bases[-1] + 4 * Shared.arrays[Shared.bases[-1]].size + 1024) if Shared.bases else 1 << 20
            base = (base + 255) // 256 * 256
            Shared.bases.append(base)
            Shared.arrays[base] = np.zeros(n, np.float32)
            self.bytes_allocated += nbytes
            return base

        def free(self, ptr):
            pass

        def view(self, ptr, n):
            i = bisect.bisect_right(Shared.bases, ptr) - 1
            base = Shared.bases[i]
            off = (ptr - base) // 4
            arr = Shared.arrays[base]
            assert 0 <= off and off + n <= arr.size, "out-of-range host view"
            return arr[off:off + n]

        def empty(self, shape):
            shape = tuple(int(s) for s in shape)
            n = int(np.prod(shape))
            return DevTensor(self, self.alloc(4 * n), shape if len(shape) in (2, 4) else (1, n, 1, 1))

        zeros = empty

        def tensor(self, arr):
            arr = np.ascontiguousarray(arr, np.float32)
            t = self.empty(arr.shape if arr.ndim in (2, 4) else (1, arr.size, 1, 1))
            self.h2d(t.ptr, arr)
            return t

        def h2d(self, ptr, arr):
            a = np.ascontiguousarray(arr).view(np.uint8).ravel()
            self.view(ptr, (a.size + 3) // 4).view(np.uint8)[:a.size] = a

        def d2h(self, arr, ptr, nbytes):
            arr.reshape(-1).view(np.uint8)[:nbytes] = self.view(ptr, (nbytes + 3) // 4).view(np.uint8)[:nbytes]

        def memset_zero(self, ptr, nbytes):
            self.view(ptr, nbytes // 4)[:] = 0

        def sync(self):
            pass

        def wait_for(self, other):
            Shared.log.append((self.name, "wait_for", other.name))

        def event_create(self):
            Shared.nev = getattr(Shared, "nev", 0) + 1
            return "ev%d" % Shared.nev

        def event_record(self, ev):
            Shared.log.append((self.name, "event_record", ev))

        def event_wait(self, ev):
            Shared.log.append((self.name, "event_wait", ev))

        # ---- the input pipeline's surface (step.py upload_async / upload_resident_async): synchronous here, logged ----
        class pinned_array:
            def __init__(self, shape, dtype=np.float32):
                self.shape, self.array, self.ptr = tuple(shape), np.zeros(shape, dtype), 1

            def close(self):
                self.array = None

        @staticmethod
        def event_sync(ev):
            Shared.log.append(("host", "event_sync", ev))

        @staticmethod
        def event_destroy(ev):
            pass

        def h2d_async(self, ptr, pinned):
            self.h2d(ptr, pinned.array)
            Shared.log.append((self.name, "h2d_async", int(ptr)))

        def d2d(self, dst, src, nbytes):
            self.view(dst, nbytes // 4)[:] = self.view(src, nbytes // 4)
            Shared.log.append((self.name, "d2d", int(dst), int(src), int(nbytes)))

        def close(self):
            pass

    return HostDevice

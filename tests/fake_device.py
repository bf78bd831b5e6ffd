"""Host-only stand-ins for gan_heightmaps_amd.device.{Device, Ops}: fake HBM addresses and a recorder of the
op calls, so the lowering / placement / program-emission logic can be tested without a GPU.  No arithmetic."""
import numpy as np

from gan_heightmaps_amd.device import DevTensor


class FakeDevice:
    def __init__(self):
        self.h = None
        self._next = 1 << 20
        self.bytes_allocated = 0
        self.uploads = 0

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
        return self.empty(shape)

    def h2d(self, ptr, arr):
        self.uploads += 1

    def d2h(self, arr, ptr, nbytes):
        arr[...] = 0

    def memset_zero(self, ptr, nbytes):
        pass

    def sync(self):
        pass


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

    def __getattr__(self, name):
        def rec(*args, **kw):
            self.calls.append((name, args, kw))
        return rec

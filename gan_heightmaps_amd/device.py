"""Device context and tensors over libghm.so (thin, explicit; no torch)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ConvDesc, call

ACT_CODES = {'linear': 0, 'relu': 1, 'lrelu': 2, 'sigmoid': 3, 'tanh': 4}
DTYPE_CODES = {'f32': 0, 'bf16': 1, 'f16': 2}       # GHM_DTYPE_* (include/ghm.h)
# 'bf16x3': fp32 arithmetic on the bf16 matrix cores by operand splitting (csrc/conv_split.hip) -- a host-side name: Ops
# routes the *_lp / *_lp_q calls made with it to the ghm_*_split entry points, and a QTensor of this dtype is three planes
SPLIT = 'bf16x3'
DTYPE_CODES[SPLIT] = 3     # understood by the q-epilogue producers (csrc/elementwise_q.hip) only
# 'bf16x2': the same machinery with TWO pieces per operand and three products (x0 w0 + x1 w0 + x0 w1): 16-bit operands at half
# the matrix-core time of 'bf16x3' -- BASELINE config 4's arithmetic (plain bf16 operands miss north_star's 1e-3 on the outputs)
SPLIT2 = 'bf16x2'
DTYPE_CODES[SPLIT2] = 4
SPLITS = {SPLIT: 3, SPLIT2: 2}     # split dtypes -> pieces per operand (the ``pieces`` argument of the ghm_*_split entry points)


def device_count():
    n = C.c_int32(0)
    call("ghm_device_count", C.byref(n))
    return n.value


class DevTensor:
    """fp32 [N, C, H, W] view in HBM with an explicit sample stride (elements)."""
    __slots__ = ("dev", "ptr", "shape", "nstride", "base")

    def __init__(self, dev, ptr, shape, nstride=None, base=None):
        self.dev = dev
        self.ptr = int(ptr)
        shape = tuple(int(s) for s in shape)
        if len(shape) == 2:
            shape = (shape[0], shape[1], 1, 1)
        assert len(shape) == 4
        self.shape = shape
        self.nstride = int(nstride) if nstride is not None else shape[1] * shape[2] * shape[3]
        self.base = base

    N = property(lambda s: s.shape[0])
    Cc = property(lambda s: s.shape[1])
    H = property(lambda s: s.shape[2])
    W = property(lambda s: s.shape[3])
    HW = property(lambda s: s.shape[2] * s.shape[3])
    size = property(lambda s: s.shape[0] * s.shape[1] * s.shape[2] * s.shape[3])
    contiguous = property(lambda s: s.nstride == s.shape[1] * s.shape[2] * s.shape[3])

    def channels(self, c0, c1):
        """View of channels [c0, c1) (ConcatLayer axis=1 without a copy)."""
        assert 0 <= c0 < c1 <= self.shape[1]
        return DevTensor(self.dev, self.ptr + 4 * c0 * self.HW, (self.N, c1 - c0, self.H, self.W), self.nstride,
                         self.base if self.base is not None else self)

    def samples(self, n0, n1):
        assert 0 <= n0 < n1 <= self.shape[0]
        return DevTensor(self.dev, self.ptr + 4 * n0 * self.nstride, (n1 - n0,) + self.shape[1:], self.nstride,
                         self.base if self.base is not None else self)

    def reshape(self, shape):
        assert self.contiguous
        t = DevTensor(self.dev, self.ptr, shape, None, self.base if self.base is not None else self)
        assert t.size == self.size
        return t

    def numpy(self):
        n, c, h, w = self.shape
        out = np.empty(self.shape, np.float32)
        if self.contiguous:
            self.dev.d2h(out, self.ptr, out.nbytes)
        else:
            for i in range(n):
                self.dev.d2h(out[i], self.ptr + 4 * i * self.nstride, out[i].nbytes)
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, np.float32).reshape(self.shape)
        if self.contiguous:
            self.dev.h2d(self.ptr, arr)
        else:
            for i in range(self.N):
                self.dev.h2d(self.ptr + 4 * i * self.nstride, arr[i])
        return self


class QTensor:
    """bf16 / fp16 copy of a [N, C, H, W] tensor in HBM in the channel-block-of-8 layout of include/ghm.h ("q tensor"):
    16-byte units = 8 channels of one pixel; ``nstride`` counts UNITS between samples, so a channel slice (multiple of 8)
    of a wider buffer is a view."""
    __slots__ = ("dev", "ptr", "shape", "nstride", "dtype", "base", "pstride")

    def __init__(self, dev, ptr, shape, dtype, nstride=None, base=None, pstride=None):
        self.dev, self.ptr, self.dtype, self.base = dev, int(ptr), dtype, base
        self.shape = tuple(int(v) for v in shape)
        assert len(self.shape) == 4 and self.shape[1] % 8 == 0, self.shape
        self.nstride = int(nstride) if nstride is not None else self.shape[1] // 8 * self.shape[2] * self.shape[3]
        # 'bf16x3' (split fp32): units between the three piece planes -- a property of the allocation, kept by every view
        self.pstride = int(pstride) if pstride is not None else self.shape[0] * self.nstride

    N = property(lambda s: s.shape[0])
    Cc = property(lambda s: s.shape[1])
    H = property(lambda s: s.shape[2])
    W = property(lambda s: s.shape[3])
    HW = property(lambda s: s.shape[2] * s.shape[3])
    contiguous = property(lambda s: s.nstride == s.shape[1] // 8 * s.shape[2] * s.shape[3])
    nbytes = property(lambda s: 16 * s.nstride * s.shape[0])

    @staticmethod
    def empty(dev, shape, dtype):
        shape = tuple(int(v) for v in shape)
        planes = SPLITS.get(dtype, 1)
        return QTensor(dev, dev.alloc(planes * 16 * shape[0] * (shape[1] // 8) * shape[2] * shape[3]), shape, dtype)

    def channels(self, c0, c1):
        assert 0 <= c0 < c1 <= self.shape[1] and c0 % 8 == 0 and (c1 - c0) % 8 == 0
        return QTensor(self.dev, self.ptr + 16 * (c0 // 8) * self.HW, (self.N, c1 - c0, self.H, self.W), self.dtype,
                       self.nstride, self.base if self.base is not None else self, self.pstride)

    def samples(self, n0, n1):
        assert 0 <= n0 < n1 <= self.shape[0]
        return QTensor(self.dev, self.ptr + 16 * n0 * self.nstride, (n1 - n0,) + self.shape[1:], self.dtype, self.nstride,
                       self.base if self.base is not None else self, self.pstride)

    def reshape(self, shape):
        """[N, 4K, H, W] <-> [4N, K, H, W] (the parity-planar view of the collapsed up-sample convolutions)"""
        assert self.contiguous
        t = QTensor(self.dev, self.ptr, shape, self.dtype, None, self.base if self.base is not None else self, self.pstride)
        assert t.nbytes == self.nbytes
        return t

    def numpy(self, piece=None):
        """-> float32 [N, C, H, W] (the exact values of the stored halfwords); 'bf16x3': the sum of the three pieces (in
        float64, rounded once: the fp32 value that was split), or one piece"""
        if self.dtype in SPLITS and piece is None:
            return sum(self.numpy(p).astype(np.float64) for p in range(SPLITS[self.dtype])).astype(np.float32)
        raw = np.empty((self.N, self.Cc * self.HW), np.uint16)       # a sample's channel blocks are contiguous planes
        for n in range(self.N):
            self.dev.d2h(raw[n], self.ptr + 16 * ((piece or 0) * self.pstride + n * self.nstride), raw[n].nbytes)
        raw = raw.reshape(self.N, self.Cc // 8, self.HW, 8).transpose(0, 1, 3, 2)
        raw = np.ascontiguousarray(raw).reshape(self.shape)
        if self.dtype == 'bf16' or self.dtype in SPLITS:
            return (raw.astype(np.uint32) << 16).view(np.float32)
        return raw.view(np.float16).astype(np.float32)


class PinnedArray:
    """page-locked host memory seen as a numpy array (ghm_host_alloc): the source of asynchronous uploads"""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in shape)
        dt = np.dtype(dtype)
        n = int(np.prod(self.shape))
        p = C.c_void_p()
        call("ghm_host_alloc", max(n * dt.itemsize, 16), C.byref(p))
        self.ptr = p.value
        raw = np.ctypeslib.as_array(C.cast(C.c_void_p(self.ptr), C.POINTER(C.c_uint8)), shape=(max(n * dt.itemsize, 16),))
        self.array = raw[:n * dt.itemsize].view(dt).reshape(self.shape)

    def close(self):
        if self.ptr:
            self.array = None
            call("ghm_host_free", C.c_void_p(self.ptr))
            self.ptr = None


class Device:
    def __init__(self, index=0):
        _lib.load()
        h = C.c_void_p()
        call("ghm_ctx_create", int(index), C.byref(h))
        self.h = h
        self.index = index
        self._allocs = {}
        self.bytes_allocated = 0

    @classmethod
    def with_priority(cls, index, priority):
        """a context whose stream has the given HIP priority (negative = more urgent)"""
        self = cls.__new__(cls)
        _lib.load()
        h = C.c_void_p()
        call("ghm_ctx_create_prio", int(index), int(priority), C.byref(h))
        self.h, self.index, self._allocs, self.bytes_allocated = h, index, {}, 0
        return self

    # ---- memory ----
    def alloc(self, nbytes):
        p = C.c_void_p()
        call("ghm_alloc", self.h, int(max(nbytes, 16)), C.byref(p))
        self._allocs[p.value] = nbytes
        self.bytes_allocated += nbytes
        return p.value

    def free(self, ptr):
        if ptr in self._allocs:
            self.bytes_allocated -= self._allocs.pop(ptr)
            call("ghm_free", self.h, C.c_void_p(ptr))

    def empty(self, shape):
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape)) if len(shape) else 1
        return DevTensor(self, self.alloc(4 * n), shape if len(shape) in (2, 4) else (1, n, 1, 1))

    def zeros(self, shape):
        t = self.empty(shape)
        self.memset_zero(t.ptr, 4 * t.size)
        return t

    def tensor(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        shape = arr.shape
        if arr.ndim == 1:
            shape = (1, arr.shape[0], 1, 1)
        elif arr.ndim == 0:
            shape = (1, 1, 1, 1)
        elif arr.ndim == 3:
            shape = (1,) + arr.shape
        t = self.empty(shape)
        self.h2d(t.ptr, arr)
        return t

    def h2d(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        call("ghm_h2d", self.h, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes)

    def queue_interference(self, other, probe, spin_us=1500):
        """microseconds a trivial kernel on ``other`` needs to finish while THIS context's stream sits in a wait for a
        ~spin_us spin kernel on ``probe``: ~spin_us if the two streams share a hardware queue, a few microseconds if not"""
        us = C.c_float()
        call("ghm_queue_interference", self.h, other.h, probe.h, int(spin_us), C.byref(us))
        return us.value

    def event_create(self):
        e = C.c_void_p()
        call("ghm_event_create", self.h, C.byref(e))
        return e

    def event_record(self, ev):
        call("ghm_event_record", self.h, ev)

    def event_wait(self, ev):
        """later work on this context's stream waits for the point ``ev`` marks (on whichever stream it was recorded)"""
        call("ghm_event_wait", self.h, ev)

    @staticmethod
    def event_sync(ev):
        call("ghm_event_sync", ev)

    @staticmethod
    def event_destroy(ev):
        call("ghm_event_destroy", ev)

    def h2d_async(self, ptr, pinned):
        """copy a PinnedArray to the device, ordered on this context's stream, without waiting for it"""
        call("ghm_h2d_async", self.h, C.c_void_p(ptr), C.c_void_p(pinned.ptr), pinned.array.nbytes)

    def d2h(self, arr, ptr, nbytes):
        assert arr.flags['C_CONTIGUOUS']
        call("ghm_d2h", self.h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), int(nbytes))

    def d2d(self, dst, src, nbytes):
        call("ghm_d2d", self.h, C.c_void_p(dst), C.c_void_p(src), int(nbytes))

    def memset_zero(self, ptr, nbytes):
        call("ghm_memset_zero", self.h, C.c_void_p(ptr), int(nbytes))

    def sync(self):
        call("ghm_sync", self.h)

    def scratch_info(self):
        """(bytes of the library workspace, bytes of retired blocks, recorded steps / graphs pinning them)"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int32()
        call("ghm_scratch_info", self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def set_loss_scale_state(self, state):
        """attach (DevTensor of 8 floats) / detach (None) the dynamic loss-scale state of this context"""
        call("ghm_set_loss_scale_state", self.h, _vp(state))

    def wait_for(self, other):
        """later work on this context's stream waits for everything already enqueued on ``other``'s stream"""
        call("ghm_stream_wait", self.h, other.h)

    def info(self):
        name = C.create_string_buffer(256)
        cu, hbm = C.c_int32(), C.c_int64()
        call("ghm_device_info", self.h, name, 256, C.byref(cu), C.byref(hbm))
        return {"name": name.value.decode(), "num_cu": cu.value, "hbm_bytes": hbm.value}

    # ---- graph capture / timers ----
    def capture_begin(self):
        call("ghm_capture_begin", self.h)

    def capture_end(self):
        g = C.c_void_p()
        call("ghm_capture_end", self.h, C.byref(g))
        return g

    def graph_launch(self, g):
        call("ghm_graph_launch", self.h, g)

    def graph_destroy(self, g):
        call("ghm_graph_destroy", g)

    @staticmethod
    def step_build(stages):
        """stages: [(Device, graph handle)] -> step handle: ghm_step_run launches every stage graph in one call"""
        n = len(stages)
        ctxs = (C.c_void_p * n)(*[d.h for d, _ in stages])
        graphs = (C.c_void_p * n)(*[g for _, g in stages])
        st = C.c_void_p()
        call("ghm_step_build", n, ctxs, graphs, C.byref(st))
        return st

    @staticmethod
    def step_record_begin(devs):
        """every operation issued on ``devs`` from now on is appended to the returned step instead of being executed
        (ghm_step_record_begin); end with step_record_end, replay with step_run"""
        n = len(devs)
        ctxs = (C.c_void_p * n)(*[d.h for d in devs])
        st = C.c_void_p()
        call("ghm_step_record_begin", n, ctxs, C.byref(st))
        return st

    @staticmethod
    def step_record_end(st):
        call("ghm_step_record_end", st)

    @staticmethod
    def step_timer_stride(st, stride):
        call("ghm_step_timer_stride", st, int(stride))

    @staticmethod
    def step_run(st):
        call("ghm_step_run", st)

    @staticmethod
    def step_destroy(st):
        call("ghm_step_destroy", st)

    def timer_start(self, slot=0):
        call("ghm_timer_start", self.h, slot)

    def timer_stop(self, slot=0):
        call("ghm_timer_stop", self.h, slot)

    def timer_ms(self, slot=0):
        ms = C.c_float()
        call("ghm_timer_elapsed_ms", self.h, slot, C.byref(ms))
        return ms.value

    def close(self):
        if self.h is not None:
            for p in list(self._allocs):
                self.free(p)
            call("ghm_ctx_destroy", self.h)
            self.h = None


# ---- weight layout (include/ghm.h): wp[c][a*kw+b][k] = W[k][c][kh-1-a][kw-1-b] -------------------

def pack_conv_w(W):
    """lasagne Conv2DLayer W[K, C, kh, kw] (or Deconv2DLayer W[Cin=K, Cout=C, kh, kw]) -> packed [C, kh*kw, K]."""
    K, Cc, kh, kw = W.shape
    return np.ascontiguousarray(W[:, :, ::-1, ::-1].transpose(1, 2, 3, 0)).reshape(Cc, kh * kw, K).astype(np.float32)


def unpack_conv_w(wp, K, Cc, kh, kw):
    return np.ascontiguousarray(np.asarray(wp).reshape(Cc, kh, kw, K).transpose(3, 0, 1, 2)[:, :, ::-1, ::-1])


def conv_desc(N, Cc, H, W, K, kh, kw, stride, pad, x_nstride=None, y_nstride=None):
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    return ConvDesc(N, Cc, H, W, K, Ho, Wo, kh, kw, stride, pad,
                    x_nstride if x_nstride is not None else Cc * H * W,
                    y_nstride if y_nstride is not None else K * Ho * Wo)


def _vp(x):
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, DevTensor):
        return C.c_void_p(x.ptr)
    return C.c_void_p(int(x))


class Ops:
    """Python spellings of the C-ABI ops on DevTensors (each is one asynchronous call)."""

    def __init__(self, dev):
        self.dev = dev
        self.h = dev.h

    def conv2d_fwd(self, d, x, wp, bias, y, act='linear', alpha=0.0, accumulate=False):
        call("ghm_conv2d_fwd", self.h, C.byref(d), _vp(x), _vp(wp), _vp(bias), _vp(y), ACT_CODES[act], alpha,
             int(accumulate))

    def conv2d_dgrad(self, d, dy, wp, dx, bias=None, act='linear', alpha=0.0, accumulate=False):
        call("ghm_conv2d_dgrad", self.h, C.byref(d), _vp(dy), _vp(wp), _vp(bias), _vp(dx), ACT_CODES[act], alpha,
             int(accumulate))

    def dgrad_t_supported(self, d):
        return bool(_lib.load().ghm_dgrad_t_supported(C.byref(d)))

    def transpose_weights(self, d, wp, wpT):
        call("ghm_conv2d_transpose_weights", self.h, C.byref(d), _vp(wp), _vp(wpT))

    def collapse_table(self, items):
        """items: [(wp5, bias, wpc, bias4 DevTensors, C, K[, mode])] -> (device table ptr, n, total blocks) for
        upconv_collapse_batched (uploaded once: the pointers are fixed for the life of a plan); mode 1 = the bilinear form"""
        rec = np.zeros(len(items), dtype=[('wp5', '<u8'), ('bias', '<u8'), ('wpc', '<u8'), ('b4', '<u8'), ('C', '<i4'),
                                          ('K', '<i4'), ('b0', '<i4'), ('pad', '<i4')])
        b0 = 0
        for i, (w5, b, wpc, b4, Cc, K, *mode) in enumerate(items):
            rec[i] = (w5.ptr, b.ptr if b is not None else 0, wpc.ptr, b4.ptr if b4 is not None else 0, Cc, K, b0, mode[0] if mode else 0)
            b0 += (36 * Cc * K + 4 * K + 255) // 256
        ptr = self.dev.alloc(max(rec.nbytes, 40))
        self.dev.h2d(ptr, rec.view(np.uint8))
        return ptr, len(items), b0

    def upconv_collapse_batched(self, table):
        ptr, n, blocks = table
        call("ghm_upconv_collapse_batched", self.h, C.c_void_p(ptr), n, blocks)

    def expand_table(self, items):
        """items: [(dwpc, dwp5 DevTensors, C, K)] -> (device table ptr, n, total blocks) for upconv_expand_batched"""
        rec = np.zeros(len(items), dtype=[('dwpc', '<u8'), ('dwp5', '<u8'), ('C', '<i4'), ('K', '<i4'), ('b0', '<i4'),
                                          ('pad', '<i4')])
        b0 = 0
        for i, (dwpc, dwp5, Cc, K, *mode) in enumerate(items):
            rec[i] = (dwpc.ptr, dwp5.ptr, Cc, K, b0, mode[0] if mode else 0)
            b0 += (25 * Cc * K + 255) // 256
        ptr = self.dev.alloc(max(rec.nbytes, 32))
        self.dev.h2d(ptr, rec.view(np.uint8))
        return ptr, len(items), b0

    def upconv_expand_batched(self, table, accumulate=False):
        ptr, n, blocks = table
        call("ghm_upconv_expand_batched", self.h, C.c_void_p(ptr), n, blocks, int(accumulate))

    # ---- BilinearUpsample2DLayer(2) -> 3x3 conv: the frame the collapsed convolution leaves out (csrc/conv_bilinear.hip) ----
    def blconv_supported(self, N, Cc, K, n1, n2):
        return bool(_lib.load().ghm_blconv_supported(int(N), int(Cc), int(K), int(n1), int(n2)))

    def blconv_frame_sizes(self, N, Cc, K, n1, n2):
        a, b = C.c_int64(), C.c_int64()
        call("ghm_blconv_frame_sizes", int(N), int(Cc), int(K), int(n1), int(n2), C.byref(a), C.byref(b))
        return a.value, b.value

    def blconv_frame_fwd(self, x, wp, y_pp, K, FL):
        """y_pp [N, 4K, n1, n2] += conv3x3(frame of x); x: the coarse fp32 input; wp: the layer's packed 3x3 weights"""
        call("ghm_blconv_frame_fwd", self.h, _vp(x), x.nstride, _vp(wp), _vp(y_pp), y_pp.nstride, x.N, x.Cc, int(K), x.H, x.W, _vp(FL))

    def blconv_frame_gather(self, dy_pp, Cc, K, DYL):
        call("ghm_blconv_frame_gather", self.h, _vp(dy_pp), dy_pp.nstride, dy_pp.N, int(Cc), int(K), dy_pp.H, dy_pp.W, _vp(DYL))

    def blconv_frame_dgrad(self, DYL, wp, dx, K):
        call("ghm_blconv_frame_dgrad", self.h, _vp(DYL), _vp(wp), _vp(dx), dx.nstride, dx.N, dx.Cc, int(K), dx.H, dx.W)

    def blconv_frame_wgrad(self, DYL, FL, dwp, N, Cc, K, n1, n2):
        call("ghm_blconv_frame_wgrad", self.h, _vp(DYL), _vp(FL), _vp(dwp), int(N), int(Cc), int(K), int(n1), int(n2))

    def blconv_split_supported(self, d, kind, dtype):
        """may the collapsed bilinear convolution ``d`` (4K filters) run with its structural zero taps skipped?"""
        return dtype in SPLITS and bool(_lib.load().ghm_blconv_split_supported(C.byref(d), int(kind)))

    def blconv_fwd_split(self, d, xq, wq, bias, y, dtype):
        call("ghm_blconv_fwd_split", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, xq.pstride, _vp(wq), _vp(bias), _vp(y),
             SPLITS[dtype])

    def blconv_dgrad_split(self, d, dyq, wqT, dx, dtype, accumulate=False):
        call("ghm_blconv_dgrad_split", self.h, C.byref(d), C.c_void_p(dyq.ptr), dyq.nstride, dyq.pstride, _vp(wqT), _vp(dx),
             int(accumulate), SPLITS[dtype])

    def blconv_wgrad_split(self, d, xq, dyq, dwp, ws, dtype, accumulate=False):
        call("ghm_blconv_wgrad_split", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, xq.pstride, C.c_void_p(dyq.ptr),
             dyq.nstride, dyq.pstride, _vp(dwp), _vp(ws), int(accumulate), SPLITS[dtype])

    def transpose_table(self, items):
        """items: [(wp DevTensor, wpT DevTensor, C, T, K)] -> (device table ptr, n, total blocks) for
        transpose_weights_batched (uploaded once: the pointers are fixed for the life of a plan)"""
        rec = np.zeros(len(items), dtype=[('wp', '<u8'), ('wpT', '<u8'), ('C', '<i4'), ('T', '<i4'), ('K', '<i4'),
                                          ('b0', '<i4')])
        b0 = 0
        for i, (wp, wpT, Cc, T, K) in enumerate(items):
            rec[i] = (wp.ptr, wpT.ptr, Cc, T, K, b0)
            b0 += T * ((Cc + 31) // 32) * ((K + 31) // 32)
        ptr = self.dev.alloc(max(rec.nbytes, 32))
        self.dev.h2d(ptr, rec.view(np.uint8))
        return ptr, len(items), b0

    def transpose_weights_batched(self, table):
        ptr, n, blocks = table
        call("ghm_transpose_weights_batched", self.h, C.c_void_p(ptr), n, blocks)

    def conv2d_dgrad_t(self, d, dy, wpT, dx, bias=None, act='linear', alpha=0.0, accumulate=False):
        call("ghm_conv2d_dgrad_t", self.h, C.byref(d), _vp(dy), _vp(wpT), _vp(bias), _vp(dx), ACT_CODES[act], alpha,
             int(accumulate))

    def dgrad_dact_supported(self, d, dtype='f32'):
        """0: not served; 1 / 2 / 3: served, reading the fp32 packed wp / the fp32 wpT / the low-precision wqT"""
        if dtype in SPLITS:
            if _lib.load().ghm_split_dgrad_dact_supported(C.byref(d)):
                return 3
            dtype = 'f32'
        return int(_lib.load().ghm_dgrad_dact_supported(C.byref(d), DTYPE_CODES[dtype]))

    def conv2d_dgrad_dact(self, d, dy, w, dx, y, act, alpha, dtype='f32'):
        """dx = conv^T(dy) * act'(y): the data gradient with the producer's activation backward in its epilogue"""
        assert y.shape == dx.shape
        if dtype in SPLITS:
            # a split weight pack has no fp32-operand form of this product: ghm_conv2d_dgrad_dact would read it as an fp32 /
            # low-precision pack.  The split data gradient with the activation backward takes the q copy of dy
            raise ValueError("conv2d_dgrad_dact(dtype='bf16x3'): use conv2d_dgrad_dact_lp_q with the split q tensor of dy")
        call("ghm_conv2d_dgrad_dact", self.h, C.byref(d), _vp(dy), _vp(w), _vp(dx), _vp(y), y.nstride, ACT_CODES[act], alpha,
             DTYPE_CODES[dtype])

    def wgrad_workspace(self, d):
        n = C.c_size_t()
        call("ghm_conv2d_wgrad_workspace", C.byref(d), C.byref(n))
        return n.value

    def conv2d_wgrad(self, d, x, dy, dwp, ws, accumulate=False):
        call("ghm_conv2d_wgrad", self.h, C.byref(d), _vp(x), _vp(dy), _vp(dwp), _vp(ws), int(accumulate))

    # ---- bf16 / fp16 matrix-core convolutions (fp32 tensors in HBM; include/ghm.h GHM_DTYPE_*) ----
    def lp_supported(self, d, kind, dtype):
        if dtype in SPLITS:
            return self.split_supported(d, kind)
        return bool(_lib.load().ghm_lp_supported(C.byref(d), int(kind), DTYPE_CODES[dtype]))

    # ---- fp32 on the bf16 matrix cores by operand splitting (csrc/conv_split.hip; opt-in) ----
    def split_supported(self, d, kind):
        return bool(_lib.load().ghm_split_supported(C.byref(d), int(kind)))

    def split_weight_bytes(self, d, transposed=False, pieces=3):
        n = C.c_size_t()
        call("ghm_split_weight_bytes", C.byref(d), int(transposed), C.byref(n), pieces)
        return n.value

    def split_pack_weights(self, d, wp, wq, transposed=False, pieces=3):
        call("ghm_split_pack_weights", self.h, C.byref(d), _vp(wp), _vp(wq), int(transposed), pieces)

    def split_pack(self, x, q_ptr, q_nstride=None, q_pstride=None, pieces=3):
        """fp32 view -> split q tensor (``pieces`` planes) at q_ptr; returns (nstride, pstride) in 16-byte units"""
        ns = (x.Cc // 8) * x.HW if q_nstride is None else q_nstride
        ps = x.N * ns if q_pstride is None else q_pstride
        call("ghm_split_pack", self.h, _vp(x), x.nstride, x.N, x.Cc, x.HW, C.c_void_p(int(q_ptr)), ns, ps, pieces)
        return ns, ps

    def split_q_direct(self, d, kind):
        return bool(_lib.load().ghm_split_q_direct(C.byref(d), int(kind)))

    def conv2d_fwd_split(self, d, x, wq, bias, y, act='linear', alpha=0.0, accumulate=False, xq=None, yq=None, pieces=3):
        """xq = (ptr, nstride, pstride) of the input already split, or None (the entry point splits x); yq: a whole split
        QTensor (or a channel slice of one) that also receives the result; y may then be None"""
        q = xq or (None, 0, 0)
        assert yq is None or yq.pstride == yq.N * yq.nstride
        call("ghm_conv2d_fwd_split", self.h, C.byref(d), _vp(x), C.c_void_p(int(q[0])) if q[0] else None, q[1], q[2],
             _vp(wq), _vp(bias), _vp(y), C.c_void_p(yq.ptr) if yq is not None else None, yq.nstride if yq is not None else 0,
             ACT_CODES[act], alpha, int(accumulate), pieces)

    def conv2d_dgrad_split(self, d, dy, wqT, dx, bias=None, act='linear', alpha=0.0, accumulate=False, dyq=None, dxq=None,
                           pieces=3):
        q = dyq or (None, 0, 0)
        assert dxq is None or dxq.pstride == dxq.N * dxq.nstride
        call("ghm_conv2d_dgrad_split", self.h, C.byref(d), _vp(dy), C.c_void_p(int(q[0])) if q[0] else None, q[1], q[2],
             _vp(wqT), _vp(bias), _vp(dx), C.c_void_p(dxq.ptr) if dxq is not None else None,
             dxq.nstride if dxq is not None else 0, ACT_CODES[act], alpha, int(accumulate), pieces)

    def lp_weight_bytes(self, d, transposed=False, dtype=None):
        if dtype in SPLITS:
            return self.split_weight_bytes(d, transposed, SPLITS[dtype])
        n = C.c_size_t()
        call("ghm_lp_weight_bytes", C.byref(d), int(transposed), C.byref(n))
        return n.value

    def lp_pack_weights(self, d, wp, wq, dtype, transposed=False):
        if dtype in SPLITS:
            return self.split_pack_weights(d, wp, wq, transposed, SPLITS[dtype])
        call("ghm_lp_pack_weights", self.h, C.byref(d), _vp(wp), _vp(wq), DTYPE_CODES[dtype], int(transposed))

    def lp_pack_table(self, items):
        """items: [(wp DevTensor | ptr, wq ptr, red, T, rows, transposed)] -> (device table ptr, n, total blocks) for
        lp_pack_batched (uploaded once: the pointers are fixed for the life of a plan)"""
        rec = np.zeros(len(items), dtype=[('wp', '<u8'), ('wq', '<u8'), ('red', '<i4'), ('T', '<i4'), ('rows', '<i4'),
                                          ('nblk', '<i4'), ('rpad', '<i4'), ('tr', '<i4'), ('b0', '<i4'), ('pad', '<i4')])
        b0 = 0
        for i, (wp, wq, red, T, rows, tr) in enumerate(items):
            nblk, rpad = (red + 15) // 16 * 2, (rows + 127) // 128 * 128
            rec[i] = (wp.ptr if isinstance(wp, DevTensor) else int(wp), int(wq), red, T, rows, nblk, rpad, int(tr), b0, 0)
            b0 += (nblk * T * rpad + 255) // 256
        ptr = self.dev.alloc(max(rec.nbytes, 48))
        self.dev.h2d(ptr, rec.view(np.uint8))
        return ptr, len(items), b0

    def lp_pack_batched(self, table, dtype):
        ptr, n, blocks = table
        if dtype in SPLITS:
            return call("ghm_split_pack_batched", self.h, C.c_void_p(ptr), n, blocks, SPLITS[dtype])
        call("ghm_lp_pack_batched", self.h, C.c_void_p(ptr), n, blocks, DTYPE_CODES[dtype])

    def conv2d_fwd_lp(self, d, x, wq, bias, y, dtype, act='linear', alpha=0.0, accumulate=False):
        if dtype in SPLITS:
            return self.conv2d_fwd_split(d, x, wq, bias, y, act, alpha, accumulate, pieces=SPLITS[dtype])
        call("ghm_conv2d_fwd_lp", self.h, C.byref(d), _vp(x), _vp(wq), _vp(bias), _vp(y), ACT_CODES[act], alpha,
             int(accumulate), DTYPE_CODES[dtype])

    def conv2d_dgrad_lp(self, d, dy, wqT, dx, dtype, bias=None, act='linear', alpha=0.0, accumulate=False):
        if dtype in SPLITS:
            return self.conv2d_dgrad_split(d, dy, wqT, dx, bias, act, alpha, accumulate, pieces=SPLITS[dtype])
        call("ghm_conv2d_dgrad_lp", self.h, C.byref(d), _vp(dy), _vp(wqT), _vp(bias), _vp(dx), ACT_CODES[act], alpha,
             int(accumulate), DTYPE_CODES[dtype])

    def wgrad_lp_workspace(self, d):
        n, m = C.c_size_t(), C.c_size_t()
        call("ghm_conv2d_wgrad_lp_workspace", C.byref(d), C.byref(n))
        call("ghm_conv2d_wgrad_split_workspace", C.byref(d), C.byref(m))
        return max(n.value, m.value)

    def conv2d_wgrad_lp(self, d, x, dy, dwp, ws, dtype, accumulate=False):
        call("ghm_conv2d_wgrad_lp", self.h, C.byref(d), _vp(x), _vp(dy), _vp(dwp), _vp(ws), int(accumulate),
             DTYPE_CODES[dtype])

    # ---- q tensors (include/ghm.h): low-precision products on operands rounded once at their producer ----
    def q_pack(self, x, q):
        assert x.shape == q.shape
        if q.dtype in SPLITS:
            return self.split_pack(x, q.ptr, q.nstride, q.pstride, SPLITS[q.dtype])
        call("ghm_q_pack", self.h, _vp(x), x.nstride, x.N, x.Cc, x.HW, C.c_void_p(q.ptr), q.nstride, DTYPE_CODES[q.dtype])

    def q_unpack(self, q, x):
        assert x.shape == q.shape
        call("ghm_q_unpack", self.h, C.c_void_p(q.ptr), q.nstride, q.N, q.Cc, q.HW, _vp(x), x.nstride, DTYPE_CODES[q.dtype])

    def lp_q_direct(self, d, kind, dtype):
        if dtype in SPLITS:
            return self.split_q_direct(d, kind)
        return bool(_lib.load().ghm_lp_q_direct(C.byref(d), int(kind), DTYPE_CODES[dtype]))

    def conv_variant_lp(self, d, kind, dtype):
        """kernel family serving low-precision product ``kind`` (0 forward, 1 data gradient, 2 weight gradient)"""
        if dtype in SPLITS:
            w = d.W if kind == 1 and d.stride == 1 else d.Wo          # width of the pixel grid the kernel tiles
            if kind == 1 and d.stride == 2:
                return "sp_dgrad_s2_kernel"
            return ("sp_wgrad_kernel<%d, %d>%s" if kind == 2 else "sp_conv2_kernel<%d, %d>%s") % (
                d.kh, d.stride, "" if w % 32 == 0 else " narrow")
        out = C.create_string_buffer(128)
        call("ghm_lp_variant", C.byref(d), int(kind), DTYPE_CODES[dtype], out, 128)
        return out.value.decode()

    def conv_bn_fused_supported(self, d, dtype):
        if dtype in SPLITS:
            return False
        if dtype == 'f32':
            return bool(_lib.load().ghm_conv_bn_fused_supported_f32(C.byref(d)))
        return bool(_lib.load().ghm_conv_bn_fused_supported(C.byref(d), DTYPE_CODES[dtype]))

    def conv2d_bn_fwd(self, d, x, wp, bias, conv_out, y, gamma, beta, mean, inv, run_mean, run_inv, eps, run_alpha,
                      act='linear', alpha=0.0):
        """fp32: y = act(bn(conv(x) + bias)) with batch statistics; conv_out keeps conv(x) + bias for the backward"""
        assert conv_out.nstride == d.y_nstride
        call("ghm_conv2d_bn_fwd", self.h, C.byref(d), _vp(x), _vp(wp), _vp(bias), _vp(conv_out), _vp(y), y.nstride, _vp(gamma),
             _vp(beta), _vp(mean), _vp(inv), _vp(run_mean), _vp(run_inv), eps, run_alpha, ACT_CODES[act], alpha)

    def conv2d_bn_fwd_lp_q(self, d, xq, wq, bias, conv_out, y, yq, gamma, beta, mean, inv, run_mean, run_inv, eps, run_alpha,
                           dtype, act='linear', alpha=0.0):
        """y / yq = act(bn(conv(xq) + bias)) with batch statistics; conv_out (fp32) keeps conv(xq) + bias for the backward"""
        assert conv_out.nstride == d.y_nstride
        call("ghm_conv2d_bn_fwd_lp_q", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, _vp(wq), _vp(bias), _vp(conv_out),
             _vp(y), y.nstride if y is not None else 0, C.c_void_p(yq.ptr if yq is not None else 0),
             yq.nstride if yq is not None else 0, _vp(gamma), _vp(beta), _vp(mean), _vp(inv), _vp(run_mean), _vp(run_inv),
             eps, run_alpha, ACT_CODES[act], alpha, DTYPE_CODES[dtype])

    def conv2d_fwd_lp_q(self, d, xq, wq, bias, y, yq, dtype, act='linear', alpha=0.0, accumulate=False):
        if dtype in SPLITS:
            return self.conv2d_fwd_split(d, None, wq, bias, y, act, alpha, accumulate, xq=(xq.ptr, xq.nstride, xq.pstride), yq=yq,
                                         pieces=SPLITS[dtype])
        call("ghm_conv2d_fwd_lp_q", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, _vp(wq), _vp(bias), _vp(y),
             C.c_void_p(yq.ptr if yq is not None else 0), yq.nstride if yq is not None else 0, ACT_CODES[act], alpha,
             int(accumulate), DTYPE_CODES[dtype])

    def conv2d_dgrad_lp_q(self, d, dyq, wqT, dx, dxq, dtype, bias=None, act='linear', alpha=0.0, accumulate=False):
        if dtype in SPLITS:
            return self.conv2d_dgrad_split(d, None, wqT, dx, bias, act, alpha, accumulate,
                                           dyq=(dyq.ptr, dyq.nstride, dyq.pstride), dxq=dxq, pieces=SPLITS[dtype])
        call("ghm_conv2d_dgrad_lp_q", self.h, C.byref(d), C.c_void_p(dyq.ptr), dyq.nstride, _vp(wqT), _vp(bias), _vp(dx),
             C.c_void_p(dxq.ptr if dxq is not None else 0), dxq.nstride if dxq is not None else 0, ACT_CODES[act], alpha,
             int(accumulate), DTYPE_CODES[dtype])

    def conv2d_dgrad_dact_lp_q(self, d, dyq, wqT, dx, dxq, y, act, alpha, dtype):
        if dtype in SPLITS:
            assert dxq is None or dxq.pstride == dxq.N * dxq.nstride
            if isinstance(y, QTensor):          # the slope from the sign of the producer's q copy (its first piece plane)
                assert y.dtype == dtype and y.shape[1:] == (d.C, d.H, d.W)
                return call("ghm_conv2d_dgrad_dact_split_q", self.h, C.byref(d), C.c_void_p(dyq.ptr), dyq.nstride, dyq.pstride,
                            _vp(wqT), _vp(dx), C.c_void_p(dxq.ptr) if dxq is not None else None,
                            dxq.nstride if dxq is not None else 0, C.c_void_p(y.ptr), y.nstride, ACT_CODES[act], alpha,
                            SPLITS[dtype])
            return call("ghm_conv2d_dgrad_dact_split", self.h, C.byref(d), C.c_void_p(dyq.ptr), dyq.nstride, dyq.pstride,
                        _vp(wqT), _vp(dx), C.c_void_p(dxq.ptr) if dxq is not None else None,
                        dxq.nstride if dxq is not None else 0, _vp(y), y.nstride, ACT_CODES[act], alpha, SPLITS[dtype])
        call("ghm_conv2d_dgrad_dact_lp_q", self.h, C.byref(d), C.c_void_p(dyq.ptr), dyq.nstride, _vp(wqT), _vp(dx),
             C.c_void_p(dxq.ptr if dxq is not None else 0), dxq.nstride if dxq is not None else 0, _vp(y), y.nstride,
             ACT_CODES[act], alpha, DTYPE_CODES[dtype])

    @staticmethod
    def _whole_planes(q):
        """the q epilogues of the element-wise producers place the piece planes of a split q tensor N x nstride units apart: a
        sample-sliced view (which keeps the allocation's pstride) would have pieces 1 and 2 written over other samples' data"""
        assert q is None or q.dtype not in SPLITS or q.pstride == q.N * q.nstride, \
            "split q tensor: a sample slice cannot be the target of an element-wise q epilogue"

    def bn_apply_q(self, x, y, mean, inv, gamma, beta, yq, act='linear', alpha=0.0):
        self._whole_planes(yq)
        call("ghm_bn_apply_q", self.h, _vp(x), x.nstride, _vp(y), y.nstride if y is not None else 0, x.N, x.Cc, x.HW, _vp(mean),
             _vp(inv), _vp(gamma), _vp(beta), ACT_CODES[act], alpha, C.c_void_p(yq.ptr), yq.nstride, DTYPE_CODES[yq.dtype])

    def bn_backward_q(self, dout, y, x, dx, mean, inv, gamma, dgamma, dbeta, ws, dxq, act='linear', alpha=0.0, accumulate=False,
                      beta=None):
        """y may be None when beta is given: the layer output is recomputed from x"""
        self._whole_planes(dxq)
        call("ghm_bn_backward_q", self.h, _vp(dout), dout.nstride, _vp(y), y.nstride if y is not None else 0, _vp(x), x.nstride,
             _vp(dx), dx.nstride if dx is not None else 0, x.N, x.Cc, x.HW, _vp(mean), _vp(inv), _vp(gamma), _vp(beta),
             _vp(dgamma), _vp(dbeta), ACT_CODES[act], alpha, int(accumulate), _vp(ws), C.c_void_p(dxq.ptr), dxq.nstride,
             DTYPE_CODES[dxq.dtype])

    def bn_backward_x(self, dout, x, dx, mean, inv, gamma, beta, dgamma, dbeta, ws, act='linear', alpha=0.0, accumulate=False):
        """ghm_bn_backward without reading y (recomputed from x, bit-identical to the forward pass's value)"""
        call("ghm_bn_backward_x", self.h, _vp(dout), dout.nstride, _vp(x), x.nstride, _vp(dx), dx.nstride, x.N, x.Cc, x.HW,
             _vp(mean), _vp(inv), _vp(gamma), _vp(beta), _vp(dgamma), _vp(dbeta), ACT_CODES[act], alpha, int(accumulate), _vp(ws))

    def upsample_bilinear2_fwd_q(self, x, y, yq):
        assert y is None or y.contiguous
        self._whole_planes(yq)
        call("ghm_upsample_bilinear2_fwd_q", self.h, _vp(x), x.nstride, _vp(y), x.N, x.Cc, x.H, x.W, C.c_void_p(yq.ptr),
             yq.nstride, DTYPE_CODES[yq.dtype])

    def pp_to_hi_q(self, pp, hi, hiq):
        self._whole_planes(hiq)
        assert pp.contiguous and pp.N == 4 * hiq.N
        call("ghm_pp_to_hi_q", self.h, _vp(pp), _vp(hi), hi.nstride if hi is not None else 0, hiq.N, pp.Cc, pp.H, pp.W,
             C.c_void_p(hiq.ptr), hiq.nstride, DTYPE_CODES[hiq.dtype])

    def bn_apply_hi(self, x_pp, hi, hiq, mean, inv, gamma, beta, act='linear', alpha=0.0):
        """BatchNorm + activation of a parity-planar tensor written straight in the interleaved layout (hi and / or hiq)"""
        self._whole_planes(hiq)
        assert x_pp.contiguous and (hi is not None or hiq is not None)
        call("ghm_bn_apply_hi", self.h, _vp(x_pp), _vp(hi), hi.nstride if hi is not None else 0, x_pp.N // 4, x_pp.Cc, x_pp.H,
             x_pp.W, _vp(mean), _vp(inv), _vp(gamma), _vp(beta), ACT_CODES[act], alpha,
             C.c_void_p(hiq.ptr if hiq is not None else 0), hiq.nstride if hiq is not None else 0,
             DTYPE_CODES[hiq.dtype] if hiq is not None else 0)

    def bn_backward_hi(self, dhi, x_pp, dx_pp, dxq, mean, inv, gamma, beta, dgamma, dbeta, ws, act='linear', alpha=0.0,
                       accumulate=False):
        """backward of bn_apply_hi: dhi in the interleaved layout, dx (fp32 and / or q) parity-planar"""
        self._whole_planes(dxq)
        assert x_pp.contiguous and (dx_pp is None or dx_pp.contiguous) and (dx_pp is not None or dxq is not None)
        assert dxq is None or dxq.contiguous
        call("ghm_bn_backward_hi", self.h, _vp(dhi), dhi.nstride, _vp(x_pp), _vp(dx_pp), x_pp.N // 4, x_pp.Cc, x_pp.H, x_pp.W,
             _vp(mean), _vp(inv), _vp(gamma), _vp(beta), _vp(dgamma), _vp(dbeta), ACT_CODES[act], alpha, int(accumulate), _vp(ws),
             C.c_void_p(dxq.ptr if dxq is not None else 0), DTYPE_CODES[dxq.dtype] if dxq is not None else 0)

    def maxpool2_mask_bwd_q(self, mask_ptr, y, dy, dx, dxq, act, alpha, dbias=None, accumulate=False):
        self._whole_planes(dxq)
        N, Cc, H, W = dxq.shape
        call("ghm_maxpool2_mask_bwd_q", self.h, C.c_void_p(int(mask_ptr)), _vp(y), _vp(dy), _vp(dx), N, Cc, H, W,
             ACT_CODES[act], alpha, _vp(dbias), int(accumulate), C.c_void_p(dxq.ptr), dxq.nstride, DTYPE_CODES[dxq.dtype])

    def maxpool2_mask_bwd_compress_q(self, mask_ptr, y, dy, cq, idx, flags, act, alpha):
        """the same gradient as half-width rows + column bits (the sparse matrix instruction's operand; ghm.h); ``cq`` is a
        QTensor of shape (N, C, H, W / 2), ``idx`` (N * C / 8 * H * W / 32 units of 16 bytes) and ``flags`` (N * H int32) raw"""
        N, Cc, H, Wh = cq.shape
        call("ghm_maxpool2_mask_bwd_compress_q", self.h, C.c_void_p(int(mask_ptr)), _vp(y), _vp(dy), N, Cc, H, 2 * Wh,
             ACT_CODES[act], alpha, C.c_void_p(cq.ptr), cq.nstride, _vp(idx), _vp(flags), DTYPE_CODES[cq.dtype])

    def wgrad_pooled_split_supported(self, d, dtype):
        return dtype in SPLITS and bool(_lib.load().ghm_conv2d_wgrad_pooled_split_supported(C.byref(d)))

    def conv2d_wgrad_pooled_split(self, d, xq, dyq, cq, idx, flags, dwp, ws, dtype, accumulate=False):
        call("ghm_conv2d_wgrad_pooled_split", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, xq.pstride, C.c_void_p(dyq.ptr),
             dyq.nstride, dyq.pstride, C.c_void_p(cq.ptr), cq.nstride, cq.pstride, _vp(idx), _vp(flags), _vp(dwp), _vp(ws),
             int(accumulate), SPLITS[dtype])

    def lp_wgrad_q_supported(self, d, dtype):
        if dtype in SPLITS:
            return self.split_supported(d, 2)
        return bool(_lib.load().ghm_lp_wgrad_q_supported(C.byref(d), DTYPE_CODES[dtype]))

    def conv2d_wgrad_lp_q(self, d, xq, dyq, dwp, ws, dtype, accumulate=False):
        if dtype in SPLITS:
            return call("ghm_conv2d_wgrad_split", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, xq.pstride,
                        C.c_void_p(dyq.ptr), dyq.nstride, dyq.pstride, _vp(dwp), _vp(ws), int(accumulate), SPLITS[dtype])
        call("ghm_conv2d_wgrad_lp_q", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, C.c_void_p(dyq.ptr), dyq.nstride,
             _vp(dwp), _vp(ws), int(accumulate), DTYPE_CODES[dtype])

    def conv2d_fwd_pool_lp_q(self, d, xq, wq, bias, pooled, pooledq, mask_ptr, act, alpha, dtype):
        if dtype in SPLITS:
            assert pooledq is None or pooledq.pstride == pooledq.N * pooledq.nstride
            return call("ghm_conv2d_fwd_pool_split", self.h, C.byref(d), None, C.c_void_p(xq.ptr), xq.nstride, xq.pstride,
                        _vp(wq), _vp(bias), _vp(pooled), C.c_void_p(pooledq.ptr) if pooledq is not None else None,
                        pooledq.nstride if pooledq is not None else 0, C.c_void_p(int(mask_ptr)), ACT_CODES[act], alpha,
                        SPLITS[dtype])
        call("ghm_conv2d_fwd_pool_lp_q", self.h, C.byref(d), C.c_void_p(xq.ptr), xq.nstride, _vp(wq), _vp(bias), _vp(pooled),
             C.c_void_p(pooledq.ptr if pooledq is not None else 0), pooledq.nstride if pooledq is not None else 0,
             C.c_void_p(int(mask_ptr)), ACT_CODES[act], alpha, DTYPE_CODES[dtype])

    # ---- conv + activation + 2x2 max-pool fused (architectures/dcgan.py:42-47) ----
    def conv_pool_supported(self, d, act, dtype='f32'):
        """0: not served; 1: served with the fp32 packed weights; 2: served with the low-precision pack"""
        if dtype in SPLITS:
            if d.C > 4 and _lib.load().ghm_split_pool_supported(C.byref(d), ACT_CODES[act]):
                return 2
            dtype = 'f32'
        return int(_lib.load().ghm_conv2d_pool_supported(C.byref(d), ACT_CODES[act], DTYPE_CODES[dtype]))

    def conv2d_fwd_pool(self, d, x, w, bias, pooled, mask_ptr, act, alpha, dtype='f32'):
        assert pooled.contiguous
        if dtype in SPLITS:
            return call("ghm_conv2d_fwd_pool_split", self.h, C.byref(d), _vp(x), None, 0, 0, _vp(w), _vp(bias), _vp(pooled),
                        None, 0, C.c_void_p(int(mask_ptr)), ACT_CODES[act], alpha, SPLITS[dtype])
        call("ghm_conv2d_fwd_pool", self.h, C.byref(d), _vp(x), _vp(w), _vp(bias), _vp(pooled), C.c_void_p(int(mask_ptr)),
             ACT_CODES[act], alpha, DTYPE_CODES[dtype])

    def maxpool2_mask_bwd(self, mask_ptr, y, dy, dx, act, alpha, dbias=None, accumulate=False):
        """y, dy: pooled [N,C,H/2,W/2]; dx: full-resolution [N,C,H,W]; dbias: also (+)= the per-channel sum of dx"""
        assert (y is None or y.contiguous) and dy.contiguous and dx.contiguous      # y None: slope from the mask's sign bit
        if dbias is None:
            call("ghm_maxpool2_mask_bwd", self.h, C.c_void_p(int(mask_ptr)), _vp(y), _vp(dy), _vp(dx), dx.N, dx.Cc, dx.H,
                 dx.W, ACT_CODES[act], alpha)
        else:
            call("ghm_maxpool2_mask_bwd_bias", self.h, C.c_void_p(int(mask_ptr)), _vp(y), _vp(dy), _vp(dx), dx.N, dx.Cc,
                 dx.H, dx.W, ACT_CODES[act], alpha, _vp(dbias), int(accumulate))

    def thin_fwd_q_supported(self, d, act, pooled, dtype):
        return bool(_lib.load().ghm_thin_fwd_q_supported(C.byref(d), ACT_CODES[act], int(pooled), DTYPE_CODES[dtype]))

    def thin_pool_lp_served(self, d, act, alpha, dtype):
        """the pooled first-layer forward runs on the bf16 / fp16 matrix cores (rounded operands) for this layer"""
        return bool(_lib.load().ghm_thin_pool_lp_served(C.byref(d), ACT_CODES[act], alpha, DTYPE_CODES[dtype]))

    def conv2d_fwd_thin_q(self, d, x, w, bias, y, yq, act='linear', alpha=0.0):
        """first-layer forward (<= 4 input channels) writing its fp32 result and the q copy in one pass"""
        call("ghm_conv2d_fwd_thin_q", self.h, C.byref(d), _vp(x), _vp(w), _vp(bias), _vp(y), ACT_CODES[act], alpha,
             C.c_void_p(yq.ptr), yq.nstride, DTYPE_CODES[yq.dtype])

    def conv2d_fwd_pool_thin_q(self, d, x, w, bias, pooled, mask_ptr, yq, act, alpha):
        assert pooled is None or pooled.contiguous      # None: only the q copy and the mask are written
        call("ghm_conv2d_fwd_pool_thin_q", self.h, C.byref(d), _vp(x), _vp(w), _vp(bias), _vp(pooled),
             C.c_void_p(int(mask_ptr)), ACT_CODES[act], alpha, C.c_void_p(yq.ptr), yq.nstride, DTYPE_CODES[yq.dtype])

    def pool_bwd_sparse_supported(self, d, act):
        """bit 0: weight + bias gradient, bit 1: data gradient of a fused conv + act + pool layer from the pooled operands"""
        return int(_lib.load().ghm_conv2d_pool_bwd_sparse_supported(C.byref(d), ACT_CODES[act]))

    def pool_wgrad_sparse_workspace(self, d):
        n = C.c_size_t()
        call("ghm_conv2d_pool_wgrad_sparse_workspace", C.byref(d), C.byref(n))
        return n.value

    def conv2d_pool_wgrad_sparse(self, d, x, mask_ptr, yp, gp, dwp, dbias, ws, act, alpha, accumulate=False):
        assert (yp is None or yp.contiguous) and gp.contiguous       # yp None: slope from the mask's sign bit
        call("ghm_conv2d_pool_wgrad_sparse", self.h, C.byref(d), _vp(x), C.c_void_p(int(mask_ptr)), _vp(yp), _vp(gp), _vp(dwp),
             _vp(dbias), ACT_CODES[act], alpha, int(accumulate), _vp(ws))

    def conv2d_pool_dgrad_sparse(self, d, mask_ptr, yp, gp, wp, dx, act, alpha, accumulate=False):
        assert (yp is None or yp.contiguous) and gp.contiguous
        call("ghm_conv2d_pool_dgrad_sparse", self.h, C.byref(d), C.c_void_p(int(mask_ptr)), _vp(yp), _vp(gp), _vp(wp), _vp(dx),
             ACT_CODES[act], alpha, int(accumulate))

    def channel_sum(self, x, out, accumulate=False):
        call("ghm_channel_sum", self.h, _vp(x), x.N, x.Cc, x.HW, x.nstride, _vp(out), int(accumulate))

    def bn_workspace(self, Cc):
        return _lib.load().ghm_bn_workspace(Cc)

    def bn_stats(self, x, mean, inv, ws, run_mean=None, run_inv=None, eps=1e-4, run_alpha=0.1):
        call("ghm_bn_stats", self.h, _vp(x), x.N, x.Cc, x.HW, x.nstride, eps, _vp(mean), _vp(inv), _vp(run_mean),
             _vp(run_inv), run_alpha, _vp(ws))

    def bn_apply(self, x, y, mean, inv, gamma, beta, act='linear', alpha=0.0):
        call("ghm_bn_apply", self.h, _vp(x), x.nstride, _vp(y), y.nstride, x.N, x.Cc, x.HW, _vp(mean), _vp(inv),
             _vp(gamma), _vp(beta), ACT_CODES[act], alpha)

    def bn_forward(self, x, y, mean, inv, gamma, beta, ws, run_mean=None, run_inv=None, eps=1e-4, run_alpha=0.1,
                   act='linear', alpha=0.0):
        """batch statistics (+ running update) and normalise + activation: one launch for small tensors"""
        call("ghm_bn_forward", self.h, _vp(x), x.nstride, _vp(y), y.nstride, x.N, x.Cc, x.HW, eps, _vp(mean), _vp(inv),
             _vp(run_mean), _vp(run_inv), run_alpha, _vp(gamma), _vp(beta), ACT_CODES[act], alpha, _vp(ws))

    def instance_norm_fwd(self, x, y, mean, inv, gamma, beta, ws, eps=1e-4, act='linear', alpha=0.0, group=1):
        """InstanceNorm: BatchNorm's statistics per (instance, channel); mean / inv: [N / group, C]"""
        call("ghm_instance_norm_fwd", self.h, _vp(x), x.nstride, _vp(y), y.nstride, x.N, x.Cc, x.HW, eps, _vp(mean), _vp(inv),
             _vp(gamma), _vp(beta), ACT_CODES[act], alpha, _vp(ws), group)

    def instance_norm_bwd(self, dout, x, dx, mean, inv, gamma, beta, dgamma, dbeta, ws, act='linear', alpha=0.0,
                          accumulate=False, group=1):
        call("ghm_instance_norm_bwd", self.h, _vp(dout), dout.nstride, _vp(x), x.nstride, _vp(dx), dx.nstride, x.N, x.Cc, x.HW,
             _vp(mean), _vp(inv), _vp(gamma), _vp(beta), _vp(dgamma), _vp(dbeta), ACT_CODES[act], alpha, int(accumulate), _vp(ws),
             group)

    def bn_backward(self, dout, y, x, dx, mean, inv, gamma, dgamma, dbeta, ws, act='linear', alpha=0.0,
                    accumulate=False):
        call("ghm_bn_backward", self.h, _vp(dout), dout.nstride, _vp(y), y.nstride, _vp(x), x.nstride, _vp(dx),
             dx.nstride, x.N, x.Cc, x.HW, _vp(mean), _vp(inv), _vp(gamma), _vp(dgamma), _vp(dbeta), ACT_CODES[act],
             alpha, int(accumulate), _vp(ws))

    def act_fwd(self, x, y, act, alpha=0.0):
        call("ghm_act_fwd", self.h, _vp(x), x.nstride, _vp(y), y.nstride, x.N, x.Cc, x.HW, ACT_CODES[act], alpha)

    def act_bwd(self, dout, y, dx, act, alpha=0.0, accumulate=False):
        call("ghm_act_bwd", self.h, _vp(dout), dout.nstride, _vp(y), y.nstride, _vp(dx), dx.nstride, y.N, y.Cc, y.HW,
             ACT_CODES[act], alpha, int(accumulate))

    def maxpool2_fwd(self, x, y):
        assert x.contiguous and y.contiguous
        call("ghm_maxpool2_fwd", self.h, _vp(x), _vp(y), x.N, x.Cc, x.H, x.W)

    def maxpool2_bwd(self, x, y, dy, dx, act='linear', alpha=0.0):
        assert x.contiguous and y.contiguous and dy.contiguous and dx.contiguous
        call("ghm_maxpool2_bwd", self.h, _vp(x), _vp(y), _vp(dy), _vp(dx), x.N, x.Cc, x.H, x.W, ACT_CODES[act], alpha)

    def avgpool_fwd(self, x, y, p):
        assert x.contiguous and y.contiguous
        call("ghm_avgpool_fwd", self.h, _vp(x), _vp(y), x.N, x.Cc, x.H, x.W, p)

    def avgpool_bwd(self, dy, dx, p):
        assert dy.contiguous and dx.contiguous
        call("ghm_avgpool_bwd", self.h, _vp(dy), _vp(dx), dx.N, dx.Cc, dx.H, dx.W, p)

    def dropout(self, x, y, p, key, counter):
        """y = dropout(x); with x = dy it is the backward (same mask: same key, same counter value)"""
        call("ghm_dropout", self.h, _vp(x), x.nstride, _vp(y), y.nstride, x.N, x.Cc, x.HW, float(p), int(key) & 0xffffffff,
             _vp(counter))

    def counter_tick(self, counter):
        call("ghm_counter_tick", self.h, _vp(counter))

    def upconv_collapse_weights(self, wp5, bias, wpc, bias4, C, K):
        call("ghm_upconv_collapse_weights", self.h, _vp(wp5), _vp(bias), _vp(wpc), _vp(bias4), C, K)

    def upconv_expand_wgrad(self, dwpc, dwp5, C, K, accumulate=False):
        call("ghm_upconv_expand_wgrad", self.h, _vp(dwpc), _vp(dwp5), C, K, int(accumulate))

    def pp_to_hi(self, pp, hi):
        """pp [4N,K,H,W] (contiguous) -> hi [N,K,2H,2W]"""
        assert pp.contiguous and pp.N == 4 * hi.N
        call("ghm_pp_to_hi", self.h, _vp(pp), _vp(hi), hi.nstride, hi.N, pp.Cc, pp.H, pp.W)

    def hi_to_pp(self, hi, pp):
        assert pp.contiguous and pp.N == 4 * hi.N
        call("ghm_hi_to_pp", self.h, _vp(hi), hi.nstride, _vp(pp), hi.N, pp.Cc, pp.H, pp.W)

    def upsample_nearest2_fwd(self, x, y):
        assert y.contiguous
        call("ghm_upsample_nearest2_fwd", self.h, _vp(x), x.nstride, _vp(y), x.N, x.Cc, x.H, x.W)

    def upsample_nearest2_bwd(self, dy, dx, accumulate=False):
        assert dy.contiguous
        call("ghm_upsample_nearest2_bwd", self.h, _vp(dy), _vp(dx), dx.nstride, dx.N, dx.Cc, dx.H, dx.W,
             int(accumulate))

    def upsample_bilinear2_fwd(self, x, y):
        assert y.contiguous
        call("ghm_upsample_bilinear2_fwd", self.h, _vp(x), x.nstride, _vp(y), x.N, x.Cc, x.H, x.W)

    def upsample_bilinear2_bwd(self, dy, dx, accumulate=False):
        assert dy.contiguous
        call("ghm_upsample_bilinear2_bwd", self.h, _vp(dy), _vp(dx), dx.nstride, dx.N, dx.Cc, dx.H, dx.W,
             int(accumulate))

    def copy_view(self, x, y, accumulate=False):
        assert x.shape == y.shape
        call("ghm_copy_view", self.h, _vp(x), x.nstride, _vp(y), y.nstride, x.N, x.Cc, x.HW, int(accumulate))

    def scale_samples(self, x, num, den):
        """x[n] *= num[n] / den[n] (0 where den[n] == 0); num, den: one value per sample"""
        assert num.N == x.N and den.N == x.N and num.Cc * num.HW == 1 and den.Cc * den.HW == 1
        call("ghm_scale_samples", self.h, _vp(x), x.nstride, x.N, x.Cc, x.HW, _vp(num), num.nstride, _vp(den), den.nstride)

    def axpby(self, a, x, b, y, n):
        call("ghm_axpby", self.h, a, _vp(x), b, _vp(y), int(n))

    def image_batch(self, src_u8_ptr, N, H, W, Cc, xform_ptr, tanh_range, dst):
        call("ghm_image_batch", self.h, C.c_void_p(int(src_u8_ptr)), N, H, W, Cc, C.c_void_p(int(xform_ptr)),
             int(tanh_range), _vp(dst), dst.nstride)

    def lsgan_loss(self, d, target, loss_out, grad=None, grad_scale=1.0, accumulate_loss=False):
        assert d.contiguous
        call("ghm_lsgan_loss", self.h, _vp(d), d.size, target, _vp(loss_out), _vp(grad), grad_scale,
             int(accumulate_loss))

    def bce_loss(self, d, target, loss_out, grad=None, grad_scale=1.0, accumulate_loss=False):
        assert d.contiguous
        call("ghm_bce_loss", self.h, _vp(d), d.size, target, _vp(loss_out), _vp(grad), grad_scale,
             int(accumulate_loss))

    def recon_loss(self, a, b, loss_out, grad=None, grad_scale=1.0, l2=False, accumulate_grad=False):
        call("ghm_recon_loss", self.h, _vp(a), a.nstride, _vp(b), b.nstride, a.N, a.Cc, a.HW, int(l2), _vp(loss_out),
             _vp(grad), grad.nstride if grad is not None else 0, grad_scale, int(accumulate_grad))

    def rmsprop(self, p, g, acc, n, hyper, rho=0.9, eps=1e-6, grad_scale=1.0):
        call("ghm_rmsprop", self.h, _vp(p), _vp(g), _vp(acc), int(n), _vp(hyper), rho, eps, grad_scale)

    def adam(self, p, g, m, v, n, hyper, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        call("ghm_adam", self.h, _vp(p), _vp(g), _vp(m), _vp(v), int(n), _vp(hyper), b1, b2, eps, grad_scale)

    def adam_tick(self, hyper):
        call("ghm_adam_tick", self.h, _vp(hyper))

    def grad_check(self, g, n):
        call("ghm_grad_check", self.h, _vp(g), int(n))

    def loss_scale_update(self, growth_interval=2000, min_scale=1.0, max_scale=2.0 ** 24):
        call("ghm_loss_scale_update", self.h, int(growth_interval), float(min_scale), float(max_scale))

    def allreduce_sum(self, buf, n):
        call("ghm_allreduce_sum", self.h, _vp(buf), int(n))

    def allreduce_sum_bf16(self, buf, n, scratch_ptr):
        """the sum through a bf16 exchange buffer (``scratch_ptr``: n halfwords of device memory): half the bytes, reduced precision"""
        call("ghm_allreduce_sum_bf16", self.h, _vp(buf), int(n), C.c_void_p(int(scratch_ptr)))

    def reduce_scatter_sum(self, buf, shard):
        """buf = world x shard elements: this rank's shard receives the sum over the ranks (in place)"""
        call("ghm_reduce_scatter_sum", self.h, _vp(buf), int(shard))

    def all_gather(self, buf, shard):
        """buf = world x shard elements: every rank's shard is made whole on every rank (in place)"""
        call("ghm_all_gather", self.h, _vp(buf), int(shard))

    def allreduce_max(self, buf, n):
        call("ghm_allreduce_max", self.h, _vp(buf), int(n))

    def conv_variant(self, d, kind):
        out = C.create_string_buffer(128)
        call("ghm_conv2d_variant", C.byref(d), kind, out, 128)
        return out.value.decode()

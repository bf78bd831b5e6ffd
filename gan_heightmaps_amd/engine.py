"""Lowering of lasagne-style layer graphs to libghm.so launches, with hand-derived backward.

This replaces what Theano does for the reference at ``theano.function`` time
(/root/reference/pix2pix.py:142-147): a layer graph becomes a static *program* -- a list of kernel
launches over preallocated HBM buffers -- that is recorded once and then replayed (optionally as a
captured HIP graph).  There is no autodiff: every node type has its backward written out here.

Graph rewrites before placement (both exact):
  * act(concat(a, b)) -> concat(act(a), act(b)) with sharing of an identical act(a) that already
    exists (U-Net skips: architectures/p2p.py:202-203 applies leaky_rectify after the concat while the
    encoder applies the same nonlinearity to the same tensor),
  * a NonlinearityLayer whose producer is a conv / deconv / dense / BN node with no other consumer is
    folded into the producer's epilogue.
Placement: inputs of a ConcatLayer(axis=1) are written in place into channel slices of the concat
buffer (sample stride = total channels * H * W), so concatenation costs no copy.
"""
import os

import numpy as np

from . import layers as L
from .architectures.layers import BilinearUpsample2DLayer
from .device import SPLITS, DevTensor, QTensor, conv_desc, pack_conv_w, unpack_conv_w
from .nonlinearities import linear

ALIGN = 64      # elements; keeps every parameter 256-B aligned inside the flat buffers


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class ParamStore:
    """All parameters of one network in three flat fp32 buffers: trainable values ``w``, their
    gradients ``g`` (same layout) and non-trainable state ``s`` (BN mean / inv_std).  The flat layout
    makes the optimiser one kernel and the data-parallel exchange one all-reduce per network."""

    def __init__(self, dev, params, pad_to=1):
        """``pad_to``: the flat buffers are allocated with a multiple of this many elements (the sharded data-parallel
        update cuts them into world equal, 256-byte-aligned shards); n_train stays the number of elements in use"""
        self.dev = dev
        self.params = list(params)
        nt = ns = 0
        for p in self.params:
            n = int(np.prod(p.shape))
            if 'trainable' in p.tags:
                p.index = ('w', nt)
                nt += _align(n)
            else:
                p.index = ('s', ns)
                ns += _align(n)
        self.n_train, self.n_state = nt, ns
        self.n_pad = (max(nt, 1) + pad_to - 1) // pad_to * pad_to
        self.w = dev.zeros((1, self.n_pad, 1, 1))
        self.g = dev.zeros((1, self.n_pad, 1, 1))
        self.s = dev.zeros((1, max(ns, 1), 1, 1))
        self.opt_state = {}
        for p in self.params:
            p.store = self
            self.upload(p)

    def _view(self, base, p):
        n = int(np.prod(p.shape))
        return DevTensor(self.dev, base.ptr + 4 * p.index[1], (1, n, 1, 1), None, base)

    def value(self, p):
        return self._view(self.w if p.index[0] == 'w' else self.s, p)

    def grad(self, p):
        assert p.index[0] == 'w'
        return self._view(self.g, p)

    @staticmethod
    def _to_device_layout(p, v):
        if p.kind == 'conv_w':
            return pack_conv_w(v).ravel()
        return np.ascontiguousarray(v, np.float32).ravel()

    @staticmethod
    def _from_device_layout(p, flat):
        if p.kind == 'conv_w':
            K, C, kh, kw = p.shape
            return unpack_conv_w(flat, K, C, kh, kw).astype(np.float32)
        return flat.reshape(p.shape).copy()

    def upload(self, p):
        self.value(p).set(self._to_device_layout(p, p.value))

    def download(self, p):
        return self._from_device_layout(p, self.value(p).numpy().ravel())

    def download_grad(self, p):
        return self._from_device_layout(p, self.grad(p).numpy().ravel())

    def lp_pack(self, key, nbytes):
        """device buffer of a 16-byte-unit bf16 / fp16 weight pack (ghm_lp_pack_weights), refreshed from the fp32
        master weights once per step before its first use"""
        if not hasattr(self, '_lp'):
            self._lp = {}
        if key not in self._lp:
            self._lp[key] = self.dev.alloc(nbytes)
        return self._lp[key]

    def transposed(self, p):
        """scratch for the transposed packed weights of a stride-1 conv (refreshed each step before its first
        data-gradient use: ghm_conv2d_transpose_weights)"""
        if not hasattr(self, '_wT'):
            self._wT = {}
        if id(p) not in self._wT:
            self._wT[id(p)] = self.dev.empty((1, int(np.prod(p.shape)), 1, 1))
        return self._wT[id(p)]


# --------------------------------------------------------------------------------------------------
# IR
# --------------------------------------------------------------------------------------------------
class Node:
    def __init__(self, op, inputs, layer=None, **attrs):
        self.op = op
        self.inputs = list(inputs)
        self.layer = layer
        self.attrs = attrs
        self.act = linear               # epilogue nonlinearity of conv / deconv / dense / bn / act nodes
        self.consumers = []
        self.shape = None
        self.out = None
        self.alias = None               # (concat node, channel offset)
        self.aux = {}

    def __repr__(self):
        return "<%s %s act=%s>" % (self.op, self.shape, self.act.kind)


def _build_ir(out_layer):
    nodes, of = [], {}
    for l in L.get_all_layers(out_layer):
        if isinstance(l, L.InputLayer):
            n = Node('input', [], l)
        elif isinstance(l, L.DenseLayer):
            n = Node('dense', [of[id(l.input_layer)]], l)
            n.act = l.nonlinearity
        elif isinstance(l, L.Conv2DLayer):
            n = Node('conv', [of[id(l.input_layer)]], l)
            n.act = l.nonlinearity
        elif isinstance(l, L.TransposedConv2DLayer):
            n = Node('deconv', [of[id(l.input_layer)]], l)
            n.act = l.nonlinearity
        elif isinstance(l, L.BatchNormLayer):
            n = Node('bn', [of[id(l.input_layer)]], l)
        elif isinstance(l, L.InstanceNormLayer):
            # a 'bn' node whose statistics are per instance (layers.InstanceNormLayer): the BatchNorm lowering with its own
            # kernels entry points (ghm_instance_norm_*), none of BatchNorm's fusions, no running statistics
            n = Node('bn', [of[id(l.input_layer)]], l)
            n.instance = True
        elif isinstance(l, L.NonlinearityLayer):
            if l.nonlinearity == linear:
                of[id(l)] = of[id(l.input_layer)]
                continue
            n = Node('act', [of[id(l.input_layer)]], l)
            n.act = l.nonlinearity
        elif isinstance(l, L.ReshapeLayer):
            n = Node('reshape', [of[id(l.input_layer)]], l)
        elif isinstance(l, L.Upscale2DLayer):
            n = Node('up_nearest', [of[id(l.input_layer)]], l)
        elif isinstance(l, BilinearUpsample2DLayer):
            n = Node('up_bilinear', [of[id(l.input_layer)]], l)
        elif isinstance(l, L.Pool2DLayer):
            if l.mode == 'max':
                if l.pool_size != (2, 2):
                    raise NotImplementedError("max pooling other than 2x2")
                n = Node('maxpool', [of[id(l.input_layer)]], l)
            else:
                n = Node('avgpool', [of[id(l.input_layer)]], l, p=l.pool_size[0])
        elif isinstance(l, L.ConcatLayer):
            n = Node('concat', [of[id(i)] for i in l.input_layers], l)
        elif isinstance(l, L.DropoutLayer):
            if l.p <= 0:
                of[id(l)] = of[id(l.input_layer)]
                continue
            n = Node('dropout', [of[id(l.input_layer)]], l, p=float(l.p))
        else:
            raise NotImplementedError("no lowering for %r" % (l,))
        of[id(l)] = n
        nodes.append(n)
    return nodes, of


def _toposort(out_node):
    order, seen = [], set()
    stack = [(out_node, False)]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if id(n) in seen:
            continue
        seen.add(id(n))
        stack.append((n, True))
        for i in reversed(n.inputs):
            if id(i) not in seen:
                stack.append((i, False))
    for n in order:
        n.consumers = []
    for n in order:
        for i in n.inputs:
            i.consumers.append(n)
    return order


def _replace(order, old, new):
    for n in order:
        n.inputs = [new if i is old else i for i in n.inputs]


def _blconv_wanted(conv_layer, up_layer, dtype):
    """BilinearUpsample2DLayer(2) -> 3x3 'same' stride-1 conv evaluated on the coarse grid (csrc/conv_bilinear.hip)?  Served
    geometries: channels and filters multiples of 32, coarse maps of at least 32 x 32 (GHM_BLCONV_MIN; on 16-wide coarse maps
    the narrow-map kernels run the 4K-filter form slower than the literal 32-wide one).  In the split arithmetic modes only (the fp32-MFMA mode stays the literal
    bit-reference of the layer); GHM_NO_BLCONV=1 keeps the literal form everywhere, GHM_BLCONV=all extends it to every mode."""
    if os.environ.get("GHM_NO_BLCONV") is not None:
        return False
    if dtype not in SPLITS and os.environ.get("GHM_BLCONV") != 'all':
        return False
    l = conv_layer
    if l.filter_size != (3, 3) or l.stride != (1, 1) or l.pad != (1, 1) or getattr(up_layer, 'factor', 2) != 2:
        return False
    cs = up_layer.input_layer.output_shape
    nmin = int(os.environ.get("GHM_BLCONV_MIN", 32))
    return (len(cs) == 4 and cs[1] % 32 == 0 and l.num_filters % 32 == 0 and cs[2] >= nmin and cs[3] >= nmin
            and cs[2] % 8 == 0 and cs[3] % 8 == 0)


def _rewrite(out_node, dtype='f32'):
    # R1: act(concat(..)) -> concat(act(..)..) with CSE
    changed = True
    while changed:
        changed = False
        order = _toposort(out_node)
        for n in order:
            if n.op == 'act' and n.inputs[0].op == 'concat' and len(n.inputs[0].consumers) == 1:
                cat = n.inputs[0]
                new_inputs = []
                for a in cat.inputs:
                    shared = [c for c in a.consumers if c.op == 'act' and c.act == n.act and c is not n]
                    if shared:
                        new_inputs.append(shared[0])
                    else:
                        m = Node('act', [a], n.layer)
                        m.act = n.act
                        new_inputs.append(m)
                new_cat = Node('concat', new_inputs, cat.layer)
                if n is out_node:
                    out_node = new_cat
                _replace(order, n, new_cat)
                changed = True
                break
    # R2: fold a sole-consumer act into its producer's epilogue
    changed = True
    while changed:
        changed = False
        order = _toposort(out_node)
        for n in order:
            if n.op == 'act':
                m = n.inputs[0]
                if m.op in ('conv', 'deconv', 'dense', 'bn') and m.act == linear and len(m.consumers) == 1:
                    m.act = n.act
                    if n is out_node:
                        out_node = m
                    _replace(order, n, m)
                    changed = True
                    break
    # R3: Upscale2DLayer(2) -> 5x5 'same' conv  ==>  3x3 conv with 4K filters on the low-res input whose output is
    # the parity-planar [4B, K, H, W] tensor (BatchNorm / activation run on it unchanged), then one interleave
    # pass to [B, K, 2H, 2W].  9 instead of 25 MACs per output, no 4x up-sampled tensor (csrc/elementwise.hip).
    if os.environ.get("GHM_NO_UPCONV") is None:
        changed = True
        while changed:
            changed = False
            order = _toposort(out_node)
            for n in order:
                if n.op != 'conv' or n.inputs[0].op not in ('up_nearest', 'up_bilinear') or len(n.inputs[0].consumers) != 1:
                    continue
                l = n.layer
                if n.inputs[0].op == 'up_nearest':
                    if l.filter_size != (5, 5) or l.stride != (1, 1) or l.pad != (2, 2):
                        continue
                    mode = 0
                else:
                    # R3b: BilinearUpsample2DLayer(2) -> 3x3 'same' conv (p2p.py:204-267) the same way: a packed 3x3 conv with
                    # 4K filters on the coarse input (25 of its 36 collapsed taps non-zero) + the frame Theano's border
                    # handling adds (csrc/conv_bilinear.hip); mode 1 of the same node
                    if not _blconv_wanted(l, n.inputs[0].layer, dtype):
                        continue
                    mode = 1
                uc = Node('upconv', [n.inputs[0].inputs[0]], l, mode=mode)
                uc.act = n.act
                tail_old, tail_new = n, uc
                if len(n.consumers) == 1 and n.consumers[0].op == 'bn':
                    tail_old = tail_new = n.consumers[0]
                    tail_new.inputs = [uc]
                sh = Node('pp_to_hi', [tail_new], None)
                if tail_old is out_node:
                    out_node = sh
                for m in order:
                    if m is not tail_new:
                        m.inputs = [sh if i is tail_old else i for i in m.inputs]
                changed = True
                break
    return out_node


class NetPlan:
    """One network lowered for a fixed batch size: buffers + emitters of forward/backward programs."""

    def __init__(self, dev, ops, out_layer, batch, store, inputs=None, out_tensor=None, name="net", side=None,
                 bn_groups=1, rng_seed=0, rng_counter=None, dtype='f32'):
        """``side=(Device, Ops)``: a second stream of the same GPU for the weight / bias gradients, which only
        feed the optimiser: they fork off the main stream where their output gradient is ready and run beside
        the data-gradient chain (the caller joins the two streams before the update)."""
        self.dev, self.ops, self.batch, self.store, self.name = dev, ops, batch, store, name
        self.side = side
        # arithmetic of the convolution products: 'f32' (the reference's floatX) or 'bf16' / 'f16' on the matrix cores
        # for every geometry the low-precision kernels serve; tensors in HBM are fp32 either way (include/ghm.h)
        assert dtype in ('f32', 'bf16', 'f16') or dtype in SPLITS, dtype
        self.dtype = dtype
        # bn_groups=2: the batch is [real | fake] (two get_output calls of the reference, pix2pix.py:94-95,98-101):
        # every BatchNormLayer normalises each half with its own statistics
        self.bn_groups = bn_groups
        nodes, of = _build_ir(out_layer)
        self.out_node = _rewrite(of[id(out_layer)], dtype)
        self.order = _toposort(self.out_node)
        self.node_of_layer = of
        self.input_nodes = [n for n in self.order if n.op == 'input']
        self._shapes()
        self._fuse_convpool()
        self._place(inputs or {}, out_tensor)
        # q tensors (include/ghm.h): in the reduced-precision modes every tensor a low-precision product reads as an
        # operand also exists as a bf16 / fp16 copy in channel-block-of-8 layout, written by its producer
        self.use_q = self.dtype != 'f32' and os.environ.get("GHM_NO_Q") is None and hasattr(ops, 'q_pack')
        # every producer with a q epilogue writes its q copy itself in every q mode (and fp32 tensors nobody reads are dropped)
        self.q_epi = self.use_q
        for n in self.order:
            n.outq = None
        if self.use_q:
            self._place_q()
        self._scratch = {}
        self.bn_ws = None
        cmax = max([n.shape[1] for n in self.order if n.op == 'bn'] + [0])
        if cmax:
            self.bn_ws = dev.alloc(ops.bn_workspace(cmax))
            self._bn_scratch = dev.empty((1, 2 * cmax, 1, 1))
        self.wgrad_ws = None
        self._wgrad_ws_bytes = 0
        # DropoutLayer: masks are hashes of (element, per-layer key, step counter); the counter lives in HBM and is
        # advanced once per non-deterministic forward pass (so a captured graph draws fresh masks on replay)
        self.dropout_nodes = [n for n in self.order if n.op == 'dropout']
        self.rng_counter = rng_counter
        if self.dropout_nodes and self.rng_counter is None:
            self.rng_counter = dev.zeros((1, 1, 1, 1))
        for i, n in enumerate(self.dropout_nodes):
            n.aux['key'] = (rng_seed * 0x9E3779B1 + (i + 1) * 0x85EBCA77) & 0xffffffff
        self._lp_wq, self._lp_table = {}, None
        if self.dtype != 'f32':
            self._plan_lp_packs()

    def _plan_lp_packs(self):
        """bf16 / fp16 weight packs of the whole net (forward and transposed, plain and collapsed up-sample convs):
        buffers + ONE device table, refreshed by one ghm_lp_pack_batched launch at the start of every forward program
        (after the collapse of the generator's 5x5 weights)"""
        items = []
        for n in self.order:
            if n.op in ('conv', 'convpool'):
                d = self._desc(n, n.inputs[0].out, self._full(n))
                src, key = self.store.value(n.layer.W), ('w', id(n.layer.W))
            elif n.op == 'upconv':
                d = self._upconv_desc(n, n.inputs[0].out)
                src, key = n.aux['wpc'], ('c', id(n.layer.W))
            else:
                continue
            T = d.kh * d.kw
            for transposed, kind in ((False, 0), (True, 1)):
                if self._lp(d, kind) and (key, transposed) not in self._lp_wq:
                    wq = self.store.lp_pack((key, transposed, self.dtype), self.ops.lp_weight_bytes(d, transposed, self.dtype))
                    self._lp_wq[(key, transposed)] = wq
                    items.append((src, wq, d.K if transposed else d.C, T, d.C if transposed else d.K, transposed))
        if items:
            self._lp_table = self.ops.lp_pack_table(items)

    # ---- shapes and placement ----------------------------------------------------------------
    def _shapes(self):
        B = self.batch
        for n in self.order:
            if n.op == 'input':
                s = n.layer.output_shape
                n.shape = (B,) + tuple(s[1:]) if len(s) == 4 else (B, s[1], 1, 1)
            elif n.op == 'reshape':
                tgt = n.layer.shape
                per = int(np.prod(n.inputs[0].shape[1:]))
                rest = int(np.prod(tgt[1:]))
                if rest != per:
                    raise NotImplementedError("ReshapeLayer that changes the batch dimension")
                n.shape = (B,) + tuple(tgt[1:]) if len(tgt) == 4 else (B, tgt[1], 1, 1)
            elif n.op == 'concat':
                s0 = n.inputs[0].shape
                n.shape = (B, sum(i.shape[1] for i in n.inputs), s0[2], s0[3])
            elif n.op == 'upconv':
                s0 = n.inputs[0].shape
                n.shape = (4 * s0[0], n.layer.num_filters, s0[2], s0[3])
            elif n.op == 'pp_to_hi':
                s0 = n.inputs[0].shape
                n.shape = (s0[0] // 4, s0[1], 2 * s0[2], 2 * s0[3])
            elif n.op == 'bn':
                n.shape = tuple(n.inputs[0].shape)
            else:
                ls = n.layer.get_output_shape_for((B,) + tuple(n.inputs[0].shape[1:])) \
                    if n.op not in ('dense',) else (B, n.layer.num_units)
                n.shape = tuple(ls) if len(ls) == 4 else (B, ls[1], 1, 1)

    def _fuse_convpool(self):
        """Conv2DLayer -> LeakyRectify -> MaxPool2DLayer(2) (architectures/dcgan.py:42-47) -> ONE node whose kernel
        pools in its epilogue and keeps a 4-bit arg-max mask (ghm_conv2d_fwd_pool): the full-resolution activation is
        never written, the backward rebuilds the conv's output gradient from mask + pooled value + pooled gradient."""
        if os.environ.get("GHM_NO_POOL_FUSE") is not None:
            return
        for n in list(self.order):
            if n.op != 'maxpool' or n is self.out_node:
                continue
            c = n.inputs[0]
            if c.op != 'conv' or len(c.consumers) != 1 or c.act.kind not in ('linear', 'relu', 'lrelu'):
                continue
            l, xs = c.layer, c.inputs[0].shape
            d = conv_desc(xs[0], xs[1], xs[2], xs[3], l.num_filters, l.filter_size[0], l.filter_size[1], l.stride[0], l.pad[0])
            if not self.ops.conv_pool_supported(d, c.act.kind, self.dtype):
                continue
            c.op = 'convpool'
            c.aux['full_shape'] = tuple(c.shape)
            c.shape = tuple(n.shape)
            for m in self.order:
                m.inputs = [c if i is n else i for i in m.inputs]
            self.order.remove(n)
        for m in self.order:
            m.consumers = []
        for m in self.order:
            for i in m.inputs:
                i.consumers.append(m)

    def _place(self, inputs, out_tensor):
        for n in self.order:
            if n.op == 'concat':
                c0 = 0
                for i in n.inputs:
                    if i.alias is None and i.op in ('conv', 'deconv', 'dense', 'bn', 'act', 'input', 'pp_to_hi') and i.out is None:
                        i.alias = (n, c0)
                    c0 += i.shape[1]
        if out_tensor is not None:
            assert out_tensor.shape == self.out_node.shape, (out_tensor.shape, self.out_node.shape)
            assert self.out_node.alias is None
            self.out_node.out = out_tensor
        for n in self.input_nodes:
            if n.layer in inputs:
                assert n.alias is None, "external tensor for an input that lives inside a concat buffer"
                t = inputs[n.layer]
                assert t.shape == n.shape, (t.shape, n.shape)
                n.out = t

        def get_out(n):
            if n.out is not None:
                return n.out
            if n.alias is not None:
                cat, c0 = n.alias
                n.out = get_out(cat).channels(c0, c0 + n.shape[1])
            elif n.op == 'reshape':
                n.out = get_out(n.inputs[0]).reshape(n.shape)
            else:
                n.out = self.dev.empty(n.shape)
            return n.out

        for n in self.order:
            get_out(n)
            if n.op == 'bn' and getattr(n, 'instance', False):
                # one instance = one sample; the four parity planes of one image behind a collapsed up-sample convolution
                grp = 4 if n.inputs[0].op == 'upconv' else 1
                n.aux['group'] = grp
                n.aux['mean'] = self.dev.empty((1, n.shape[0] // grp * n.shape[1], 1, 1))
                n.aux['inv'] = self.dev.empty((1, n.shape[0] // grp * n.shape[1], 1, 1))
            elif n.op == 'bn':
                C = n.shape[1]
                n.aux['mean'] = self.dev.empty((1, C, 1, 1))
                n.aux['inv'] = self.dev.empty((1, C, 1, 1))
                if self.bn_groups == 2:
                    n.aux['mean_g'] = [n.aux['mean'], self.dev.empty((1, C, 1, 1))]
                    n.aux['inv_g'] = [n.aux['inv'], self.dev.empty((1, C, 1, 1))]
            if n.op == 'convpool':
                n.aux['mask'] = self.dev.alloc(int(np.prod(n.shape)))          # one byte per pooled pixel
            if n.op == 'upconv':
                C, K = n.inputs[0].shape[1], n.shape[1]
                for name in ('wpc', 'wpcT', 'dwpc'):
                    n.aux[name] = self.dev.empty((1, C * 9 * 4 * K, 1, 1))
                n.aux['b4'] = self.dev.empty((1, 4 * K, 1, 1))
                if n.attrs.get('mode') == 1:
                    xs = n.inputs[0].shape
                    if not self.ops.blconv_supported(xs[0], C, K, xs[2], xs[3]):
                        raise NotImplementedError("bilinear up-sample convolution %r: geometry not served by the library" % (n,))
                    nfl, ndyl = self.ops.blconv_frame_sizes(xs[0], C, K, xs[2], xs[3])
                    n.aux['fl'], n.aux['dyl'] = self.dev.empty((1, nfl, 1, 1)), self.dev.empty((1, ndyl, 1, 1))

    def _place_q(self):
        """allocate the q copy of every node output that a low-precision forward product reads (the conv's input): the
        same placement as the fp32 tensors -- inputs of a ConcatLayer are channel slices of the concat's q buffer"""
        need = []
        for n in self.order:
            if n.op in ('conv', 'convpool'):
                d = self._desc(n, n.inputs[0].out, self._full(n))
            elif n.op == 'upconv':
                d = self._upconv_desc(n, n.inputs[0].out)
            else:
                continue
            if self._lp(d, 0) and n.inputs[0].shape[1] % 8 == 0:
                need.append(n.inputs[0])

        def get_outq(n):
            if n.outq is not None:
                return n.outq
            if n.alias is not None:
                cat, c0 = n.alias
                if c0 % 8 == 0 and n.shape[1] % 8 == 0 and cat.shape[1] % 8 == 0:
                    n.outq = get_outq(cat).channels(c0, c0 + n.shape[1])
                    return n.outq
            n.outq = QTensor.empty(self.dev, n.shape, self.dtype)
            return n.outq

        for n in need:
            get_outq(n)
            n.aux['q_whole'] = True
            if n.op == 'concat':            # read as a whole: every input that lives inside it writes its q slice
                for i in n.inputs:
                    if i.alias is not None and i.alias[0] is n:
                        get_outq(i)

    def _wq(self, d):
        """is the weight gradient of this convolution the q-operand kernel (lp_wgrad_q: 3x3 stride 1 / 2, 5x5 stride 1)?
        Measured against the register-staged lp_wgrad kernel at one round of resident blocks (bf16, TFLOP/s): 3x3 stride 1
        594-830 vs 267-474, 5x5 1024-1115 vs 517-584, 3x3 stride 2 317-386 vs 277-319 (N4 C64 256^2: 193 vs 220)."""
        return bool(self.use_q and self.dtype != 'f32' and self.ops.lp_wgrad_q_supported(d, self.dtype))

    def _fp32_needed(self, n):
        """does anything read the fp32 output of node n (which also has a q copy)?  Not when every consumer is a
        low-precision convolution whose forward AND weight gradient read the q copy."""
        if n is self.out_node or n.outq is None or n.alias is not None or not n.consumers or not self.q_epi:
            return True
        for c in n.consumers:
            if c.op in ('conv', 'convpool') and c.inputs[0] is n:
                d = self._desc(c, n.out, self._full(c))
            elif c.op == 'upconv' and c.inputs[0] is n:
                if c.attrs.get('mode') == 1:
                    return True             # the frame kernels (conv_bilinear.hip) read the coarse fp32 border rows / columns
                d = self._upconv_desc(c, n.out)
            else:
                return True
            if not (self._lp(d, 0) and self._wq(d)):
                return True
        return False

    def _act_fp32_dropped(self, n):
        """conv -> relu / leaky relu -> conv (the PatchGAN's chain, p2p.py:278-292) in the split modes: is the fp32 tensor of this
        conv's activated output never read?  Its one consumer is a convolution whose forward and weight gradient read the q
        copy and whose data gradient differentiates this layer's nonlinearity in its own epilogue from the SIGN of that q copy
        (emit_backward, form 3 with ``ysrc`` = the q tensor).  GHM_KEEP_ACT_FP32=1 / GHM_DACT_FP32=1 keep the fp32 tensor."""
        if (n.op != 'conv' or self.dtype not in SPLITS or not self.q_epi or not self.use_q or n.act.kind not in ('relu', 'lrelu')
                or os.environ.get("GHM_KEEP_ACT_FP32") or os.environ.get("GHM_DACT_FP32")):
            return False
        if n is self.out_node or n.outq is None or n.alias is not None or len(n.consumers) != 1 or n.shape[1] % 8:
            return False
        c = n.consumers[0]
        if c.op not in ('conv', 'convpool') or c.inputs[0] is not n or c.shape[1] % 8 or self._fp32_needed(n):
            return False
        # the consumer's data gradient must take the fused form that reads the q copy (the same tests as emit_backward)
        dc = self._desc(c, n.out, self._full(c) if c.op == 'convpool' else c.out)
        return self.ops.dgrad_dact_supported(dc, self.dtype) == 3

    def _pool_y_dropped(self, n):
        """fused conv + activation + max-pool node whose pooled fp32 tensor is never written: every consumer reads the q
        copy, and the backward pass takes the activation slope from the sign bit the forward kernel leaves in the mask"""
        if (n.op != 'convpool' or not self.q_epi or os.environ.get("GHM_KEEP_POOL_Y") is not None
                or os.environ.get("GHM_POOL_READ_Y") is not None):      # (the A/B switch of the backward reads it back)
            return False
        if n.act.kind not in ('linear', 'relu', 'lrelu') or self._fp32_needed(n):
            return False
        d = self._desc(n, n.inputs[0].out, self._full(n))
        form = self.ops.conv_pool_supported(d, n.act.kind, self.dtype)
        if form == 2:
            return n.inputs[0].outq is not None                      # the q-operand pooled kernel writes n.outq itself
        return form == 1 and bool(self.ops.thin_fwd_q_supported(d, n.act.kind, True, self.dtype))

    def input_tensor(self, layer):
        return self.node_of_layer[id(layer)].out

    @property
    def out(self):
        return self.out_node.out

    # ---- helpers ---------------------------------------------------------------------------------
    def _desc(self, n, x_t, y_t):
        """conv descriptor of node n for tensors (conv-input-side x_t, conv-output-side y_t)."""
        l = n.layer
        if n.op == 'dense':
            return conv_desc(x_t.N, x_t.Cc * x_t.HW, 1, 1, l.num_units, 1, 1, 1, 0, x_t.nstride, y_t.nstride)
        k = l.filter_size
        if n.op in ('conv', 'convpool'):
            return conv_desc(x_t.N, x_t.Cc, x_t.H, x_t.W, l.num_filters, k[0], k[1], l.stride[0], l.pad[0],
                             x_t.nstride, y_t.nstride)
        # deconv: descriptor of the conv it is the adjoint of (conv input = deconv OUTPUT = x_t here)
        return conv_desc(x_t.N, x_t.Cc, x_t.H, x_t.W, y_t.Cc, k[0], k[1], l.stride[0], 0, x_t.nstride, y_t.nstride)

    def _full(self, n, nb=None):
        """geometry stand-in for the conv output of node n (the fused conv + pool node has no such tensor in the
        forward pass: only its shape and dense sample stride matter)"""
        if n.op != 'convpool':
            return n.out
        fs = n.aux['full_shape']
        return DevTensor(self.dev, 0, ((nb if nb is not None else fs[0]),) + tuple(fs[1:]))

    def _upconv_desc(self, n, x_t):
        """3x3 'same' conv with 4K filters on the low-res input x_t -> parity-planar output seen as [N, 4K, H, W]"""
        K = n.shape[1]
        return conv_desc(x_t.N, x_t.Cc, x_t.H, x_t.W, 4 * K, 3, 3, 1, 1, x_t.nstride, 4 * K * x_t.H * x_t.W)

    def _bl_skip(self, n, d, kind):
        """does the collapsed bilinear convolution of node n run with its structurally zero taps skipped (split modes)?"""
        return (n.attrs.get('mode') == 1 and self.dtype in SPLITS and hasattr(self.ops, 'blconv_split_supported')
                and bool(self.ops.blconv_split_supported(d, kind, self.dtype)))

    def _use_dgrad_t(self, d, W):
        """data gradient through the transposed weight copy (forward-form kernels)?  Not for <= 4 filters unless the
        thin fan-out kernel serves them: their reduction is element-wise work (smallk_dgrad_kernel)"""
        if W.shape[1] <= 4 or not self.ops.dgrad_t_supported(d):
            return False
        return d.K > 4 or self.ops.conv_variant(d, 3).startswith("fanout_kernel")

    def _lp(self, d, kind):
        return self.dtype != 'f32' and self.ops.lp_supported(d, kind, self.dtype)

    def _lp_pack_entry(self, prog, d, w_src, key, transposed, done):
        """-> device pointer of the bf16 / fp16 pack of ``w_src``; emits the refresh once per program set ``done``"""
        if (key, transposed) in self._lp_wq:
            return self._lp_wq[(key, transposed)]         # refreshed by the batched pack at the start of the forward
        wq = self.store.lp_pack((key, transposed, self.dtype), self.ops.lp_weight_bytes(d, transposed, self.dtype))
        if done is None or (key, transposed) not in done:
            if done is not None:
                done.add((key, transposed))
            prog.append(("lp_pack_t" if transposed else "lp_pack", lambda d=d, w=w_src, wq=wq, t=transposed:
                         self.ops.lp_pack_weights(d, w, wq, self.dtype, t)))
        return wq

    def _q_bytes(self):
        """bytes per element of a q tensor in this engine's arithmetic (2 per piece)"""
        return 2.0 * SPLITS.get(self.dtype, 1)

    def _conv_bn_fusable(self, n, d, xq, deterministic):
        """-> the BatchNorm node behind convolution n when the pair runs as one product (ghm_conv2d_bn_fwd_lp_q), else None"""
        if (n.op != 'conv' or deterministic or self.bn_groups != 1 or n.act != linear or len(n.consumers) != 1
                or os.environ.get("GHM_NO_CONV_BN_FUSE") is not None or not hasattr(self.ops, 'conv_bn_fused_supported')):
            return None
        bnn = n.consumers[0]
        if bnn.op != 'bn' or getattr(bnn, 'instance', False) or self._bn_hi(bnn) or n.out.nstride != d.y_nstride:
            return None
        if self._lp(d, 0):
            return bnn if (xq is not None and self.ops.conv_bn_fused_supported(d, self.dtype)) else None
        # fp32 product (ghm_conv2d_bn_fwd: the generic gather kernel in split-K form + the same finishing kernel).  Built,
        # parity-tested, and NOT the default: in the fp32 step it measured 169.4 img/s against 171.5 for the split-K
        # epilogue + one-launch BatchNorm it replaces (two runs each, same box) -- GHM_CONV_BN_F32=1 turns it on
        if os.environ.get("GHM_CONV_BN_F32") is None:
            return None
        return bnn if (bnn.outq is None and self.ops.conv_bn_fused_supported(d, 'f32')) else None

    def _bn_hi(self, n):
        """is n the BatchNorm of a collapsed up-sample convolution whose only reader is the parity interleave?  Then the
        two run as one pass in both directions (csrc/elementwise_q.hip: bn_apply_hi / bn_backward_hi)."""
        return (n.op == 'bn' and self.bn_groups == 1 and not getattr(n, 'instance', False) and os.environ.get("GHM_NO_BN_HI") is None
                and len(n.consumers) == 1 and n.consumers[0].op == 'pp_to_hi' and n.inputs[0].op == 'upconv'
                and n.shape[1] % 8 == 0 and n.consumers[0].out.nstride % 2 == 0)

    def _need_wgrad_ws(self, d):
        b = self.ops.wgrad_workspace(d)
        if self._lp(d, 2) or self._wq(d):
            b = max(b, self.ops.wgrad_lp_workspace(d))
        self._grow_wgrad_ws(b)

    def _grow_wgrad_ws(self, b):
        """one weight-gradient workspace per plan, sized while its programs are emitted (i.e. before any of them is
        recorded: a recorded step holds the pointer by value)"""
        if b > self._wgrad_ws_bytes:
            if self.wgrad_ws is not None:
                self.dev.free(self.wgrad_ws)
            self.wgrad_ws = self.dev.alloc(b)
            self._wgrad_ws_bytes = b

    # ---- forward ---------------------------------------------------------------------------------
    def emit_forward(self, prog, deterministic=False, update_running=True, lp_done=None):
        """``lp_done``: set of low-precision weight packs already refreshed in the program being built (a net that is
        evaluated twice per step packs once)"""
        ops, st = self.ops, self.store
        if self.dropout_nodes and not deterministic:
            prog.append(("rng_tick", lambda c=self.rng_counter: ops.counter_tick(c)))
        # the collapsed 3x3 weights of every up-sample convolution of the net: ONE launch at the head of the forward pass
        # (per layer it was two small launches in front of each of the generator's first, latency-bound stages)
        ups = [n for n in self.order if n.op == 'upconv']
        if ups:
            if getattr(self, '_collapse_tab', None) is None:
                self._collapse_tab = ops.collapse_table(
                    [(st.value(n.layer.W), st.value(n.layer.b), n.aux['wpc'], n.aux['b4'], n.inputs[0].shape[1], n.shape[1],
                      n.attrs.get('mode', 0)) for n in ups])
            prog.append(("collapse_w", lambda t=self._collapse_tab: ops.upconv_collapse_batched(t)))
        if self._lp_table is not None:
            prog.append(("lp_pack", lambda t=self._lp_table: ops.lp_pack_batched(t, self.dtype)))
        qpack = lambda t, q: prog.append(("q_pack", lambda t=t, q=q: ops.q_pack(t, q), pack_meta(t)))
        fused_hi = set()            # pp_to_hi nodes whose output the BatchNorm in front of them writes
        fused_bn = set()            # BatchNorm nodes that ran inside the finishing kernel of the convolution in front of them
        for n in self.order:
            if id(n) in fused_bn:
                continue
            y = n.out
            if n.op in ('input', 'reshape', 'concat'):
                if n.op == 'concat':
                    c0 = 0
                    for i in n.inputs:
                        if i.alias is None or i.alias[0] is not n:
                            dst = y.channels(c0, c0 + i.shape[1])
                            prog.append(("concat_copy", lambda a=i.out, b=dst: ops.copy_view(a, b)))
                            if n.aux.get('q_whole'):
                                qpack(dst, n.outq.channels(c0, c0 + i.shape[1]))
                        elif n.aux.get('q_whole'):
                            if i.outq is None or i.outq.base is not (n.outq.base if n.outq.base is not None else n.outq):
                                raise NotImplementedError("ConcatLayer input at a channel offset that is not a multiple "
                                                          "of 8 feeding a low-precision convolution")
                        c0 += i.shape[1]
                elif n.aux.get('q_whole'):      # a net input / reshaped tensor that a low-precision product reads
                    qpack(y, n.outq)
                continue
            x = n.inputs[0].out
            xq = n.inputs[0].outq
            q_direct = False                # did the node's own kernel write n.outq?
            a = n.act
            if n.op in ('conv', 'dense'):
                d = self._desc(n, x, y)
                w, b = st.value(n.layer.W), st.value(n.layer.b)
                bnn = self._conv_bn_fusable(n, d, xq, deterministic)
                if bnn is not None:
                    # small maps: Conv2DLayer -> BatchNormLayer (-> nonlinearity) as ONE product -- the finishing kernel of
                    # the convolution holds the whole map of its channels (csrc/conv_small.hip)
                    lb = bnn.layer
                    fused_bn.add(id(bnn))
                    y32 = bnn.out if (bnn.outq is None or self._fp32_needed(bnn)) else None
                    rm, ri = (st.value(lb.mean), st.value(lb.inv_std)) if update_running else (None, None)
                    if not self._lp(d, 0):
                        prog.append(("conv_bn_fwd", lambda d=d, x=x, w=w, b=b, co=y, y32=bnn.out, lb=lb, rm=rm, ri=ri,
                                     m=bnn.aux['mean'], iv=bnn.aux['inv'], ba=bnn.act:
                                     ops.conv2d_bn_fwd(d, x, w, b, co, y32, st.value(lb.gamma), st.value(lb.beta), m, iv, rm, ri,
                                                       lb.epsilon, lb.alpha, ba.kind, ba.alpha), conv_meta(ops, d, 0)))
                        continue
                    wq = self._lp_pack_entry(prog, d, w, ('w', id(n.layer.W)), False, lp_done)
                    prog.append(("conv_bn_fwd", lambda d=d, xq=xq, wq=wq, b=b, co=y, y32=y32, yq=bnn.outq, lb=lb, rm=rm, ri=ri,
                                 m=bnn.aux['mean'], iv=bnn.aux['inv'], ba=bnn.act:
                                 ops.conv2d_bn_fwd_lp_q(d, xq, wq, b, co, y32, yq, st.value(lb.gamma), st.value(lb.beta), m, iv,
                                                        rm, ri, lb.epsilon, lb.alpha, self.dtype, ba.kind, ba.alpha),
                                 conv_meta(ops, d, 0, self.dtype)))
                elif n.op == 'conv' and self._lp(d, 0):
                    wq = self._lp_pack_entry(prog, d, w, ('w', id(n.layer.W)), False, lp_done)
                    if xq is not None:
                        q_direct = n.outq is not None and ops.lp_q_direct(d, 0, self.dtype)
                        yq = n.outq if q_direct else None
                        y32 = None if (q_direct and self._act_fp32_dropped(n)) else y
                        prog.append(("conv_fwd", lambda d=d, xq=xq, wq=wq, b=b, y=y32, yq=yq, a=a:
                                     ops.conv2d_fwd_lp_q(d, xq, wq, b, y, yq, self.dtype, a.kind, a.alpha),
                                     conv_meta(ops, d, 0, self.dtype)))
                    else:
                        prog.append(("conv_fwd", lambda d=d, x=x, wq=wq, b=b, y=y, a=a:
                                     ops.conv2d_fwd_lp(d, x, wq, b, y, self.dtype, a.kind, a.alpha),
                                     conv_meta(ops, d, 0, self.dtype)))
                elif n.op == 'conv' and n.outq is not None and ops.thin_fwd_q_supported(d, a.kind, False, self.dtype):
                    q_direct = True         # a first layer (fp32 operands) whose epilogue also writes the q copy
                    y32 = None if self._act_fp32_dropped(n) else y
                    prog.append(("conv_fwd", lambda d=d, x=x, w=w, b=b, y=y32, yq=n.outq, a=a:
                                 ops.conv2d_fwd_thin_q(d, x, w, b, y, yq, a.kind, a.alpha),
                                 conv_meta(ops, d, 0, moved=4.0 * d.N * d.C * d.H * d.W
                                           + ((0.0 if y32 is None else 4.0) + self._q_bytes()) * d.N * d.K * d.Ho * d.Wo)))
                else:
                    prog.append(("%s_fwd" % n.op, lambda d=d, x=x, w=w, b=b, y=y, a=a:
                                 ops.conv2d_fwd(d, x, w, b, y, a.kind, a.alpha), conv_meta(ops, d, 0)))
            elif n.op == 'convpool':
                d = self._desc(n, x, self._full(n))
                w, b = st.value(n.layer.W), st.value(n.layer.b)
                form = ops.conv_pool_supported(d, a.kind, self.dtype)
                assert form in (1, 2), "fused conv + pool no longer served for %r" % (n,)
                wsrc, dt = w, 'f32'
                if form == 2:
                    wsrc, dt = self._lp_pack_entry(prog, d, w, ('w', id(n.layer.W)), False, lp_done), self.dtype
                if self._pool_y_dropped(n):
                    y = None                # (pooled fp32 tensor not written: see _pool_y_dropped)
                if form == 2 and xq is not None:
                    q_direct = n.outq is not None
                    prog.append(("convpool_fwd", lambda d=d, xq=xq, wsrc=wsrc, b=b, y=y, yq=n.outq, m=n.aux['mask'], a=a:
                                 ops.conv2d_fwd_pool_lp_q(d, xq, wsrc, b, y, yq, m, a.kind, a.alpha, self.dtype),
                                 conv_meta(ops, d, 0, dt, pooled=True)))
                elif form == 1 and n.outq is not None and ops.thin_fwd_q_supported(d, a.kind, True, self.dtype):
                    q_direct = True
                    prog.append(("convpool_fwd", lambda d=d, x=x, w=w, b=b, y=y, yq=n.outq, m=n.aux['mask'], a=a:
                                 ops.conv2d_fwd_pool_thin_q(d, x, w, b, y, m, yq, a.kind, a.alpha),
                                 conv_meta(ops, d, 0, 'f32', pooled=True,
                                           moved=4.0 * d.N * d.C * d.H * d.W + ((0.0 if y is None else 4.0) + 1.0 + self._q_bytes())
                                           * d.N * d.K * (d.Ho // 2) * (d.Wo // 2))))
                else:
                    prog.append(("convpool_fwd", lambda d=d, x=x, wsrc=wsrc, b=b, y=y, m=n.aux['mask'], a=a, dt=dt:
                                 ops.conv2d_fwd_pool(d, x, wsrc, b, y, m, a.kind, a.alpha, dt),
                                 conv_meta(ops, d, 0, dt, pooled=True)))
            elif n.op == 'deconv':
                d = self._desc(n, y, x)
                w, b = st.value(n.layer.W), st.value(n.layer.b)
                prog.append(("deconv_fwd", lambda d=d, x=x, w=w, b=b, y=y, a=a:
                             ops.conv2d_dgrad(d, x, w, y, b, a.kind, a.alpha), conv_meta(ops, d, 1)))
            elif n.op == 'bn' and getattr(n, 'instance', False):
                l = n.layer
                g, be = st.value(l.gamma), st.value(l.beta)
                prog.append(("in_fwd", lambda x=x, y=y, m=n.aux['mean'], iv=n.aux['inv'], g=g, be=be, l=l, a=a, grp=n.aux['group']:
                             ops.instance_norm_fwd(x, y, m, iv, g, be, self.bn_ws, l.epsilon, a.kind, a.alpha, grp)))
            elif n.op == 'bn':
                l = n.layer
                g, be = st.value(l.gamma), st.value(l.beta)
                rm, ri = st.value(l.mean), st.value(l.inv_std)
                if deterministic:
                    prog.append(("bn_apply_det", lambda x=x, y=y, rm=rm, ri=ri, g=g, be=be, a=a:
                                 ops.bn_apply(x, y, rm, ri, g, be, a.kind, a.alpha)))
                elif self.bn_groups == 2:
                    # per-half statistics; the running statistics take the update of the SECOND half only: Lasagne
                    # attaches one default_update per get_output call to the same storage, both computed from the
                    # old value, so one of them survives (unspecified which; the later call is assumed here)
                    hb = x.N // 2
                    for h in (0, 1):
                        xs, ys = x.samples(h * hb, (h + 1) * hb), y.samples(h * hb, (h + 1) * hb)
                        m, iv = n.aux['mean_g'][h], n.aux['inv_g'][h]
                        upd = update_running and h == 1
                        prog.append(("bn_fwd", lambda xs=xs, ys=ys, m=m, iv=iv, rm=rm, ri=ri, l=l, upd=upd, g=g, be=be, a=a:
                                     ops.bn_forward(xs, ys, m, iv, g, be, self.bn_ws, rm if upd else None,
                                                    ri if upd else None, l.epsilon, l.alpha, a.kind, a.alpha)))
                elif self._bn_hi(n):
                    # BatchNorm of a collapsed up-sample convolution: statistics, then normalise + activation written
                    # straight in the interleaved layout of the pp_to_hi node behind it (fp32 and / or q); the parity-planar
                    # result is never stored (the backward pass recomputes it from x)
                    m, iv = n.aux['mean'], n.aux['inv']
                    upd = update_running
                    sh = n.consumers[0]
                    fused_hi.add(id(sh))
                    hi32 = sh.out if (sh.outq is None or self._fp32_needed(sh)) else None
                    prog.append(("bn_fwd", lambda x=x, m=m, iv=iv, rm=rm, ri=ri, l=l, upd=upd:
                                 ops.bn_stats(x, m, iv, self.bn_ws, rm if upd else None, ri if upd else None, l.epsilon, l.alpha)))
                    prog.append(("bn_fwd", lambda x=x, hi32=hi32, hiq=sh.outq if self.q_epi else None, m=m, iv=iv, g=g, be=be, a=a:
                                 ops.bn_apply_hi(x, hi32, hiq, m, iv, g, be, a.kind, a.alpha)))
                elif n.outq is not None and self.q_epi and x.HW % 2 == 0 and x.nstride % 2 == 0 and y.nstride % 2 == 0:
                    # statistics, then normalise + activation writing the fp32 result AND its q copy in one pass
                    m, iv = n.aux['mean'], n.aux['inv']
                    upd = update_running
                    q_direct = True
                    prog.append(("bn_fwd", lambda x=x, m=m, iv=iv, rm=rm, ri=ri, l=l, upd=upd:
                                 ops.bn_stats(x, m, iv, self.bn_ws, rm if upd else None, ri if upd else None, l.epsilon, l.alpha)))
                    y32 = y if (self._fp32_needed(n) or os.environ.get("GHM_BN_FP32") is not None) else None
                    prog.append(("bn_fwd", lambda x=x, y32=y32, yq=n.outq, m=m, iv=iv, g=g, be=be, a=a:
                                 ops.bn_apply_q(x, y32, m, iv, g, be, yq, a.kind, a.alpha)))
                else:
                    m, iv = n.aux['mean'], n.aux['inv']
                    upd = update_running
                    prog.append(("bn_fwd", lambda x=x, y=y, m=m, iv=iv, rm=rm, ri=ri, l=l, upd=upd, g=g, be=be, a=a:
                                 ops.bn_forward(x, y, m, iv, g, be, self.bn_ws, rm if upd else None, ri if upd else None,
                                                l.epsilon, l.alpha, a.kind, a.alpha)))
            elif n.op == 'upconv':
                d = self._upconv_desc(n, x)
                w5, b = st.value(n.layer.W), st.value(n.layer.b)
                wpc, b4 = n.aux['wpc'], n.aux['b4']
                C, K = x.Cc, n.shape[1]
                ulab = 'blconv' if n.attrs.get('mode') == 1 else 'upconv'     # (bench.py prices 'upconv' launches at 25 / 9)
                y4 = y.reshape((x.N, 4 * K, x.H, x.W))
                if self._lp(d, 0):
                    wq = self._lp_pack_entry(prog, d, wpc, ('c', id(n.layer.W)), False, None)    # after collapse_w
                    if xq is not None and self._bl_skip(n, d, 0) and a == linear:
                        # the bilinear form's structurally zero taps skipped (25 of 36 k-steps): bit-identical to the full kernel
                        prog.append((ulab + "_fwd", lambda d=d, xq=xq, wq=wq, b4=b4, y4=y4:
                                     ops.blconv_fwd_split(d, xq, wq, b4, y4, self.dtype), bl_meta(conv_meta(ops, d, 0, self.dtype))))
                    elif xq is not None:
                        prog.append((ulab + "_fwd", lambda d=d, xq=xq, wq=wq, b4=b4, y4=y4, a=a:
                                     ops.conv2d_fwd_lp_q(d, xq, wq, b4, y4, None, self.dtype, a.kind, a.alpha),
                                     conv_meta(ops, d, 0, self.dtype)))
                    else:
                        prog.append((ulab + "_fwd", lambda d=d, x=x, wq=wq, b4=b4, y4=y4, a=a:
                                     ops.conv2d_fwd_lp(d, x, wq, b4, y4, self.dtype, a.kind, a.alpha),
                                     conv_meta(ops, d, 0, self.dtype)))
                else:
                    prog.append((ulab + "_fwd", lambda d=d, x=x, wpc=wpc, b4=b4, y4=y4, a=a:
                                 ops.conv2d_fwd(d, x, wpc, b4, y4, a.kind, a.alpha), conv_meta(ops, d, 0)))
                if n.attrs.get('mode') == 1:
                    if a != linear:
                        raise NotImplementedError("bilinear up-sample convolution with its own nonlinearity")
                    # what the zero-extended coarse convolution leaves out: Theano's border rows / columns (conv_bilinear.hip)
                    prog.append(("blconv_frame_fwd", lambda x=x, w5=w5, y4=y4, K=K, fl=n.aux['fl']: ops.blconv_frame_fwd(x, w5, y4, K, fl)))
            elif n.op == 'pp_to_hi':
                if id(n) in fused_hi:
                    q_direct = self.q_epi                          # written by the BatchNorm in front of it
                elif n.outq is not None and self.q_epi and y.nstride % 2 == 0:
                    q_direct = True
                    y32 = y if self._fp32_needed(n) else None      # every consumer reads the q copy: no fp32 tensor
                    prog.append(("pp_to_hi", lambda x=x, y32=y32, yq=n.outq: ops.pp_to_hi_q(x, y32, yq)))
                else:
                    prog.append(("pp_to_hi", lambda x=x, y=y: ops.pp_to_hi(x, y)))
            elif n.op == 'dropout':
                if deterministic:
                    prog.append(("dropout_det", lambda x=x, y=y: ops.copy_view(x, y)))
                else:
                    prog.append(("dropout_fwd", lambda x=x, y=y, p=n.attrs['p'], k=n.aux['key'], c=self.rng_counter:
                                 ops.dropout(x, y, p, k, c)))
            elif n.op == 'act':
                prog.append(("act_fwd", lambda x=x, y=y, a=a: ops.act_fwd(x, y, a.kind, a.alpha)))
            elif n.op == 'up_nearest':
                prog.append(("up_nearest_fwd", lambda x=x, y=y: ops.upsample_nearest2_fwd(x, y)))
            elif n.op == 'up_bilinear':
                if n.outq is not None and self.q_epi:
                    q_direct = True
                    y32 = y if self._fp32_needed(n) else None
                    prog.append(("up_bilinear_fwd", lambda x=x, y32=y32, yq=n.outq: ops.upsample_bilinear2_fwd_q(x, y32, yq)))
                else:
                    prog.append(("up_bilinear_fwd", lambda x=x, y=y: ops.upsample_bilinear2_fwd(x, y)))
            elif n.op == 'maxpool':
                prog.append(("maxpool_fwd", lambda x=x, y=y: ops.maxpool2_fwd(x, y)))
            elif n.op == 'avgpool':
                prog.append(("avgpool_fwd", lambda x=x, y=y, p=n.attrs['p']: ops.avgpool_fwd(x, y, p)))
            else:
                raise NotImplementedError(n.op)
            if n.outq is not None and not q_direct:
                qpack(y, n.outq)            # producers without a q epilogue of their own: one extra pass

    def emit_transposes(self, prog, transposed):
        """One launch that refreshes every transposed weight copy the data-gradient kernels of this net read
        (instead of one small launch per layer inside emit_backward); call after the forward pass of the step."""
        ops, st = self.ops, self.store
        items = []
        for n in self.order:
            if n.op in ('conv', 'convpool'):
                l = n.layer
                d = self._desc(n, n.inputs[0].out, self._full(n))
                if self._lp(d, 1):
                    continue                      # its data gradient reads the low-precision transposed pack instead
                if self._use_dgrad_t(d, l.W) and id(l.W) not in transposed:
                    transposed.add(id(l.W))
                    items.append((st.value(l.W), st.transposed(l.W), d.C, d.kh * d.kw, d.K))
            elif n.op == 'upconv':
                l = n.layer
                d = self._upconv_desc(n, n.inputs[0].out)
                if self._lp(d, 1):
                    continue
                if d.C > 4 and ops.dgrad_t_supported(d) and ('c', id(l.W)) not in transposed:
                    transposed.add(('c', id(l.W)))
                    items.append((n.aux['wpc'], n.aux['wpcT'], d.C, 9, d.K))
        if items:
            table = ops.transpose_table(items)
            prog.append(("transpose_w", lambda table=table: ops.transpose_weights_batched(table)))

    # ---- backward --------------------------------------------------------------------------------
    def emit_backward(self, prog, seed, nslice=None, wgrad=True, input_grads=(), accumulate_wgrad=False, tag="bwd",
                      transposed=None, on_grads=None, resume=None):
        """Append the backward program.  ``seed``: DevTensor holding dLoss/d(output) (it may be modified in
        place).  ``nslice=(n0, n1)``: run on that sample range of the saved activations.  ``input_grads``:
        InputLayers whose gradient is wanted.  Returns {InputLayer: DevTensor grad}.
        ``on_grads(prog, params)``: called (with wgrad) right after the last launch that writes the gradients of
        ``params`` has been appended -- the data-parallel exchange hangs its sub-bucket all-reduces there (step.py).
        ``resume={node: gradient w.r.t. node.out}`` (instead of ``seed``): start from gradients an EARLIER emit of this plan
        left behind (``grads_of``) and walk on from those nodes only -- the tail of a pass whose head another pass already
        ran (step.py: the generator gradient of a per-sample-scalar discriminator)."""
        ops, st, dev = self.ops, self.store, self.dev
        n0, n1 = nslice if nslice is not None else (0, self.batch)
        nb = n1 - n0
        if transposed is None:
            transposed = set()      # weights already transposed earlier in this step's program

        def sl(t):
            return t if nslice is None else t.samples(n0, n1)

        want_in = {id(self.node_of_layer[id(l)]) for l in input_grads}
        # which nodes need a gradient at all
        req = {}
        for n in self.order:
            has_p = wgrad and n.op in ('conv', 'convpool', 'deconv', 'dense', 'bn', 'upconv')
            req[id(n)] = has_p or any(req[id(i)] for i in n.inputs) or id(n) in want_in
        grads, written = {}, set()
        expands, expand_params, frames = [], [], []
        key = (tag, n0, n1)
        cache = self._scratch.setdefault(key, {})
        for n in self.order:                # flags of an earlier emit of the same (tag, slice): every emit decides afresh
            n.aux.pop(('grad_is_pre', key), None)

        def grad_of(n):
            """gradient buffer w.r.t. n.out (allocated once per (tag, slice))."""
            if id(n) in grads:
                return grads[id(n)]
            if id(n) in cache:
                g = cache[id(n)]
            elif n.alias is not None:
                cat, c0 = n.alias
                g = grad_of(cat).channels(c0, c0 + n.shape[1])
            else:
                g = dev.empty((nb * (n.shape[0] // self.batch),) + tuple(n.shape[1:]))
            cache[id(n)] = g
            grads[id(n)] = g
            return g

        def mark_written(n):
            written.add(id(n))
            if n.op == 'concat':
                for i in n.inputs:
                    if i.alias is not None and i.alias[0] is n:
                        if id(i) in written:
                            raise NotImplementedError("gradient of a concat input written before the concat's own "
                                                      "consumer ran (unsupported graph ordering)")
                        mark_written(i)

        def target(n):
            """-> (grad tensor of n, accumulate flag) for a consumer about to write it."""
            if n.alias is not None and id(n) not in written:
                # a concat input's gradient slice is first written by the concat's own consumer
                raise NotImplementedError("gradient of a ConcatLayer input written before the concat's consumer "
                                          "ran (unsupported graph ordering)")
            return grad_of(n), id(n) in written

        if resume is None:
            grads[id(self.out_node)] = seed
            written.add(id(self.out_node))
            mark_written(self.out_node)
        else:
            for rn, rg in resume.items():
                grads[id(rn)] = rg
                written.add(id(rn))
                mark_written(rn)
                # what the earlier pass left in the buffer of a layer with its own nonlinearity is the gradient in FRONT of it
                # (act_bwd runs in place, or the consumer's data gradient applied it in its epilogue)
                if rn.op in ('conv', 'deconv', 'dense') and rn.act != linear:
                    rn.aux[('grad_is_pre', key)] = True

        def done(*params):
            if on_grads is not None and wgrad:
                on_grads(prog, [p for p in params if p is not None])

        hi_grads = set()        # BatchNorm nodes whose output gradient is held in the interleaved layout of their pp_to_hi reader
        gq_ready = set()        # nodes whose output-gradient q tensor was written by the kernel that produced the gradient
        pooled_c = {}           # convpool nodes whose gradient also exists in the sparse instruction's operand form

        def gradq_of(n, G, pack=True):
            """q copy of the (final) output gradient G of node n: the operand of its low-precision data / weight
            gradient.  Written by G's producer where that kernel has a q epilogue, else packed here in one pass."""
            if not self.use_q or G.Cc % 8:
                return None
            Gq = cache.get(('gq', id(n)))
            if Gq is None:
                Gq = cache[('gq', id(n))] = QTensor.empty(dev, G.shape, self.dtype)
            if pack and id(n) not in gq_ready:
                gq_ready.add(id(n))
                prog.append(("q_pack", lambda G=G, Gq=Gq: ops.q_pack(G, Gq), pack_meta(G)))
            return Gq

        def fused_gq(xin, gi, acc):
            """may the data-gradient kernel that writes gi (the gradient of xin's output) also write its q copy?  Only
            when gi is final as written: single consumer, nothing accumulates into it, no activation backward runs on
            it afterwards, and xin's own backward is a low-precision product that reads it."""
            if not self.use_q or acc or xin.op != 'conv' or len(xin.consumers) != 1 or gi.Cc % 8:
                return None
            if xin.act != linear and not xin.aux.get(('grad_is_pre', key)):
                return None
            dq = self._desc(xin, sl(xin.inputs[0].out), gi)
            if not (self._lp(dq, 1) or self._lp(dq, 2)):
                return None
            gq_ready.add(id(xin))
            return gradq_of(xin, gi, pack=False)

        for n in reversed(self.order):
            if id(n) not in written or not req[id(n)]:
                continue
            G = grad_of(n)
            if n.op in ('input', 'concat'):
                if n.op == 'concat':
                    c0 = 0
                    for i in n.inputs:
                        if (i.alias is None or i.alias[0] is not n) and req[id(i)]:
                            gi, acc = target(i)
                            src = G.channels(c0, c0 + i.shape[1])
                            prog.append(("concat_bwd_copy", lambda a=src, b=gi, acc=acc: ops.copy_view(a, b, acc)))
                            mark_written(i)
                        c0 += i.shape[1]
                continue
            xin = n.inputs[0]
            x, y = sl(xin.out), sl(n.out)
            if n.op == 'convpool' and n.act.kind in ('linear', 'relu', 'lrelu') and os.environ.get("GHM_POOL_READ_Y") is None:
                y = None                    # the mask carries the sign of the pooled activation: its backward never reads it
            a = n.act
            need_dx = req[id(xin)]
            if n.op == 'convpool':
                fs = n.aux['full_shape']
                per = int(np.prod(n.shape[1:]))                  # mask bytes per sample
                # the discriminator's first block (one input channel): both gradients straight from the pooled operands
                # (csrc/conv_pool_bwd.hip) -- the 537 MB full-resolution gradient is neither written nor read
                dS = self._desc(n, x, self._full(n, nb))
                sparse = int(ops.pool_bwd_sparse_supported(dS, a.kind) or 0)
                if (sparse & 1 or not wgrad) and (sparse & 2 or not need_dx) and sparse:
                    l = n.layer
                    mptr = n.aux['mask'] + n0 * per
                    if wgrad:
                        self._grow_wgrad_ws(ops.pool_wgrad_sparse_workspace(dS))
                        gw, gb = st.grad(l.W), st.grad(l.b)
                        wo, wdev = ops, None
                        if self.side is not None:
                            wdev, wo = self.side
                            prog.append(("fork", lambda wdev=wdev: wdev.wait_for(dev), None, wdev))
                        prog.append(("conv_wgrad", lambda dS=dS, x=x, m=mptr, y=y, G=G, gw=gw, gb=gb, a=a, aw=accumulate_wgrad, wo=wo:
                                     wo.conv2d_pool_wgrad_sparse(dS, x, m, y, G, gw, gb, self.wgrad_ws, a.kind, a.alpha, aw),
                                     pool_sparse_meta(dS, 2), wdev))
                        done(l.W, l.b)
                    if need_dx:
                        gi, acc = target(xin)
                        w = st.value(l.W)
                        dG = self._desc(n, gi, self._full(n, nb))        # the kernel strides dx by ITS sample stride, not x's
                        prog.append(("conv_dgrad", lambda dG=dG, m=mptr, y=y, G=G, w=w, gi=gi, a=a, acc=acc:
                                     ops.conv2d_pool_dgrad_sparse(dG, m, y, G, w, gi, a.kind, a.alpha, acc), pool_sparse_meta(dG, 1)))
                        mark_written(xin)
                    continue
                # otherwise: the gradient of the conv's (never materialised in the forward pass) full-resolution output from
                # the arg-max mask, the pooled value (sign -> activation derivative) and the pooled gradient; then an
                # ordinary conv backward
                Gf = cache.get(('full', id(n)))
                if Gf is None:
                    Gf = cache[('full', id(n))] = dev.empty((nb,) + tuple(fs[1:]))
                mptr = n.aux['mask'] + n0 * per
                # with the weight gradients wanted, the same pass also sums what it writes per channel: the bias gradient
                gb_fused = st.grad(n.layer.b) if wgrad else None
                # who reads the full-resolution gradient: the conv's low-precision data / weight gradients read its q copy;
                # the fp32 tensor is written only if a fp32 kernel reads it (thin first layer, geometries not served)
                dF = self._desc(n, x, Gf)
                xq_ = xin.outq if nslice is None else (xin.outq.samples(n0, n1) if xin.outq is not None else None)
                w_q = xq_ is not None and self._wq(dF)
                d_lp = self.use_q and need_dx and self._lp(self._desc(n, sl(xin.out), Gf), 1)
                q_wanted = (wgrad and w_q) or d_lp
                Gfq = gradq_of(n, Gf, pack=False) if (q_wanted and self.q_epi and Gf.Cc % 8 == 0 and Gf.H % 2 == 0
                                                      and Gf.W % 4 == 0) else None
                if Gfq is not None:         # the full-resolution gradient (if anybody reads it) and its q copy in one pass
                    gq_ready.add(id(n))
                    Gf32 = None if ((w_q or not wgrad) and (d_lp or not need_dx)) else Gf
                    prog.append(("maxpool_mask_bwd", lambda m=mptr, y=y, G=G, Gf32=Gf32, Gfq=Gfq, a=a, gb=gb_fused, aw=accumulate_wgrad:
                                 ops.maxpool2_mask_bwd_q(m, y, G, Gf32, Gfq, a.kind, a.alpha, gb, aw)))
                    # the same gradient once more as half-width rows + column bits: the operand of the sparse matrix
                    # instruction, for the 5x5 weight gradient (rows with a tied window row stay on the dense q copy)
                    if wgrad and w_q and ops.wgrad_pooled_split_supported(dF, self.dtype):
                        pc = cache.get(('pooled_c', id(n), nb))
                        if pc is None:
                            Kp, Hf, Wf = Gf.Cc, Gf.H, Gf.W
                            pc = cache[('pooled_c', id(n), nb)] = (
                                QTensor.empty(dev, (nb, Kp, Hf, Wf // 2), self.dtype),
                                dev.alloc(nb * (Kp // 8) * Hf * (Wf // 32) * 16 + 256), dev.alloc(nb * Hf * 4 + 256))
                        # (written by the stream that runs the weight gradient, its only reader: see conv_wgrad below)
                        pooled_c[id(n)] = pc + (mptr, y, G, a)
                else:
                    prog.append(("maxpool_mask_bwd", lambda m=mptr, y=y, G=G, Gf=Gf, a=a, gb=gb_fused, aw=accumulate_wgrad:
                                 ops.maxpool2_mask_bwd(m, y, G, Gf, a.kind, a.alpha, gb, aw)))
                G, a = Gf, linear
            if n.op in ('conv', 'convpool', 'deconv', 'dense'):
                if a != linear and not n.aux.get(('grad_is_pre', key)):
                    prog.append(("act_bwd", lambda G=G, y=y, a=a: ops.act_bwd(G, y, G, a.kind, a.alpha)))
                l = n.layer
                w = st.value(l.W)
                if n.op == 'deconv':
                    d = self._desc(n, G, x)          # conv input side = deconv output grad, output side = x
                else:
                    d = self._desc(n, x, G)
                # the q copy of the output gradient, for the low-precision weight gradient (before its stream forks off)
                xq = None if (nslice is not None and xin.outq is None) else (xin.outq if nslice is None else
                                                                             (xin.outq.samples(n0, n1) if xin.outq is not None else None))
                wq_form = wgrad and n.op in ('conv', 'convpool') and xq is not None and self._wq(d)
                Gq_w = gradq_of(n, G) if wq_form else None
                if wgrad:
                    self._need_wgrad_ws(d)
                    gw, gb = st.grad(l.W), st.grad(l.b)
                    aw = accumulate_wgrad
                    wo, wdev = ops, None
                    if self.side is not None:
                        wdev, wo = self.side
                        prog.append(("fork", lambda wdev=wdev: wdev.wait_for(dev), None, wdev))
                    if Gq_w is not None and id(n) in pooled_c:
                        pc = pooled_c[id(n)]
                        prog.append(("maxpool_mask_compress", lambda pc=pc, wo=wo:
                                     wo.maxpool2_mask_bwd_compress_q(pc[3], pc[4], pc[5], pc[0], pc[1], pc[2], pc[6].kind, pc[6].alpha),
                                     None, wdev))
                        prog.append(("conv_wgrad", lambda d=d, xq=xq, Gq=Gq_w, pc=pc, gw=gw, aw=aw, wo=wo:
                                     wo.conv2d_wgrad_pooled_split(d, xq, Gq, pc[0], pc[1], pc[2], gw, self.wgrad_ws, self.dtype, aw),
                                     sparse_meta(conv_meta(ops, d, 2, self.dtype)), wdev))
                    elif Gq_w is not None:
                        prog.append(("conv_wgrad", lambda d=d, xq=xq, Gq=Gq_w, gw=gw, aw=aw, wo=wo:
                                     wo.conv2d_wgrad_lp_q(d, xq, Gq, gw, self.wgrad_ws, self.dtype, aw),
                                     conv_meta(ops, d, 2, self.dtype), wdev))
                    elif n.op == 'deconv':
                        prog.append(("deconv_wgrad", lambda d=d, G=G, x=x, gw=gw, aw=aw, wo=wo:
                                     wo.conv2d_wgrad(d, G, x, gw, self.wgrad_ws, aw), conv_meta(ops, d, 2), wdev))
                    elif n.op in ('conv', 'convpool') and self._lp(d, 2):
                        prog.append(("conv_wgrad", lambda d=d, G=G, x=x, gw=gw, aw=aw, wo=wo:
                                     wo.conv2d_wgrad_lp(d, x, G, gw, self.wgrad_ws, self.dtype, aw),
                                     conv_meta(ops, d, 2, self.dtype), wdev))
                    else:
                        prog.append(("%s_wgrad" % ('conv' if n.op == 'convpool' else n.op), lambda d=d, G=G, x=x, gw=gw, aw=aw, wo=wo:
                                     wo.conv2d_wgrad(d, x, G, gw, self.wgrad_ws, aw), conv_meta(ops, d, 2), wdev))
                    # a bias that feeds a BatchNorm has an identically zero gradient (the BN backward output sums to
                    # zero per channel): its slice of the zero-initialised gradient buffer is simply never written
                    bn_fed = len(n.consumers) == 1 and n.consumers[0].op == 'bn' and n.act == linear
                    if not bn_fed and n.op != 'convpool':       # convpool: summed by the mask backward pass above
                        prog.append(("bias_grad", lambda G=G, gb=gb, aw=aw, wo=wo: wo.channel_sum(G, gb, aw), None, wdev))
                    done(l.W, l.b)
                if need_dx:
                    gi, acc = target(xin)
                    # the producer's own nonlinearity (a conv -> LeakyRectify -> conv chain without BatchNorm: the
                    # PatchGAN, p2p.py:285-286) differentiated in THIS data gradient's epilogue instead of a separate
                    # read-modify-write pass over the gradient tensor
                    form = 0
                    if (not acc and n.op in ('conv', 'convpool') and xin.op in ('conv', 'deconv', 'dense')
                            and xin.act.kind in ('relu', 'lrelu') and len(xin.consumers) == 1 and not xin.aux.get(('grad_is_pre', key))):
                        form = ops.dgrad_dact_supported(self._desc(n, gi, G), self.dtype) or 0
                    if form:
                        d2 = self._desc(n, gi, G)
                        dt = 'f32'
                        if form == 1:
                            wsel = w
                        elif form == 2:
                            wsel = st.transposed(l.W)
                            if id(l.W) not in transposed:
                                transposed.add(id(l.W))
                                prog.append(("transpose_w", lambda d=d2, w=w, wT=wsel: ops.transpose_weights(d, w, wT)))
                        else:
                            wsel, dt = self._lp_pack_entry(prog, d2, w, ('w', id(l.W)), True, transposed), self.dtype
                        xa = xin.act
                        xin.aux[('grad_is_pre', key)] = True
                        Gq = gradq_of(n, G) if form == 3 else None
                        if Gq is not None and (ops.lp_q_direct(d2, 1, self.dtype) or self.dtype in SPLITS):
                            giq = fused_gq(xin, gi, acc) if ops.lp_q_direct(d2, 1, self.dtype) else None
                            # split modes: the slope from the sign of the producer's q copy (2 bytes per element, and the
                            # producer's fp32 activation need not exist: _act_fp32_dropped)
                            ysrc = x
                            if self.dtype in SPLITS and xin.outq is not None and gi.Cc % 8 == 0 and not os.environ.get("GHM_DACT_FP32"):
                                ysrc = xin.outq if nslice is None else xin.outq.samples(n0, n1)
                            prog.append(("conv_dgrad", lambda d=d2, Gq=Gq, wsel=wsel, gi=gi, giq=giq, x=ysrc, xa=xa:
                                         ops.conv2d_dgrad_dact_lp_q(d, Gq, wsel, gi, giq, x, xa.kind, xa.alpha, self.dtype),
                                         conv_meta(ops, d2, 3, dt, extra=" +dact" + (" +q" if giq is not None else ""))))
                        else:
                            prog.append(("conv_dgrad", lambda d=d2, G=G, wsel=wsel, gi=gi, x=x, xa=xa, dt=dt:
                                         ops.conv2d_dgrad_dact(d, G, wsel, gi, x, xa.kind, xa.alpha, dt),
                                         conv_meta(ops, d2, 3 if form != 1 else 1, dt)))
                    elif n.op == 'deconv':
                        d2 = self._desc(n, G, gi)
                        prog.append(("deconv_dgrad", lambda d=d2, G=G, w=w, gi=gi, acc=acc:
                                     ops.conv2d_fwd(d, G, w, None, gi, 'linear', 0.0, acc), conv_meta(ops, d2, 0)))
                    elif n.op in ('conv', 'convpool') and self._lp(self._desc(n, gi, G), 1):
                        d2 = self._desc(n, gi, G)
                        wqT = self._lp_pack_entry(prog, d2, w, ('w', id(l.W)), True, transposed)
                        Gq = gradq_of(n, G)
                        if Gq is not None:
                            giq = fused_gq(xin, gi, acc) if ops.lp_q_direct(d2, 1, self.dtype) else None
                            if giq is None:
                                gq_ready.discard(id(xin))
                            prog.append(("conv_dgrad", lambda d=d2, Gq=Gq, wqT=wqT, gi=gi, giq=giq, acc=acc:
                                         ops.conv2d_dgrad_lp_q(d, Gq, wqT, gi, giq, self.dtype, None, 'linear', 0.0, acc),
                                         conv_meta(ops, d2, 3, self.dtype,
                                                   extra=(" +q" if giq is not None else "") + (" +acc" if acc else ""))))
                        else:
                            prog.append(("conv_dgrad", lambda d=d2, G=G, wqT=wqT, gi=gi, acc=acc:
                                         ops.conv2d_dgrad_lp(d, G, wqT, gi, self.dtype, None, 'linear', 0.0, acc),
                                         conv_meta(ops, d2, 3, self.dtype)))
                    elif n.op in ('conv', 'convpool') and self._use_dgrad_t(self._desc(n, gi, G), l.W):
                        # data gradient as a forward-form conv on the transposed weights (LDS-patch kernels)
                        d2 = self._desc(n, gi, G)
                        wT = st.transposed(l.W)
                        if id(l.W) not in transposed:
                            transposed.add(id(l.W))
                            prog.append(("transpose_w", lambda d=d2, w=w, wT=wT: ops.transpose_weights(d, w, wT)))
                        prog.append(("conv_dgrad", lambda d=d2, G=G, wT=wT, gi=gi, acc=acc:
                                     ops.conv2d_dgrad_t(d, G, wT, gi, None, 'linear', 0.0, acc), conv_meta(ops, d2, 3)))
                    else:
                        d2 = self._desc(n, gi, G)
                        prog.append(("%s_dgrad" % ('conv' if n.op == 'convpool' else n.op), lambda d=d2, G=G, w=w, gi=gi, acc=acc:
                                     ops.conv2d_dgrad(d, G, w, gi, None, 'linear', 0.0, acc), conv_meta(ops, d2, 1)))
                    mark_written(xin)
            elif n.op == 'upconv':
                if nslice is not None:
                    raise NotImplementedError("sample slices through a collapsed up-sample convolution")
                if a != linear:
                    prog.append(("act_bwd", lambda G=G, y=y, a=a: ops.act_bwd(G, y, G, a.kind, a.alpha)))
                l = n.layer
                d = self._upconv_desc(n, x)
                C, K = x.Cc, n.shape[1]
                G4 = G.reshape((x.N, 4 * K, x.H, x.W))
                wpc, wpcT, dwpc = n.aux['wpc'], n.aux['wpcT'], n.aux['dwpc']
                xq = xin.outq
                G4q_w = gradq_of(n, G4) if (wgrad and xq is not None and self._wq(d)) else None
                bl = n.attrs.get('mode') == 1
                ulab = 'blconv' if bl else 'upconv'
                if bl:      # the six border lines of the fine gradient, for both frame gradients (before the gradient stream forks)
                    prog.append(("blconv_frame_gather", lambda G4=G4, C=C, K=K, dyl=n.aux['dyl']: ops.blconv_frame_gather(G4, C, K, dyl)))
                if wgrad:
                    self._need_wgrad_ws(d)
                    gw, gb = st.grad(l.W), st.grad(l.b)
                    aw = accumulate_wgrad
                    wo, wdev = ops, None
                    if self.side is not None:
                        wdev, wo = self.side
                        prog.append(("fork", lambda wdev=wdev: wdev.wait_for(dev), None, wdev))
                    if G4q_w is not None and self._bl_skip(n, d, 2):
                        prog.append((ulab + "_wgrad", lambda d=d, xq=xq, G4q=G4q_w, dwpc=dwpc, wo=wo:
                                     wo.blconv_wgrad_split(d, xq, G4q, dwpc, self.wgrad_ws, self.dtype, False),
                                     bl_meta(conv_meta(ops, d, 2, self.dtype)), wdev))
                    elif G4q_w is not None:
                        prog.append((ulab + "_wgrad", lambda d=d, xq=xq, G4q=G4q_w, dwpc=dwpc, wo=wo:
                                     wo.conv2d_wgrad_lp_q(d, xq, G4q, dwpc, self.wgrad_ws, self.dtype, False),
                                     conv_meta(ops, d, 2, self.dtype), wdev))
                    elif self._lp(d, 2):
                        prog.append((ulab + "_wgrad", lambda d=d, x=x, G4=G4, dwpc=dwpc, wo=wo:
                                     wo.conv2d_wgrad_lp(d, x, G4, dwpc, self.wgrad_ws, self.dtype, False),
                                     conv_meta(ops, d, 2, self.dtype), wdev))
                    else:
                        prog.append((ulab + "_wgrad", lambda d=d, x=x, G4=G4, dwpc=dwpc, wo=wo:
                                     wo.conv2d_wgrad(d, x, G4, dwpc, self.wgrad_ws, False), conv_meta(ops, d, 2), wdev))
                    expands.append((dwpc, gw, C, K, n.attrs.get('mode', 0)))      # 3x3 -> 5x5 (-> fine 3x3) gradient expansion: one launch for all layers, below
                    if bl:
                        frames.append((n.aux['dyl'], n.aux['fl'], gw, x.N, C, K, x.H, x.W))
                    bn_fed = len(n.consumers) == 1 and n.consumers[0].op == 'bn' and n.act == linear
                    if not bn_fed:
                        prog.append(("bias_grad", lambda G=G, gb=gb, aw=aw, wo=wo: wo.channel_sum(G, gb, aw), None, wdev))
                    done(l.b)                              # l.W: written by the batched 3x3 -> 5x5 expansion below
                    expand_params.append(l.W)
                if need_dx:
                    gi, acc = target(xin)
                    if self._lp(d, 1):
                        wqT = self._lp_pack_entry(prog, d, wpc, ('c', id(l.W)), True, transposed)
                        G4q = gradq_of(n, G4)
                        if G4q is not None and self._bl_skip(n, d, 1):
                            prog.append((ulab + "_dgrad", lambda d=d, G4q=G4q, wqT=wqT, gi=gi, acc=acc:
                                         ops.blconv_dgrad_split(d, G4q, wqT, gi, self.dtype, acc),
                                         bl_meta(conv_meta(ops, d, 3, self.dtype))))
                        elif G4q is not None:
                            prog.append((ulab + "_dgrad", lambda d=d, G4q=G4q, wqT=wqT, gi=gi, acc=acc:
                                         ops.conv2d_dgrad_lp_q(d, G4q, wqT, gi, None, self.dtype, None, 'linear', 0.0, acc),
                                         conv_meta(ops, d, 3, self.dtype)))
                        else:
                            prog.append((ulab + "_dgrad", lambda d=d, G4=G4, wqT=wqT, gi=gi, acc=acc:
                                         ops.conv2d_dgrad_lp(d, G4, wqT, gi, self.dtype, None, 'linear', 0.0, acc),
                                         conv_meta(ops, d, 3, self.dtype)))
                    elif C > 4 and ops.dgrad_t_supported(d):
                        if ('c', id(l.W)) not in transposed:
                            transposed.add(('c', id(l.W)))
                            prog.append(("transpose_w", lambda d=d, wpc=wpc, wpcT=wpcT: ops.transpose_weights(d, wpc, wpcT)))
                        prog.append((ulab + "_dgrad", lambda d=d, G4=G4, wpcT=wpcT, gi=gi, acc=acc:
                                     ops.conv2d_dgrad_t(d, G4, wpcT, gi, None, 'linear', 0.0, acc), conv_meta(ops, d, 3)))
                    else:
                        prog.append((ulab + "_dgrad", lambda d=d, G4=G4, wpc=wpc, gi=gi, acc=acc:
                                     ops.conv2d_dgrad(d, G4, wpc, gi, None, 'linear', 0.0, acc), conv_meta(ops, d, 1)))
                    if bl:
                        prog.append(("blconv_frame_dgrad", lambda dyl=n.aux['dyl'], w3=st.value(l.W), gi=gi, K=K:
                                     ops.blconv_frame_dgrad(dyl, w3, gi, K)))
                    mark_written(xin)
            elif n.op == 'dropout':
                if need_dx:
                    gi, acc = target(xin)
                    if acc or nslice is not None:
                        raise NotImplementedError("DropoutLayer input with several consumers / sample slices")
                    # same key, same counter value as the forward pass of this step -> the same mask
                    prog.append(("dropout_bwd", lambda G=G, gi=gi, p=n.attrs['p'], k=n.aux['key'], c=self.rng_counter:
                                 ops.dropout(G, gi, p, k, c)))
                    mark_written(xin)
            elif n.op == 'pp_to_hi':
                if need_dx and nslice is None and self._bn_hi(xin) and G.nstride % 2 == 0:
                    # the BatchNorm backward reads this gradient through the inverse permutation: no hi_to_pp pass
                    grads[id(xin)] = G
                    hi_grads.add(id(xin))
                    mark_written(xin)
                elif need_dx:
                    gi, acc = target(xin)
                    if acc or nslice is not None:
                        raise NotImplementedError("parity-planar tensor with several consumers / sample slices")
                    prog.append(("hi_to_pp", lambda G=G, gi=gi: ops.hi_to_pp(G, gi)))
                    mark_written(xin)
            elif n.op == 'bn':
                l = n.layer
                # (the backward kernels recompute y = act(bn(x)) from x instead of reading the output tensor: bit-identical
                # to the forward value, one tensor less to read in both of their passes)
                gam, bet = st.value(l.gamma), st.value(l.beta)
                if wgrad:
                    dg, db, aw = st.grad(l.gamma), st.grad(l.beta), accumulate_wgrad
                else:
                    C = n.shape[1]
                    dg, db, aw = self._bn_scratch.channels(0, C), self._bn_scratch.channels(C, 2 * C), False
                gi, acc = target(xin)
                dst = gi
                if acc:
                    dst = dev.empty(gi.shape) if ('bn_tmp', id(n)) not in cache else cache[('bn_tmp', id(n))]
                    cache[('bn_tmp', id(n))] = dst
                # the convolution in front of this BatchNorm reads gi as the operand of its low-precision data / weight
                # gradients: the apply pass writes the q copy itself -- and no fp32 gradient at all when both read q
                giq, gi32 = None, True
                inst = getattr(n, 'instance', False)
                if (self.q_epi and not acc and nslice is None and self.bn_groups == 1 and not inst and xin.op in ('conv', 'upconv')
                        and len(xin.consumers) == 1 and xin.act == linear and gi.HW % 2 == 0 and gi.Cc % 8 == 0
                        and gi.nstride % 2 == 0):
                    xx = xin.inputs[0]
                    if xin.op == 'conv':
                        dq, gview = self._desc(xin, xx.out, gi), gi
                    else:
                        dq = self._upconv_desc(xin, xx.out)
                        gview = gi.reshape((xx.out.N, 4 * xin.shape[1], xx.out.H, xx.out.W))
                    w_q = xx.outq is not None and self._wq(dq)
                    d_q = self._lp(dq, 1) or not req[id(xx)]
                    if (w_q or not wgrad) and (self._lp(dq, 1) or w_q):
                        giq = gradq_of(xin, gview, pack=False).reshape(gi.shape)
                        gq_ready.add(id(xin))
                        gi32 = not ((w_q or not wgrad) and d_q) or xin.attrs.get('mode') == 1     # (the frame reads its border lines)
                if inst:
                    if nslice is not None:
                        raise NotImplementedError("InstanceNorm backward on a sample slice")
                    prog.append(("in_bwd", lambda G=G, x=x, dst=dst, m=n.aux['mean'], iv=n.aux['inv'], gam=gam, bet=bet, dg=dg, db=db,
                                 a=a, aw=aw, grp=n.aux['group']:
                                 ops.instance_norm_bwd(G, x, dst, m, iv, gam, bet, dg, db, self.bn_ws, a.kind, a.alpha, aw, grp)))
                elif id(n) in hi_grads:
                    m, iv = n.aux['mean'], n.aux['inv']
                    dst32 = dst if (giq is None or gi32) else None
                    prog.append(("bn_bwd", lambda G=G, x=x, dst32=dst32, giq=giq, m=m, iv=iv, gam=gam, bet=bet, dg=dg, db=db, a=a, aw=aw:
                                 ops.bn_backward_hi(G, x, dst32, giq, m, iv, gam, bet, dg, db, self.bn_ws, a.kind, a.alpha, aw)))
                elif giq is not None:
                    m, iv = n.aux['mean'], n.aux['inv']
                    dst32 = dst if gi32 else None
                    prog.append(("bn_bwd", lambda G=G, x=x, dst32=dst32, giq=giq, m=m, iv=iv, gam=gam, bet=bet, dg=dg, db=db, a=a, aw=aw:
                                 ops.bn_backward_q(G, None, x, dst32, m, iv, gam, dg, db, self.bn_ws, giq, a.kind, a.alpha, aw, bet)))
                elif self.bn_groups == 2:
                    hb = self.batch // 2
                    halves = (0, 1) if nslice is None else ((n0 // hb,) if (n1 - n0) == hb and n0 % hb == 0 else None)
                    if halves is None:
                        raise NotImplementedError("sample slice that is not one half of a [real | fake] batch")
                    for idx, h in enumerate(halves):
                        sub = (lambda t, h=h: t.samples(h * hb, (h + 1) * hb)) if nslice is None else (lambda t: t)
                        m, iv = n.aux['mean_g'][h], n.aux['inv_g'][h]
                        awh = aw or idx > 0          # the second half adds to dgamma / dbeta
                        prog.append(("bn_bwd", lambda G=sub(G), x=sub(x), dst=sub(dst), m=m, iv=iv, gam=gam, bet=bet,
                                     dg=dg, db=db, a=a, awh=awh:
                                     ops.bn_backward_x(G, x, dst, m, iv, gam, bet, dg, db, self.bn_ws, a.kind, a.alpha, awh)))
                else:
                    m, iv = n.aux['mean'], n.aux['inv']
                    prog.append(("bn_bwd", lambda G=G, x=x, dst=dst, m=m, iv=iv, gam=gam, bet=bet, dg=dg, db=db, a=a, aw=aw:
                                 ops.bn_backward_x(G, x, dst, m, iv, gam, bet, dg, db, self.bn_ws, a.kind, a.alpha, aw)))
                if acc:
                    prog.append(("bn_bwd_acc", lambda dst=dst, gi=gi: ops.copy_view(dst, gi, True)))
                done(l.gamma, l.beta)
                mark_written(xin)
            elif n.op == 'act':
                if need_dx:
                    gi, acc = target(xin)
                    prog.append(("act_bwd", lambda G=G, y=y, gi=gi, a=a, acc=acc:
                                 ops.act_bwd(G, y, gi, a.kind, a.alpha, acc)))
                    mark_written(xin)
            elif n.op == 'reshape':
                if need_dx:
                    if id(xin) in grads or id(xin) in cache or xin.alias is not None:
                        gi, acc = target(xin)
                        prog.append(("reshape_bwd", lambda G=G, gi=gi, acc=acc:
                                     ops.copy_view(G.reshape(gi.shape), gi, acc)))
                    else:
                        grads[id(xin)] = cache[id(xin)] = G.reshape((nb,) + tuple(xin.shape[1:]))
                    mark_written(xin)
            elif n.op in ('up_nearest', 'up_bilinear'):
                if need_dx:
                    gi, acc = target(xin)
                    fn = ops.upsample_nearest2_bwd if n.op == 'up_nearest' else ops.upsample_bilinear2_bwd
                    prog.append(("%s_bwd" % n.op, lambda G=G, gi=gi, acc=acc, fn=fn: fn(G, gi, acc)))
                    mark_written(xin)
            elif n.op == 'maxpool':
                if need_dx:
                    gi, acc = target(xin)
                    if acc:
                        raise NotImplementedError("maxpool input with several consumers")
                    fa = linear
                    if xin.op in ('conv', 'deconv', 'dense') and len(xin.consumers) == 1 and xin.act.kind in ('lrelu', 'relu'):
                        # fold the producer's LeakyReLU/ReLU backward into the pooling scatter (mask from the output sign)
                        fa = xin.act
                        xin.aux[('grad_is_pre', key)] = True
                    prog.append(("maxpool_bwd", lambda x=x, y=y, G=G, gi=gi, fa=fa:
                                 ops.maxpool2_bwd(x, y, G, gi, fa.kind, fa.alpha)))
                    mark_written(xin)
            elif n.op == 'avgpool':
                if need_dx:
                    gi, acc = target(xin)
                    if acc:
                        raise NotImplementedError("avgpool input with several consumers")
                    prog.append(("avgpool_bwd", lambda G=G, gi=gi, p=n.attrs['p']: ops.avgpool_bwd(G, gi, p)))
                    mark_written(xin)
            else:
                raise NotImplementedError(n.op)
        if expands:
            # after the last collapsed layer's weight gradient, on the lane the weight gradients run on
            wo, wdev = (self.side[1], self.side[0]) if self.side is not None else (ops, None)
            tkey = ('expand', tuple(int(e[0].ptr) for e in expands))
            tab = self._scratch.setdefault('tables', {}).get(tkey)
            if tab is None:
                tab = self._scratch['tables'][tkey] = wo.expand_table(expands)
            prog.append(("expand_wgrad", lambda tab=tab, aw=accumulate_wgrad, wo=wo: wo.upconv_expand_batched(tab, aw),
                         None, wdev))
            for dyl, fl, gw, N_, C_, K_, h_, w_ in frames:       # the frame's share of the fine weight gradients, on top
                prog.append(("blconv_frame_wgrad", lambda dyl=dyl, fl=fl, gw=gw, N_=N_, C_=C_, K_=K_, h_=h_, w_=w_, wo=wo:
                             wo.blconv_frame_wgrad(dyl, fl, gw, N_, C_, K_, h_, w_), None, wdev))
            done(*expand_params)
        self._last_grads = dict(grads)
        return {l: (grad_of(self.node_of_layer[id(l)]) if id(self.node_of_layer[id(l)]) in written else None)
                for l in input_grads}

    def grads_of(self, node):
        """the buffer the LAST emit_backward of this plan holds the gradient w.r.t. ``node.out`` in (None: never written)"""
        return getattr(self, '_last_grads', {}).get(id(node))


def conv_meta(ops, d, kind, dtype='f32', pooled=False, extra='', moved=None):
    """roofline metadata of one conv launch: kernel variant name and ALGORITHMIC flops (2 x MACs)."""
    if pooled:
        name = ("sp_conv2_kernel<%d, %d> fwd+pool" % (d.kh, d.stride)) if dtype in SPLITS else \
            ("lp_conv_kernel<%s, %d, %d>" % (dtype, d.kh, d.stride)) if dtype != 'f32' else \
            ("fanout_kernel<fwd+pool>" if d.C <= 4 else ops.conv_variant(d, 0).split(" splits")[0])   # same kernel, pooled epilogue
    elif dtype != 'f32':
        fam = "wgrad" if kind == 2 else ("dgrad_s2" if kind in (1, 3) and d.stride == 2 else "conv")
        name = "lp_%s_kernel<%s, %d, %d>" % (fam, dtype, d.kh, d.stride)
        if hasattr(ops, 'conv_variant_lp'):         # (small maps run their own kernel family behind the same entry points)
            name = ops.conv_variant_lp(d, {0: 0, 1: 1, 3: 1, 2: 2}[kind], dtype)
    else:
        name = ops.conv_variant(d, kind)
    # algorithmic HBM bytes of the launch (the wide tensor(s) once): conv input + conv output, fp32; a fused
    # conv + pool writes the pooled tensor and a 1-byte mask instead of the full-resolution output
    xb, yb = 4.0 * d.N * d.C * d.H * d.W, 4.0 * d.N * d.K * d.Ho * d.Wo
    if pooled:
        yb = yb / 4 + yb / 16
    # ``moved``: the bytes the launch really moves where they differ from the algorithmic ones (a first layer that also writes
    # the q copy of its result: 2 bytes per piece and element more; a pooled one whose fp32 tensor is not written: 4 less)
    return {"kernel": name, "dtype": dtype, "bytes": xb + yb, "moved_bytes": (xb + yb) if moved is None else moved,
            "thin": min(d.C, d.K) <= 4,
            "flops": 2.0 * d.N * d.K * d.Ho * d.Wo * d.C * d.kh * d.kw,
            "geom": "N%d C%d %dx%d K%d k%d s%d%s" % (d.N, d.C, d.H, d.W, d.K, d.kh, d.stride, extra)}


def bl_meta(meta):
    """a collapsed bilinear convolution whose structurally zero taps are skipped: 25 of the 36 collapsed products execute (bench.py
    prices the launch at the reference's 36 = the fine 3x3 convolution's count; ``flops`` is what the matrix cores do)"""
    m = dict(meta)
    m["flops"] = meta["flops"] * 25.0 / 36.0
    m["nominal_flops"] = meta["flops"]
    m["kernel"] = meta["kernel"] + " cls"
    return m


def sparse_meta(meta):
    """a weight gradient contracted on the sparse matrix instruction: the operand's structural zeros (three of four pixels of a
    max-pool backward) are skipped in pairs -- half the multiplications of the dense count execute"""
    m = dict(meta)
    m["flops"] = meta["flops"] / 2.0
    m["nominal_flops"] = meta["flops"]
    m["kernel"] = meta["kernel"] + " 2:4"
    return m


def pack_meta(t):
    """metadata of a separate q_pack pass (a producer without a q epilogue): fp32 read + 2-byte write per element"""
    n = float(np.prod(t.shape))
    return {"kernel": "q_pack_kernel", "dtype": 'f32', "bytes": 6.0 * n, "thin": False, "flops": 0.0,
            "geom": "N%d C%d %dx%d" % tuple(t.shape)}


def pool_sparse_meta(d, kind):
    """metadata of the d_conv1 gradients computed from the pooled operands (csrc/conv_pool_bwd.hip): algorithmic bytes =
    pooled gradient + pooled activation (fp32) + mask (1 B) per pooled element (+ the thin tensor), flops = the 25
    multiply-adds per pooled value the gather does (a quarter of the dense convolution's)"""
    pooled = float(d.N) * d.K * (d.H // 2) * (d.W // 2)
    return {"kernel": "pool_thin_wgrad_kernel" if kind == 2 else "pool_thin_dgrad_kernel", "dtype": 'f32',
            "bytes": 9.0 * pooled + 4.0 * d.N * d.C * d.H * d.W, "thin": True, "flops": 2.0 * pooled * d.kh * d.kw,
            "geom": "N%d C%d %dx%d K%d k%d s%d" % (d.N, d.C, d.H, d.W, d.K, d.kh, d.stride)}


def run_program(prog):
    for e in prog:
        e[1]()


def time_program(dev, prog, repeat=1):
    """Per-entry HIP-event timing (synchronises per entry: for profiling, not for throughput)."""
    out = []
    for e in prog:
        label, fn = e[0], e[1]
        d = e[3] if len(e) > 3 and e[3] is not None else dev      # entries on a side stream are timed there
        d.timer_start(1)
        for _ in range(repeat):
            fn()
        d.timer_stop(1)
        out.append((label, d.timer_ms(1) / repeat, e[2] if len(e) > 2 else None))
    return out

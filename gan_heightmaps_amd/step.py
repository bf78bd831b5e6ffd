"""The two-stage GAN step of the reference (/root/reference/pix2pix.py:87-147) as static device programs.

One ``GanStep`` owns the four networks' parameters in HBM and, per batch size, a set of NetPlans wired the
way ``Pix2Pix.__init__`` wires the Theano graph:
  * G(z) is evaluated once and written straight into the second half of the DCGAN discriminator's input
    batch, so D(X) and D(G(z)) run as ONE 2B-sample pass (D has no BatchNorm in any reference experiment),
  * U(X) is written straight into channels 1.. of the second half of the PatchGAN's concat buffer,
  * four gradient roots, each w.r.t. its own net's parameters, all at the pre-update parameters:
    D-loss backward on the 2B batch (weights + data gradients), G-loss backward through D on the fake half
    only (data gradients only), then through G; same for PatchGAN / U-Net with alpha * L1 added,
  * per net: (data-parallel: one RCCL all-reduce of the flat gradient buffer) then one optimiser kernel.

The DCGAN stage and the pix2pix stage of a step share nothing but the read-only input X (pix2pix.py:99: the
U-Net consumes real X, not G(z)), so they are enqueued on TWO HIP streams (two ghm contexts on the same
device): the low-parallelism layers of one stage (4x4 .. 16x16 maps, reductions) overlap the other stage's
big convolutions.
"""
import os

import numpy as np

from . import layers as L
from .device import Ops
from .engine import NetPlan, ParamStore

TRAIN_KEYS = ['dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_recon', 'p2p_disc']
LANE_OF = {'dcgan_gen': 0, 'dcgan_disc': 0, 'p2p_gen': 1, 'p2p_disc': 1}


def _has_bn(layer):
    return any(isinstance(l, L.BatchNormLayer) for l in L.get_all_layers(layer))


class _Built:
    pass


def _interleave(a, b):
    """merge two launch lists so that both streams are fed from the start (proportional round-robin)"""
    out, i, j = [], 0, 0
    while i < len(a) or j < len(b):
        if j >= len(b) or (i < len(a) and i * len(b) <= j * len(a)):
            out.append((0, a[i]))
            i += 1
        else:
            out.append((1, b[j]))
            j += 1
    return out


class GanStep:
    def __init__(self, dev, dcgan_gen, dcgan_disc, p2p_gen, p2p_disc, alpha, lsgan, reconstruction, opt_spec,
                 train_mode='both', comm=None, use_graph=True, two_streams=True, force_exchange=False,
                 side_streams=None, dtype='bf16x3', bucket_mb=None, exchange_mode=None):
        self.dev = dev
        # form of the data-parallel exchange: 'allreduce' (every rank sums every gradient and runs the whole
        # optimiser), 'allreduce_bf16' (the same program with every gradient sub-bucket rounded to bf16 for the trip: half the
        # bytes on the xGMI links, a REDUCED-PRECISION exchange, opt-in; for the first multi-GPU node to A/B) or 'rs_ag'
        # (sharded update: a sub-bucket is reduce-SCATTERED, each rank runs RMSprop / Adam on its 1 / world
        # slice of parameters + state, and the updated slices are all-gathered -- the same bytes on the links, 1 / world of
        # the optimiser's 1.13 GB / step per rank)
        self.exchange_mode = exchange_mode or os.environ.get('GHM_EXCHANGE', 'allreduce')
        assert self.exchange_mode in ('allreduce', 'allreduce_bf16', 'rs_ag'), self.exchange_mode
        # data-parallel exchange: a net's gradient bucket travels as sub-buckets of at least this many bytes, each
        # all-reduced as soon as the backward pass has completed it (_build: bucketer)
        self.bucket_bytes = int(float(bucket_mb if bucket_mb is not None else os.environ.get('GHM_BUCKET_MB', 32)) * 2 ** 20)
        # arithmetic of the convolution products (include/ghm.h GHM_DTYPE_*): 'bf16x3' (default) = the reference's floatX by
        # operand splitting (three exact bf16 pieces per operand, csrc/conv_split.hip: fp32-accurate); 'f32' = the same on the
        # fp32 matrix instruction; 'bf16' / 'f16' = BASELINE configs 4 / 5 (matrix-core operands rounded, fp32 accumulation,
        # fp32 tensors and optimiser)
        self.dtype = dtype
        # fp16 operands underflow below 6e-8 and the per-pixel gradients of the 512x512 layers sit around 1e-6..1e-9:
        # the loss-gradient seeds are scaled (initially by 2^15) and the optimiser divides the scale out again (every
        # gradient kernel is linear in its seed; gradients in HBM are fp32, so the scale costs nothing).  The DCGAN
        # discriminator's output is linear and unbounded, so a fixed scale can push a gradient operand beyond the fp16
        # range (65504 -> inf -> nan in the fp32 sums): the scale is DYNAMIC, per stage (the two stages are independent
        # loss graphs) and entirely device state -- ghm_grad_check flags a non-finite gradient bucket, the optimiser
        # kernels skip the update of a flagged step, ghm_loss_scale_update halves / regrows the scale (include/ghm.h) --
        # so a recorded / captured step adapts without a host round trip.  bf16 has fp32's range: no scale.
        self.init_loss_scale = 32768.0 if dtype == 'f16' else 1.0
        self.ls_growth_interval, self.ls_min, self.ls_max = 2000, 1.0, 2.0 ** 24
        self._ls_state = []                 # [(Device, DevTensor of 8 floats)] one per stage stream
        # how a step is issued: False = eager (one C call per kernel), True = one captured HIP graph per stage stream,
        # 'recorded' = the eager multi-stream launch sequence recorded once in the library and replayed by ONE
        # ghm_step_run call per step (side streams, communication stream and collectives included)
        assert use_graph in (True, False, 'recorded')
        if side_streams is None:            # forked branches replay slowly inside a HIP graph: not in graph mode
            side_streams = (use_graph is not True) and two_streams
        if side_streams and use_graph is True:
            raise ValueError("side_streams needs use_graph=False or 'recorded'")
        mk = type(dev)                      # second / side streams are further contexts of the same kind on this GPU
        mkops = getattr(dev, 'ops_class', Ops)
        self.devs = [dev, mk(dev.index) if two_streams else dev]
        self.ops = [mkops(self.devs[0]), mkops(self.devs[1])]
        if dtype == 'f16':
            for d in ([self.devs[0]] if self.devs[1] is self.devs[0] else self.devs):
                t = d.tensor(np.array([self.init_loss_scale, 1.0 / self.init_loss_scale, 0, 0, 0, 0, 0, 0], np.float32))
                d.set_loss_scale_state(t)
                self._ls_state.append((d, t))

        # optional GRADIENT stream for the weight / bias gradients of both stages (engine.NetPlan side=).  ONE stream for
        # the two stages, not one each: three MFMA-heavy kernels at a time (stage A, stage B, one weight gradient) is what
        # the chip runs best -- with a gradient stream per stage the two weight-gradient kernels share CUs with each other
        # and the step is 3 % slower (162.5 vs 167.6 img/s fp32, 435 vs 465 bf16).  (A per-stage pair used to measure
        # the same as the shared stream only because ROCm's default of four hardware queues happened to put the two
        # gradient streams on one queue; GHM_GRAD_STREAM_PER_STAGE=1 restores the pair for measurements.)
        self.side = [None, None]
        if side_streams:
            per_stage = two_streams and bool(os.environ.get('GHM_GRAD_STREAM_PER_STAGE'))
            sd = [mk(dev.index), mk(dev.index) if per_stage else None]
            if sd[1] is None:
                sd[1] = sd[0]
            self.side = [(sd[0], mkops(sd[0])), (sd[1], mkops(sd[1]))]
        self.nets = {'dcgan_gen': dcgan_gen, 'dcgan_disc': dcgan_disc, 'p2p_gen': p2p_gen,
                     'p2p_disc': p2p_disc["out"]}
        self.p2p_disc_inputs = p2p_disc["inputs"]
        self.alpha, self.lsgan, self.reconstruction = float(alpha), bool(lsgan), reconstruction
        self.opt_spec, self.train_mode = opt_spec, train_mode
        self.comm = comm
        self.world = comm.world if comm is not None else 1
        self.rank = comm.rank if comm is not None else 0
        # the communicator's context is the COMMUNICATION stream: a Comm made on its own Device of the same GPU lets
        # the bucket all-reduces run beside the rest of the backward pass; a Comm made on ``dev`` itself serialises
        # them on stream A.  Either way it must be this GPU, or every rank would reduce somebody else's buffers.
        self.cdev = comm.dev if comm is not None else None
        self.cops = getattr(self.cdev, 'ops_class', Ops)(self.cdev) if comm is not None else None
        if comm is not None and getattr(comm.dev, 'index', None) != getattr(dev, 'index', None):
            raise ValueError("comm was initialised on device %r, the step runs on device %r"
                             % (getattr(comm.dev, 'index', None), getattr(dev, 'index', None)))
        # data-parallel code path (stream hand-over, RCCL all-reduce, updates on stream A) even with one rank:
        # lets a single-GPU box exercise exactly what N ranks run
        self.exchange = self.world > 1 or (force_exchange and comm is not None)
        self.use_graph = use_graph
        self.sharded = self.exchange and self.exchange_mode == 'rs_ag'
        if self.sharded and dtype == 'f16':
            raise NotImplementedError("exchange_mode='rs_ag' with the fp16 dynamic loss scale: every rank would check only its "
                                      "own gradient shard for overflow; use bf16 (no scale) or the all-reduce form")
        self.shard_unit = 64 * self.world if self.sharded else 1       # elements: world shards of whole 256-byte lines
        self.stores = {k: ParamStore(self.devs[LANE_OF[k]], L.get_all_params(v), pad_to=self.shard_unit)
                       for k, v in self.nets.items()}
        # per-net optimiser state + hyper-parameter scalars [lr, t] in HBM
        self.hyper = {}
        lr = float(opt_spec.learning_rate.get_value()) if hasattr(opt_spec.learning_rate, 'get_value') \
            else float(opt_spec.learning_rate)
        for k, st in self.stores.items():
            d = self.devs[LANE_OF[k]]
            self.hyper[k] = d.tensor(np.array([lr, 0.0], np.float32))
            n = st.n_pad
            if opt_spec.kind == 'rmsprop':
                st.opt_state = {'acc': d.zeros((1, n, 1, 1))}
            elif opt_spec.kind == 'adam':
                st.opt_state = {'m': d.zeros((1, n, 1, 1)), 'v': d.zeros((1, n, 1, 1))}
            else:
                raise ValueError(opt_spec.kind)
        if hasattr(opt_spec.learning_rate, '_listeners'):
            opt_spec.learning_rate._listeners.append(self.set_lr)
        self.losses_dev = dev.zeros((1, 8, 1, 1))
        self._built = {}
        self._infer = {}

    def loss_scale_state(self):
        """[{scale, clean_steps, skipped_steps}] per stage stream (fp16 mode; [] otherwise).  Synchronises."""
        self.sync()
        out = []
        for _, t in self._ls_state:
            v = t.numpy().ravel()
            out.append({'scale': float(v[0]), 'clean_steps': int(v[2]), 'skipped_steps': int(v[4])})
        return out

    def restore_loss_scale_state(self, states):
        """checkpointed [{scale, clean_steps, skipped_steps}] per stage stream back into the device records"""
        self.sync()
        for (d, t), st in zip(self._ls_state, states):
            v = t.numpy().ravel()
            v[0], v[1], v[2], v[3], v[4] = st['scale'], 1.0 / st['scale'], st.get('clean_steps', 0), 0, st.get('skipped_steps', 0)
            t.set(v)

    def set_loss_scale(self, scale):
        self.sync()
        for d, t in self._ls_state:
            v = t.numpy().ravel()
            v[0], v[1], v[2], v[3] = scale, 1.0 / scale, 0, 0
            t.set(v)

    @property
    def loss_scale(self):
        """the scale the gradient buffers currently carry (stage A's; the stages only differ after an overflow)"""
        if not self._ls_state:
            return 1.0
        self.sync()
        return float(self._ls_state[0][1].numpy().ravel()[0])

    def sync(self):
        self.devs[0].sync()
        if self.devs[1] is not self.devs[0]:
            self.devs[1].sync()
        if self.cdev is not None and self.cdev is not self.devs[0]:
            self.cdev.sync()
        if hasattr(self, '_pipe_state'):            # the copy stream of the input pipeline: no upload may outlive a sync()
            self._pipe_state['dev'].sync()

    def broadcast_parameters(self, root=0):
        """Make every replica start from rank ``root``'s parameters, BatchNorm state and optimiser state (the
        reference never seeds lasagne's RNG -- SURVEY Appendix A.9 -- so unseeded ranks would otherwise train
        different weights with averaged gradients).  A broadcast is an all-reduce whose other contributions are
        zero: exact, and it needs no further collective in the C ABI."""
        if self.comm is None or self.world == 1:
            return
        self.sync()
        for k in ('dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc'):
            st = self.stores[k]
            bufs = [(st.w, st.n_train), (st.s, st.n_state)] + [(t, st.n_train) for _, t in sorted(st.opt_state.items())]
            bufs.append((self.hyper[k], 2))
            for t, n in bufs:
                if n <= 0:
                    continue
                if self.rank != root:
                    self.cdev.memset_zero(t.ptr, 4 * n)
                self.cops.allreduce_sum(t, n)
        self.sync()

    def replica_checksums(self):
        """(min, max) over the ranks of a CRC of every parameter / state buffer: equal on healthy replicas."""
        import zlib
        self.sync()
        crc = 0
        for k in ('dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc'):
            st = self.stores[k]
            for t in (st.w, st.s):
                crc = zlib.crc32(t.numpy().tobytes(), crc)
        if self.comm is None or self.world == 1:
            return crc, crc
        lo16, hi16 = float(crc & 0xffff), float(crc >> 16)          # exactly representable in fp32
        mx = (self.comm.max_scalar(hi16), self.comm.max_scalar(lo16))
        mn = (-self.comm.max_scalar(-hi16), -self.comm.max_scalar(-lo16))
        return (int(mn[0]) << 16 | int(mn[1])), (int(mx[0]) << 16 | int(mx[1]))

    def set_lr(self, lr):
        self.sync()
        for k, h in self.hyper.items():
            cur = h.numpy().ravel()
            cur[0] = lr
            h.set(cur)

    # ---- building -------------------------------------------------------------------------------------
    def _build(self, B, slot=0):
        b = _Built()
        # dropout step counters live per (net, batch size) on the engine, whichever slot of that batch size is built first: both
        # slots of a batch size advance ONE counter, so the pipelined loop draws the masks of the sequential loop (a ragged last
        # batch first seen on an odd step builds slot 1 before slot 0)
        if not hasattr(self, '_rng_counters'):
            self._rng_counters = {}
        rc = self._rng_counters
        dA, dB = self.devs
        oA, oB = self.ops
        G, D, U, P = (self.nets[k] for k in ('dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc'))
        d_in_layer = [l for l in L.get_all_layers(D) if isinstance(l, L.InputLayer)][0]
        u_in_layer = [l for l in L.get_all_layers(U) if isinstance(l, L.InputLayer)][0]
        i_a, i_b = self.p2p_disc_inputs
        ca, H, W = d_in_layer.shape[1:]
        b.d_in = dA.empty((2 * B, ca, H, W))
        # which nets fork their weight / bias gradients onto the gradient stream: all but the DCGAN generator, whose
        # small weight gradients stay inline on stage A (measured img/s, fp32 / bf16 / fp32 batch 2 / 1024^2 fp16:
        # DPU 169.4 / 492.3 / 153.0 / 98.9, PU 169.3 / 498.8 / 150.5 / 97.2, GDPU 167.7 / 471.9 / 150.4 / 97.2,
        # none 164.5 / 467.1, GD 164.0).  GHM_SIDE_NETS overrides (tuning).
        # (reduced precision: +1 % with D inline too; the split-fp32 mode, whose kernels are as long as the fp32 ones: 253 -> 261 img/s with D on the side)
        _sn = os.environ.get('GHM_SIDE_NETS', 'DPU' if self.dtype in ('f32', 'bf16x3', 'bf16x2') else 'PU')
        _side = lambda k, lane: self.side[lane] if k in _sn else None
        b.G = NetPlan(dA, oA, G, B, self.stores['dcgan_gen'], out_tensor=b.d_in.samples(B, 2 * B), name="G",
                      side=_side('G', 0), rng_seed=self.rank, dtype=self.dtype,       # replicas draw different dropout masks
                      rng_counter=rc.get(('G', B)))
        b.D = NetPlan(dA, oA, D, 2 * B, self.stores['dcgan_disc'], inputs={d_in_layer: b.d_in}, name="D",
                      side=_side('D', 0), bn_groups=2 if _has_bn(D) else 1, dtype=self.dtype)
        b.P = NetPlan(dB, oB, P, 2 * B, self.stores['p2p_disc'], name="P", side=_side('P', 1),
                      bn_groups=2 if _has_bn(P) else 1, dtype=self.dtype)
        pa, pb = b.P.input_tensor(i_a), b.P.input_tensor(i_b)
        b.U = NetPlan(dB, oB, U, B, self.stores['p2p_gen'], out_tensor=pb.samples(B, 2 * B), name="U",
                      side=_side('U', 1), rng_seed=self.rank, dtype=self.dtype,
                      rng_counter=rc.get(('U', B)))
        rc.setdefault(('G', B), b.G.rng_counter)
        rc.setdefault(('U', B), b.U.rng_counter)
        b.z = b.G.input_nodes[0].out
        b.x = b.U.input_tensor(u_in_layer)
        b.y = dB.empty((B,) + tuple(pb.shape[1:]))
        b.B = B
        lo = self.losses_dev
        slot = lambda i: lo.channels(i, i + 1)
        advA = oA.lsgan_loss if self.lsgan else oA.bce_loss
        advB = oB.lsgan_loss if self.lsgan else oB.bce_loss
        l2 = self.reconstruction == 'l2'

        # ---- shared forward (pix2pix.py:92-101), one list per stream ----
        # sharded update: a net's parameters are all-gathered at the END of a step in forward order (G, U, D, P; within a net the
        # first layers first) and the NEXT step's forward waits per net, right in front of the net's first weight read -- the
        # discriminators' gathers run under the generators' forward passes instead of in front of the whole step
        # (captured HIP graphs, use_graph=True: an event wait cannot sit inside a capture -- ghm_event_wait refuses -- so that
        # form keeps one wait for the communication stream at the end of the step, as the all-reduce form does)
        per_net_waits = self.sharded and self.use_graph is not True
        gev = self._gather_events() if per_net_waits else {}

        def gwait(prog, dev_, k):
            if k in gev:
                prog.append(("wait_gather_" + k, lambda dev_=dev_, ev=gev[k]: dev_.event_wait(ev), None, dev_))

        fa = [("x_to_d_in", lambda: oA.copy_view(b.x, b.d_in.samples(0, B)))]
        fb = []
        if per_net_waits:
            # nothing waits for the communication stream at the end of a sharded step, and a stage whose nets were not
            # exchanged (train_mode 'dcgan' / 'p2p') never meets a gather event: its loss kernels of the NEXT step write
            # losses_dev, which the previous step's loss all-reduce may still be reading -- both stage streams wait for the
            # event recorded behind that all-reduce (unrecorded on the first step: the wait is a no-op)
            lev = self._losses_event()
            fa.append(("wait_losses_reduced", lambda: dA.event_wait(lev), None, dA))
            if dB is not dA:
                fb.append(("wait_losses_reduced", lambda: dB.event_wait(lev), None, dB))
        gwait(fa, dA, 'dcgan_gen')
        b.G.emit_forward(fa)
        gwait(fa, dA, 'dcgan_disc')
        b.D.emit_forward(fa)
        fb += [("x_to_p_in0", lambda: oB.copy_view(b.x, pa.samples(0, B))),
               ("x_to_p_in1", lambda: oB.copy_view(b.x, pa.samples(B, 2 * B))),
               ("y_to_p_in", lambda: oB.copy_view(b.y, pb.samples(0, B)))]
        gwait(fb, dB, 'p2p_gen')
        b.U.emit_forward(fb)
        gwait(fb, dB, 'p2p_disc')
        b.P.emit_forward(fb)
        d_out, p_out = b.D.out, b.P.out
        d_real, d_fake = d_out.samples(0, B), d_out.samples(B, 2 * B)
        p_real, p_fake = p_out.samples(0, B), p_out.samples(B, 2 * B)
        b.seed_D, b.seed_G = dA.empty(d_out.shape), dA.empty(d_fake.shape)
        b.seed_PD, b.seed_PG = dB.empty(p_out.shape), dB.empty(p_fake.shape)

        LS = 1.0            # the fp16 loss scale is device state read by the loss kernels (ghm_set_loss_scale_state)

        def losses_a(prog, g):
            # (:107) gen_loss_dcgan, (:108) disc_loss_dcgan
            prog.append(("loss", lambda: advA(d_fake, 1.0, slot(0), b.seed_G if g else None, LS)))
            prog.append(("loss", lambda: advA(d_real, 1.0, slot(1), b.seed_D.samples(0, B) if g else None, LS)))
            prog.append(("loss", lambda: advA(d_fake, 0.0, slot(1), b.seed_D.samples(B, 2 * B) if g else None,
                                              LS, True)))

        def losses_b(prog, g):
            # (:110) gen_loss_p2p, (:121) disc_loss_p2p
            prog.append(("loss", lambda: advB(p_fake, 1.0, slot(2), b.seed_PG if g else None, LS)))
            prog.append(("loss", lambda: advB(p_real, 1.0, slot(4), b.seed_PD.samples(0, B) if g else None, LS)))
            prog.append(("loss", lambda: advB(p_fake, 0.0, slot(4), b.seed_PD.samples(B, 2 * B) if g else None,
                                              LS, True)))

        # ---- loss_fn (:143): forward + losses, BN running stats still update ----
        la, lb = list(fa), list(fb)
        losses_a(la, False)
        losses_b(lb, False)
        lb.append(("recon", lambda: oB.recon_loss(b.U.out, b.y, slot(3), None, 1.0, l2)))
        b.loss_prog = [la, lb]

        # ---- train_fn (:142) ----
        ta, tb = list(fa), list(fb)
        losses_a(ta, True)
        losses_b(tb, True)
        do_dcgan = self.train_mode in ('both', 'dcgan')
        do_p2p = self.train_mode in ('both', 'p2p')
        tdone = set()      # conv weights whose transposed copy is already fresh in this program
        # ---- data-parallel exchange (no reference counterpart) ----
        # One all-reduce per net bucket on the COMMUNICATION stream (the communicator's context), enqueued where the
        # bucket's last gradient kernel has been issued: the discriminator buckets reduce under the generator's
        # backward pass, the DCGAN buckets under the pix2pix stage.  The communication stream waits for the streams
        # that wrote the bucket (events recorded at this point of the program), the collectives run in host-enqueue
        # order, and that order is a pure function of the program -- identical on every rank.
        cdev, cops = self.cdev, self.cops
        embed = self.exchange and self.use_graph is not True  # RCCL calls stay outside captured HIP graphs

        b.xchg_order = []           # [(label, net key, first element, n elements)] in collective order

        def bucketer(k, lane, prog):
            """-> (on_grads callback for emit_backward, flush): the gradient bucket of net ``k`` as SUB-BUCKETS, contiguous
            ranges of the flat gradient buffer of >= self.bucket_bytes each, cut in the order the backward pass completes
            them (last layers first = highest offsets first).  A sub-bucket's all-reduce is enqueued right after the
            launch that completes its last gradient, so only the final one (the first layers' few parameters) is
            exposed behind the stage's last kernel; the rest travels under the remaining backward pass."""
            st = self.stores[k]
            srcs = [self.devs[lane]] + ([self.side[lane][0]] if self.side[lane] is not None else [])
            tr = sorted((p for p in st.params if p.index[0] == 'w'), key=lambda p: p.index[1])
            offs = [p.index[1] for p in tr] + [st.n_train]
            unit = self.shard_unit
            buckets, hi, pend = [], (st.n_pad if self.sharded else st.n_train), []
            for i in range(len(tr) - 1, -1, -1):
                pend.append(tr[i])
                # sharded form: a bucket starts on a multiple of world x 64 elements (world equal, line-aligned shards); the
                # parameter that straddles the cut belongs to BOTH neighbours' pending sets (it completes last anyway)
                lo = offs[i] // unit * unit if i else 0
                if (4 * (hi - lo) >= self.bucket_bytes and lo < hi) or i == 0:
                    straddle = [q for q in tr[:i] if q.index[1] + int(np.prod(q.shape)) > lo] if lo < offs[i] else []
                    if hi > lo:
                        buckets.append({'lo': lo, 'hi': hi, 'pending': {id(p) for p in pend + straddle}, 'sent': False})
                    hi, pend = lo, list(straddle)
            of = {}
            for bk in buckets:
                for pid in bk['pending']:
                    of.setdefault(pid, []).append(bk)
            b.net_buckets = getattr(b, 'net_buckets', {})
            b.net_buckets[k] = buckets

            half = self.exchange_mode == 'allreduce_bf16'
            if half and not hasattr(st, 'xchg_bf16'):
                st.xchg_bf16 = cdev.alloc(2 * st.n_pad + 256)          # the net's bf16 exchange buffer (one halfword per gradient)

            def send(bk):
                bk['sent'] = True
                lo, n = bk['lo'], bk['hi'] - bk['lo']
                view = st.g.channels(lo, bk['hi'])
                sharded = self.sharded
                label = ("reducescatter_" if sharded else "allreduce_") + \
                    ("%s_%d" % (k, buckets.index(bk)) if len(buckets) > 1 else k)

                def fn():
                    for d in srcs:
                        cdev.wait_for(d)
                    if sharded:
                        cops.reduce_scatter_sum(view, n // self.world)
                    elif half:
                        cops.allreduce_sum_bf16(view, n, st.xchg_bf16 + 2 * lo)
                    else:
                        cops.allreduce_sum(view, n)
                e = (label, fn, None, cdev)
                b.xchg_order.append((label, k, lo, n))
                (prog if embed else b.exchange_late).append(e)

            def on_grads(_prog, params):
                for p in params:
                    for bk in of.get(id(p), ()):
                        bk['pending'].discard(id(p))
                        if not bk['pending'] and not bk['sent']:
                            send(bk)

            def flush():                # parameters no launch reported (none today): their bucket still travels
                for bk in buckets:
                    if not bk['sent']:
                        send(bk)
            return on_grads, flush

        b.exchange_late = []        # graph mode: the collectives stay outside the captured graphs, after both programs
        nohook = (None, lambda: None)
        if do_dcgan:
            b.D.emit_transposes(ta, tdone)
            b.G.emit_transposes(ta, tdone)
            hook, flush = bucketer('dcgan_disc', 0, ta) if self.exchange else nohook
            n1 = self._per_sample_scalar_head(b.D, d_in_layer, self.dtype)
            if n1 is not None:
                # the fake half of the discriminator-loss seed, kept aside (a backward pass may modify its seed in place)
                b.seed_Df = dA.empty(d_fake.shape)
                ta.append(("seed_copy", lambda: oA.copy_view(b.seed_D.samples(B, 2 * B), b.seed_Df)))
            b.D.emit_backward(ta, b.seed_D, wgrad=True, tag="dloss", transposed=tdone, on_grads=hook)
            flush()
            if n1 is not None:
                # D returns ONE scalar per sample and no layer couples samples: its backward pass on sample n is linear in
                # the single number dLoss/dD(G(z))_n, so the generator-loss gradient at every depth is the discriminator-loss
                # gradient of the fake half times seed_G[n] / seed_D[n].  The dloss pass above already walked the fake half
                # down to the first layer's output; only that layer's data gradient is left, then one per-sample factor
                # (:107-108: both losses read the same D(G(z)); 12 -> 8 image-backward passes through D per step).
                g1 = b.D.grads_of(n1).samples(B, 2 * B)
                gin = b.D.emit_backward(ta, None, nslice=(B, 2 * B), wgrad=False, input_grads=[d_in_layer],
                                        tag="gloss", transposed=tdone, resume={n1: g1})
                gfake = gin[d_in_layer]
                ta.append(("per_sample_ratio", lambda: oA.scale_samples(gfake, b.seed_G, b.seed_Df)))
            else:
                gin = b.D.emit_backward(ta, b.seed_G, nslice=(B, 2 * B), wgrad=False, input_grads=[d_in_layer],
                                        tag="gloss", transposed=tdone)
            hook, flush = bucketer('dcgan_gen', 0, ta) if self.exchange else nohook
            b.G.emit_backward(ta, gin[d_in_layer], wgrad=True, transposed=tdone, on_grads=hook)
            flush()
        if do_p2p:
            b.P.emit_transposes(tb, tdone)
            b.U.emit_transposes(tb, tdone)
            hook, flush = bucketer('p2p_disc', 1, tb) if self.exchange else nohook
            b.P.emit_backward(tb, b.seed_PD, wgrad=True, tag="dloss", transposed=tdone, on_grads=hook)
            flush()
            gin = b.P.emit_backward(tb, b.seed_PG, nslice=(B, 2 * B), wgrad=False, input_grads=[i_b], tag="gloss",
                                    transposed=tdone)
            gu = gin[i_b]
            # (:115-117) recon loss and alpha * d recon / d U(X) added to the adversarial gradient
            tb.append(("recon", lambda: oB.recon_loss(b.U.out, b.y, slot(3), gu, self.alpha * LS, l2, True)))
            hook, flush = bucketer('p2p_gen', 1, tb) if self.exchange else nohook
            b.U.emit_backward(tb, gu, wgrad=True, transposed=tdone, on_grads=hook)
            flush()
        else:
            tb.append(("recon", lambda: oB.recon_loss(b.U.out, b.y, slot(3), None, 1.0, l2)))
        if self.side[0] is not None:        # the gradient streams rejoin before anything consumes the gradients
            ta.append(("join", lambda: dA.wait_for(self.side[0][0])))
            tb.append(("join", lambda: dB.wait_for(self.side[1][0])))
        b.train_compute = [ta, tb]
        # ---- after both stage programs: (graph mode: the bucket all-reduces, in a fixed order), the losses, then
        # the stage streams wait for the communication stream and apply their own nets' updates (:131-141) ----
        keys = (['dcgan_gen', 'dcgan_disc'] if do_dcgan else []) + (['p2p_gen', 'p2p_disc'] if do_p2p else [])
        b.exchange = []
        if self.exchange:
            b.exchange.extend(b.exchange_late)          # graph mode: every sub-bucket, in completion order per stage

            def reduce_losses():
                cdev.wait_for(dA)
                if dB is not dA:
                    cdev.wait_for(dB)
                cops.allreduce_sum(lo, 8)
            b.exchange.append(("allreduce_losses", reduce_losses, None, cdev))
            if per_net_waits:
                b.exchange.append(("losses_reduced", lambda ev=self._losses_event(): cdev.event_record(ev), None, cdev))
            if self.sharded:
                # (the communication stream has just waited for both stage streams: every kernel that reads the pre-update
                # weights is behind it.)  Per sub-bucket, in the order it was reduced: this rank's shard of the optimiser
                # update, then the all-gather of the updated parameter shards.
                gs_, hp_ = 1.0 / self.world, self.opt_spec.hp
                # forward order: the nets as the next step reads them (the generators of both stages first), a net's
                # sub-buckets by ascending offset = first layers first; a per-net event behind its last gather
                fwd_rank = {'dcgan_gen': 0, 'p2p_gen': 1, 'dcgan_disc': 2, 'p2p_disc': 3}
                order = sorted(b.xchg_order, key=lambda t: (fwd_rank[t[1]], t[2]))
                last_of = {t[1]: i for i, t in enumerate(order)}
                for idx, (label, k, blo, n) in enumerate(order):
                    st, hy, sh = self.stores[k], self.hyper[k], n // self.world
                    a0 = blo + self.rank * sh
                    wv, gv = st.w.channels(a0, a0 + sh), st.g.channels(a0, a0 + sh)
                    if self.opt_spec.kind == 'rmsprop':
                        av = st.opt_state['acc'].channels(a0, a0 + sh)
                        b.exchange.append(("rmsprop_shard_" + label, lambda wv=wv, gv=gv, av=av, hy=hy, sh=sh: cops.rmsprop(
                            wv, gv, av, sh, hy, hp_['rho'], hp_['epsilon'], gs_), None, cdev))
                    else:
                        mv, vv = st.opt_state['m'].channels(a0, a0 + sh), st.opt_state['v'].channels(a0, a0 + sh)
                        b.exchange.append(("adam_shard_" + label, lambda wv=wv, gv=gv, mv=mv, vv=vv, hy=hy, sh=sh: cops.adam(
                            wv, gv, mv, vv, sh, hy, hp_['beta1'], hp_['beta2'], hp_['epsilon'], gs_), None, cdev))
                    full = st.w.channels(blo, blo + n)
                    b.exchange.append(("allgather_" + label[len("reducescatter_"):], lambda full=full, sh=sh: cops.all_gather(full, sh),
                                       None, cdev))
                    if last_of[k] == idx and k in gev:
                        b.exchange.append(("gathered_" + k, lambda ev=gev[k]: cdev.event_record(ev), None, cdev))
                if self.opt_spec.kind == 'adam':
                    for k in keys:
                        b.exchange.append(("adam_tick_" + k, lambda hy=self.hyper[k]: cops.adam_tick(hy), None, cdev))

            # one entry per stage stream, so that bench.py can bracket each with HIP events: the time a stage stream
            # spends in this wait is the EXPOSED part of the exchange
            # (sharded form: no wait here -- the next forward waits per net, ``wait_gather_*`` above; except under captured
            # HIP graphs, where the waits cannot sit inside the graphs)
            if not per_net_waits:
                b.exchange.append(("wait_comm", lambda: dA.wait_for(cdev), None, dA))
                if dB is not dA:
                    b.exchange.append(("wait_comm", lambda: dB.wait_for(cdev), None, dB))
        gs = 1.0 / self.world          # (x 1 / loss scale inside the optimiser kernels, from the device state)
        hp = self.opt_spec.hp
        b.update = [[], []]
        # one stream for both stages: one update list, so the fp16 sequence check* -> update* -> scale update is kept
        ulane = (lambda k: 0) if (self._ls_state and self.devs[1] is self.devs[0]) else (lambda k: LANE_OF[k])
        if self._ls_state:
            # fp16: every gradient bucket of the stage is checked (after its all-reduce: all ranks see the same sum, so
            # they skip or apply together) before the first update of the stage reads the flag
            for k in keys:
                st, lane = self.stores[k], ulane(k)
                b.update[lane].append(("grad_check_" + k, lambda st=st, o=self.ops[lane]: o.grad_check(st.g, st.n_train)))
        for k in ([] if self.sharded else keys):       # (sharded form: the updates ran on the communication stream, above)
            st, hy = self.stores[k], self.hyper[k]
            lane = ulane(k)
            o = self.ops[lane]
            if self.opt_spec.kind == 'rmsprop':
                b.update[lane].append(("rmsprop_" + k, lambda st=st, hy=hy, o=o: o.rmsprop(
                    st.w, st.g, st.opt_state['acc'], st.n_train, hy, hp['rho'], hp['epsilon'], gs)))
            else:
                b.update[lane].append(("adam_" + k, lambda st=st, hy=hy, o=o: o.adam(
                    st.w, st.g, st.opt_state['m'], st.opt_state['v'], st.n_train, hy, hp['beta1'], hp['beta2'],
                    hp['epsilon'], gs)))
                b.update[lane].append(("adam_tick_" + k, lambda hy=hy, o=o: o.adam_tick(hy)))
        if self._ls_state:
            for lane in (0, 1):
                if b.update[lane]:
                    b.update[lane].append(("loss_scale_update", lambda o=self.ops[lane]: o.loss_scale_update(
                        self.ls_growth_interval, self.ls_min, self.ls_max)))
        b.graphs = {}
        b.calls = {}
        return b

    @staticmethod
    def _per_sample_scalar_head(plan, in_layer, dtype='bf16x3'):
        """-> the node that reads ``in_layer`` if the generator-loss gradient through this discriminator may be taken from its
        discriminator-loss pass (see _build), else None: one scalar per sample out, no BatchNorm / InstanceNorm node (batch
        statistics couple the samples; the normalisation backward is not sliced), one conv reader of the input.
        GHM_NO_RANK_ONE=1 keeps the two separate passes (the A/B switch of tests/test_gpu_step.py).
        Not in 'f16': the identity is exact, but the shared pass carries the fake half at the DISCRIMINATOR-loss seed, which is
        smaller than the generator-loss seed by d / (1 - d) (LSGAN; p / (1 - p) with BCE) -- 1e-2 .. 1e-5 once the discriminator
        is winning (pix2pix.py:107-108; results.txt of the reference's run: dcgan_disc 0.0119) -- and fp16 gradient operands
        (range 6e-8 .. 65504 behind the 2^15 loss scale) flush per-pixel gradients that small to zero before the per-sample
        factor multiplies them back up.  fp32 / bf16 pieces have fp32's exponent range: there the error stays relative
        (tests/test_gpu_step.py::test_generator_gradient_shortcut_with_a_confident_discriminator)."""
        if os.environ.get('GHM_NO_RANK_ONE') or (dtype == 'f16' and not os.environ.get('GHM_RANK_ONE_F16')):      # (GHM_RANK_ONE_F16=1: measurement only)
            return None
        if int(np.prod(plan.out.shape[1:])) != 1 or any(n.op == 'bn' for n in plan.order):
            return None
        node = plan.node_of_layer[id(in_layer)]
        if len(node.consumers) != 1 or node.consumers[0].op not in ('conv', 'convpool'):
            return None
        return node.consumers[0]

    def built(self, B, slot=0):
        """the plan set of batch size B; slot 1 = a second, independent set (own activations and input buffers, the same
        parameter stores) for the input pipeline's double buffering"""
        key = B if slot == 0 else (B, slot)
        if key not in self._built:
            self._built[key] = self._build(B, slot)
        return self._built[key]

    # ---- running --------------------------------------------------------------------------------------
    def _upload(self, b, Z, X, Y):
        self.sync()                 # the previous step may still be reading the input buffers
        b.z.set(Z)
        b.x.set(X)
        b.y.set(Y)
        self.sync()

    # ---- asynchronous input pipeline (no reference counterpart: pix2pix.py:201-212 uploads, waits, steps, waits) ---------
    # The input buffers of a plan are live until almost the end of its step (the first layers' weight gradients read z and
    # x last), and consecutive steps OVERLAP in the steady state (stage A of step i+1 starts while the gradient stream still
    # finishes step i): any hand-over that waits for "the previous step" is a barrier that costs more than the upload it
    # hides (measured: 10.2 ms per bf16 step with a staging set + device-to-device hand-over, against 6.2 resident).  So the
    # pipeline double-buffers the PLAN: two complete plans per batch size (slot 0 / 1: own activations and inputs, shared
    # parameter stores; +11 GB of 288), consecutive steps alternate, and the batch of step i+1 goes from page-locked host
    # staging straight into the other slot's input buffers on a COPY stream while step i runs.  Orderings, all by events:
    #   * the stage streams of step i wait for the event recorded right behind the upload of batch i (long passed, normally);
    #   * before the HOST starts an upload into a slot it waits (hipEventSynchronize) for the events recorded behind the step
    #     that last used that slot -- a host-side wait, two steps back, never a device-side one: a copy stream that sits in a
    #     hipStreamWaitEvent for the end of a step blocks its hardware queue, and whichever compute stream ROCm mapped onto
    #     the same queue stalls with it (measured: 6.2 -> 8.8 .. 9.9 ms per bf16 step depending on which stream it hit);
    #   * the host reuses a page-locked set only after its upload has passed (event sync).
    # Results are bit-identical to the synchronous loop (the two slots run the same program on the same parameters).
    def _pipe(self):
        if not hasattr(self, '_pipe_state'):
            mk = type(self.devs[0])
            # the copy stream must not share a HARDWARE queue with a compute stream (a 16 MB upload in flight on a shared queue
            # holds that stream's kernels back for its whole duration: bf16 611 instead of 638 img/s on the boxes where ROCm
            # happened to map them together): probe candidates, keep the first one every compute stream is free of
            compute = [d for d in self._all_devs() if d is not None]
            cp, rejects, report = None, [], []
            if hasattr(mk, 'queue_interference') and os.environ.get("GHM_NO_QUEUE_PROBE") is None:
                self.sync()
                probe = mk(self.devs[0].index)
                for _ in range(6):
                    cand = mk(self.devs[0].index)
                    worst = max(max(cand.queue_interference(d, probe), d.queue_interference(cand, probe)) for d in compute)
                    report.append(round(worst, 1))
                    if worst < 300.0:               # (the spin is 1500 us; an unshared queue answers in tens of microseconds)
                        cp = cand
                        break
                    rejects.append(cand)
                for r in rejects[1:] + [probe]:
                    r.close()
                if cp is None:                      # every candidate shares a queue with somebody: take the first
                    cp = rejects[0]
                elif rejects:
                    rejects[0].close()
            else:
                cp = mk(self.devs[0].index)
            self._pipe_state = {'dev': cp, 'slots': {}, 'queue_probe_us': report}
        return self._pipe_state

    def _pipe_slot(self, b):
        pipe = self._pipe()
        if id(b) not in pipe['slots']:
            from .device import PinnedArray
            mkpin = getattr(type(self.devs[0]), 'pinned_array', PinnedArray)       # (host-memory test devices bring their own)
            assert b.z.contiguous and b.x.contiguous and b.y.contiguous
            pipe['slots'][id(b)] = {'host': {k: mkpin(t.shape) for k, t in (('z', b.z), ('x', b.x), ('y', b.y))},
                                    'landed': pipe['dev'].event_create(), 'uploads': 0,
                                    'done': [(d, d.event_create()) for d in self._all_devs()], 'used': False}
        return pipe['slots'][id(b)]

    def upload_async(self, b, Z, X, Y):
        """start the upload of a batch into plan ``b``'s input buffers on the copy stream; returns as soon as the copies
        are enqueued (it first waits, on the host, for the step that last ran on this plan)"""
        cp = self._pipe()['dev']
        sl = self._pipe_slot(b)
        if sl['used']:
            for d, ev in sl['done']:
                d.event_sync(ev)                # the step that last read these input buffers has finished
        if sl['uploads']:
            cp.event_sync(sl['landed'])         # the previous upload from this host set has passed
        for k, a, t in (('z', Z, b.z), ('x', X, b.x), ('y', Y, b.y)):
            np.copyto(sl['host'][k].array, np.asarray(a, np.float32).reshape(sl['host'][k].shape))
            cp.h2d_async(t.ptr, sl['host'][k])
        cp.event_record(sl['landed'])
        sl['uploads'] += 1

    def upload_resident_async(self, b, zt, xt, yt):
        """upload_async for a batch that already lies in HBM (three contiguous DevTensors): device-to-device copies into plan
        ``b``'s input buffers on the copy stream, same orderings (bench.py rotates resident synthetic batches through its timed
        steps this way, so that no step re-trains the batch of the step before it)"""
        cp = self._pipe()['dev']
        sl = self._pipe_slot(b)
        if sl['used']:
            for d, ev in sl['done']:
                d.event_sync(ev)                # the step that last read these input buffers has finished
        for src, dst in ((zt, b.z), (xt, b.x), (yt, b.y)):
            assert src.contiguous and dst.contiguous and src.size == dst.size
            cp.d2d(dst.ptr, src.ptr, 4 * dst.size)
        cp.event_record(sl['landed'])

    def produce_async(self, b, it, Z_sampler):
        """like upload_async, with the (A, B) batch made on the device by a data.Hdf5Iterator: uint8 rows from page-locked
        staging + ghm_image_batch straight into plan ``b``'s inputs, all on the copy stream"""
        cp = self._pipe()['dev']
        sl = self._pipe_slot(b)
        if sl['used']:
            for d, ev in sl['done']:
                d.event_sync(ev)
        if sl['uploads']:
            cp.event_sync(sl['landed'])
        # (the iterator's device-side staging buffers are per iterator, not per slot: the copy stream orders their reuse)
        n = it.next_into(b.x, b.y, via=cp, pinned=sl.setdefault('it_pinned', {}))
        assert n == b.B
        np.copyto(sl['host']['z'].array, np.ascontiguousarray(Z_sampler(n), np.float32).reshape(sl['host']['z'].shape))
        cp.h2d_async(b.z.ptr, sl['host']['z'])
        cp.event_record(sl['landed'])
        sl['uploads'] += 1

    def train_pipelined_from_iterator(self, it, Z_sampler, steps):
        """``steps`` train steps on batches of a data.Hdf5Iterator, batch i+1 produced while step i runs; yields the losses"""
        if steps <= 0:
            return
        b = self.built(it.peek_n(), 0)
        self.produce_async(b, it, Z_sampler)
        for i in range(steps):
            self.enqueue_train_uploaded(b)
            nb = None
            if i + 1 < steps:
                nb = self.built(it.peek_n(), (i + 1) & 1)
                self.produce_async(nb, it, Z_sampler)
            yield self._read_losses()
            b = nb

    def enqueue_train_uploaded(self, b, wrap=None):
        """one train step of plan ``b`` on the batch last handed to upload_async(b, ...) (asynchronous)"""
        sl = self._pipe_slot(b)
        for d in set(self.devs):
            d.event_wait(sl['landed'])
        self.enqueue_train(b, wrap)
        for d, ev in sl['done']:
            d.event_record(ev)
        sl['used'] = True

    def train_pipelined(self, batches):
        """train_fn over an iterable of (Z, X, Y) host batches with the upload of batch i+1 under step i; yields the five
        losses of every step (what train(Z, X, Y) returns), bit-identical to calling train() batch by batch"""
        it = iter(batches)
        cur = next(it, None)
        i = 0
        if cur is None:
            return
        b = self.built(int(np.shape(cur[1])[0]), 0)
        self.upload_async(b, *cur)
        while cur is not None:
            self.enqueue_train_uploaded(b)
            nxt = next(it, None)
            nb = None
            if nxt is not None:
                nb = self.built(int(np.shape(nxt[1])[0]), (i + 1) & 1)
                self.upload_async(nb, *nxt)             # batch i+1 crosses PCIe while step i runs
            yield self._read_losses()
            cur, b, i = nxt, nb, i + 1

    def close_pipeline(self):
        if hasattr(self, '_pipe_state'):
            pipe = self._pipe_state
            pipe['dev'].sync()
            for sl in pipe['slots'].values():
                for h in list(sl['host'].values()) + list(sl.get('it_pinned', {}).values()):
                    h.close()
                pipe['dev'].event_destroy(sl['landed'])
                for d, ev in sl['done']:
                    d.event_destroy(ev)
            pipe['dev'].close()
            del self._pipe_state

    def _losses_event(self):
        """persistent event on the communication stream: "the loss all-reduce of the last step has read losses_dev" """
        if not hasattr(self, '_lev'):
            self._lev = self.cdev.event_create()
        return self._lev

    def _gather_events(self):
        """one persistent event per net on the communication stream: "this net's updated parameters are gathered" """
        if not hasattr(self, '_gev'):
            self._gev = {k: self.cdev.event_create() for k in ('dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc')}
        return self._gev

    def _bind_loss_scale(self):
        """(re)attach this engine's dynamic loss-scale state to its contexts, or detach what another engine left there: the
        contexts are the caller's and several engines may live on one (an fp16 engine's state on it would scale a later
        engine's loss seeds by 2^15; found by running a bf16x2 model after an fp16 one on one Device).  The state pointer is
        read when a launch is issued or recorded, so binding at every issue point is enough -- two host calls."""
        mine = {id(d): t for d, t in self._ls_state}
        for d in {id(d): d for d in self.devs}.values():
            if hasattr(d, 'set_loss_scale_state'):
                d.set_loss_scale_state(mine.get(id(d)))

    def _run_lanes(self, b, name, lanes, wrap=None):
        """Run one launch list per stream: eager (interleaved so both streams fill) on the first call,
        captured into one HIP graph per stream on the second, replayed afterwards.  ``wrap(lane, entry)``
        replaces the plain call (used by bench.py to bracket kernels with HIP events; implies eager)."""
        self._bind_loss_scale()
        if wrap is not None or self.use_graph is not True:
            for lane, e in _interleave(lanes[0], lanes[1]):
                if wrap is not None:
                    wrap(lane, e)
                else:
                    e[1]()
            return
        n = b.calls.get(name, 0)
        b.calls[name] = n + 1
        if n == 0:
            for lane, e in _interleave(lanes[0], lanes[1]):
                e[1]()
            return
        if name not in b.graphs:
            gs = []
            for lane in (0, 1):
                if not lanes[lane]:
                    gs.append(None)
                    continue
                self.devs[lane].capture_begin()
                try:
                    for e in lanes[lane]:
                        e[1]()
                finally:
                    gs.append(self.devs[lane].capture_end())
            b.graphs[name] = gs
            stages = [(self.devs[lane], g) for lane, g in enumerate(gs) if g is not None]
            b.steps = getattr(b, 'steps', {})
            b.steps[name] = type(self.devs[0]).step_build(stages) if stages else None
        if b.steps.get(name) is not None:
            type(self.devs[0]).step_run(b.steps[name])          # the whole stage-parallel step: one C call

    def _read_losses(self):
        self.sync()
        v = self.losses_dev.numpy().ravel()[:5].astype(np.float64)
        if self.world > 1:
            v = v / self.world
        return [np.float32(x) for x in v]

    def train(self, Z, X, Y, read_losses=True):
        b = self.built(int(np.shape(X)[0]))
        self._upload(b, Z, X, Y)
        self.enqueue_train(b)
        return self._read_losses() if read_losses else None

    def run_from_iterator(self, it, Z_sampler, train=True):
        """One train_fn / loss_fn call whose (A, B) batch is produced on the device by a data.Hdf5Iterator
        (uint8 upload + ghm_image_batch straight into the step's input buffers; no fp32 host batch)."""
        n = it.peek_n()
        b = self.built(n)
        self.sync()
        it.next_into(b.x, b.y)
        b.z.set(np.ascontiguousarray(Z_sampler(n), np.float32))
        self.sync()
        if train:
            self.enqueue_train(b)
        else:
            self._run_loss(b)
        return self._read_losses()

    def _run_loss(self, b):
        if self.use_graph == 'recorded':
            self._run_recorded(b, 'loss')
        else:
            self._run_lanes(b, 'loss', b.loss_prog)
        if self.exchange:
            self._reduce_losses_now()

    def _all_devs(self):
        out = []
        for d in list(self.devs) + [sd[0] for sd in self.side if sd is not None] + [self.cdev]:
            if d is not None and all(d is not o for o in out):
                out.append(d)
        return out

    def _sequence(self, b, name):
        """the host-order launch sequence [(lane, entry)] of a whole call (both stage programs interleaved, then the
        exchange, then the updates)"""
        key = '_seq_' + name
        if not hasattr(b, key):
            if name == 'train':
                seq = list(_interleave(b.train_compute[0], b.train_compute[1]))
                seq += [(0, e) for e in b.exchange]
                seq += list(_interleave(b.update[0], b.update[1]))
            else:
                seq = list(_interleave(b.loss_prog[0], b.loss_prog[1]))
            setattr(b, key, seq)
        return getattr(b, key)

    def _run_recorded(self, b, name, wrap=None):
        """call 0 eager (library workspaces take their size), call 1 records the sequence and replays it, later calls
        are ONE ghm_step_run each.  ``wrap(lane, entry)`` at record time brackets entries with recorded timers."""
        self._bind_loss_scale()
        seq = self._sequence(b, name)
        n = b.calls.get(name, 0)
        b.calls[name] = n + 1
        if n == 0:
            for lane, e in seq:
                e[1]()
            return
        b.steps = getattr(b, 'steps', {})
        if name not in b.steps:
            D = type(self.devs[0])
            st = D.step_record_begin(self._all_devs())
            try:
                for lane, e in seq:
                    if wrap is not None:
                        wrap(lane, e)
                    else:
                        e[1]()
            finally:
                D.step_record_end(st)
            b.steps[name] = st
        type(self.devs[0]).step_run(b.steps[name])

    def enqueue_train(self, b, wrap=None):
        """one train step on the data already resident in b.z / b.x / b.y (asynchronous)"""
        if self.use_graph == 'recorded':
            return self._run_recorded(b, 'train', wrap)
        if self.exchange:
            self._run_lanes(b, 'train_compute', b.train_compute, wrap)     # eager: bucket all-reduces are inside
            for e in b.exchange:
                e[1]()
            self._run_lanes(b, 'train_update', b.update, wrap)
        else:
            if not hasattr(b, 'train_all'):
                b.train_all = [b.train_compute[0] + b.update[0], b.train_compute[1] + b.update[1]]
            self._run_lanes(b, 'train_all', b.train_all, wrap)

    def _reduce_losses_now(self):
        self.sync()
        self.cops.allreduce_sum(self.losses_dev, 8)

    def loss(self, Z, X, Y):
        b = self.built(int(np.shape(X)[0]))
        self._upload(b, Z, X, Y)
        self._run_loss(b)
        return self._read_losses()

    def profile_train(self, B):
        """[(label, ms, meta)] per program entry, stream by stream (synchronising; mutates parameters like a
        real step: compute, then the exchange, then the updates)."""
        b = self.built(B)
        out = []

        def timed(entries, dev, lane):
            for e in entries:
                side = len(e) > 3 and e[3] is not None
                d = e[3] if side else dev
                d.timer_start(1)
                e[1]()
                d.timer_stop(1)
                meta = e[2] if len(e) > 2 else None
                out.append((e[0], d.timer_ms(1), meta, "%s%s" % ("AB"[lane] if lane in (0, 1) else "C", "'" if side else "")))

        for lane in (0, 1):
            timed(b.train_compute[lane], self.devs[lane], lane)
        if self.exchange:
            self.sync()
            timed(b.exchange, self.cdev, 2)
            self.sync()
        for lane in (0, 1):
            timed(b.update[lane], self.devs[lane], lane)
        return out

    # ---- forward-only entry points (pix2pix.py:144-147) -------------------------------------------------
    def _infer_plan(self, key, B, deterministic):
        k = (key, B, deterministic)
        if k not in self._infer:
            lane = LANE_OF[key]
            plan = NetPlan(self.devs[lane], self.ops[lane], self.nets[key], B, self.stores[key], name=key + "_infer",
                           dtype=self.dtype)
            prog = []
            plan.emit_forward(prog, deterministic=deterministic)
            self._infer[k] = (plan, prog)
        return self._infer[k]

    def generate(self, key, inp, deterministic=False):
        inp = np.ascontiguousarray(inp, np.float32)
        plan, prog = self._infer_plan(key, inp.shape[0], deterministic)
        self.sync()
        plan.input_nodes[0].out.set(inp)
        for e in prog:
            e[1]()
        return plan.out.numpy()

    def generate_chain(self, Z, deterministic=True):
        """z -> G(z) -> U(G(z)) without leaving HBM (the z_fn -> gen_fn chain of generate_interpolation_clip,
        /root/reference/pix2pix.py:384-393).  Returns (heightmaps, textures) as numpy arrays."""
        Z = np.ascontiguousarray(Z, np.float32)
        pg, prog_g = self._infer_plan('dcgan_gen', Z.shape[0], deterministic)
        pu, prog_u = self._infer_plan('p2p_gen', Z.shape[0], deterministic)
        self.sync()
        pg.input_nodes[0].out.set(Z)
        for e in prog_g:
            e[1]()
        lg, lu = LANE_OF['dcgan_gen'], LANE_OF['p2p_gen']
        if self.devs[lu] is not self.devs[lg]:
            self.devs[lu].wait_for(self.devs[lg])
        self.ops[lu].copy_view(pg.out, pu.input_nodes[0].out)
        for e in prog_u:
            e[1]()
        a = pg.out.numpy()
        return a, pu.out.numpy()


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
import numpy as np

from . import layers as L
from .device import Device, Ops
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
                 side_streams=None):
        self.dev = dev
        if side_streams is None:            # forked branches replay slowly inside a HIP graph: eager mode only
            side_streams = (not use_graph) and two_streams
        if side_streams and use_graph:
            raise ValueError("side_streams needs use_graph=False")
        self.devs = [dev, Device(dev.index) if two_streams else dev]
        self.ops = [Ops(self.devs[0]), Ops(self.devs[1])]
        # optional second stream per stage for the weight / bias gradients (engine.NetPlan side=)
        self.side = [None, None]
        if side_streams:
            sd = [Device(dev.index), Device(dev.index) if two_streams else None]
            if sd[1] is None:
                sd[1] = sd[0]
            self.side = [(sd[0], Ops(sd[0])), (sd[1], Ops(sd[1]))]
        self.nets = {'dcgan_gen': dcgan_gen, 'dcgan_disc': dcgan_disc, 'p2p_gen': p2p_gen,
                     'p2p_disc': p2p_disc["out"]}
        self.p2p_disc_inputs = p2p_disc["inputs"]
        self.alpha, self.lsgan, self.reconstruction = float(alpha), bool(lsgan), reconstruction
        self.opt_spec, self.train_mode = opt_spec, train_mode
        self.comm = comm
        self.world = comm.world if comm is not None else 1
        # data-parallel code path (stream hand-over, RCCL all-reduce, updates on stream A) even with one rank:
        # lets a single-GPU box exercise exactly what N ranks run
        self.exchange = self.world > 1 or (force_exchange and comm is not None)
        self.use_graph = use_graph
        self.stores = {k: ParamStore(self.devs[LANE_OF[k]], L.get_all_params(v)) for k, v in self.nets.items()}
        # per-net optimiser state + hyper-parameter scalars [lr, t] in HBM
        self.hyper = {}
        lr = float(opt_spec.learning_rate.get_value()) if hasattr(opt_spec.learning_rate, 'get_value') \
            else float(opt_spec.learning_rate)
        for k, st in self.stores.items():
            d = self.devs[LANE_OF[k]]
            self.hyper[k] = d.tensor(np.array([lr, 0.0], np.float32))
            n = max(st.n_train, 1)
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

    def sync(self):
        self.devs[0].sync()
        if self.devs[1] is not self.devs[0]:
            self.devs[1].sync()

    def set_lr(self, lr):
        self.sync()
        for k, h in self.hyper.items():
            cur = h.numpy().ravel()
            cur[0] = lr
            h.set(cur)

    # ---- building -------------------------------------------------------------------------------------
    def _build(self, B):
        b = _Built()
        dA, dB = self.devs
        oA, oB = self.ops
        G, D, U, P = (self.nets[k] for k in ('dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc'))
        d_in_layer = [l for l in L.get_all_layers(D) if isinstance(l, L.InputLayer)][0]
        u_in_layer = [l for l in L.get_all_layers(U) if isinstance(l, L.InputLayer)][0]
        i_a, i_b = self.p2p_disc_inputs
        ca, H, W = d_in_layer.shape[1:]
        b.d_in = dA.empty((2 * B, ca, H, W))
        b.G = NetPlan(dA, oA, G, B, self.stores['dcgan_gen'], out_tensor=b.d_in.samples(B, 2 * B), name="G",
                      side=self.side[0])
        b.D = NetPlan(dA, oA, D, 2 * B, self.stores['dcgan_disc'], inputs={d_in_layer: b.d_in}, name="D",
                      side=self.side[0], bn_groups=2 if _has_bn(D) else 1)
        b.P = NetPlan(dB, oB, P, 2 * B, self.stores['p2p_disc'], name="P", side=self.side[1],
                      bn_groups=2 if _has_bn(P) else 1)
        pa, pb = b.P.input_tensor(i_a), b.P.input_tensor(i_b)
        b.U = NetPlan(dB, oB, U, B, self.stores['p2p_gen'], out_tensor=pb.samples(B, 2 * B), name="U",
                      side=self.side[1])
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
        fa = [("x_to_d_in", lambda: oA.copy_view(b.x, b.d_in.samples(0, B)))]
        b.G.emit_forward(fa)
        b.D.emit_forward(fa)
        fb = [("x_to_p_in0", lambda: oB.copy_view(b.x, pa.samples(0, B))),
              ("x_to_p_in1", lambda: oB.copy_view(b.x, pa.samples(B, 2 * B))),
              ("y_to_p_in", lambda: oB.copy_view(b.y, pb.samples(0, B)))]
        b.U.emit_forward(fb)
        b.P.emit_forward(fb)
        d_out, p_out = b.D.out, b.P.out
        d_real, d_fake = d_out.samples(0, B), d_out.samples(B, 2 * B)
        p_real, p_fake = p_out.samples(0, B), p_out.samples(B, 2 * B)
        b.seed_D, b.seed_G = dA.empty(d_out.shape), dA.empty(d_fake.shape)
        b.seed_PD, b.seed_PG = dB.empty(p_out.shape), dB.empty(p_fake.shape)

        def losses_a(prog, g):
            # (:107) gen_loss_dcgan, (:108) disc_loss_dcgan
            prog.append(("loss", lambda: advA(d_fake, 1.0, slot(0), b.seed_G if g else None)))
            prog.append(("loss", lambda: advA(d_real, 1.0, slot(1), b.seed_D.samples(0, B) if g else None)))
            prog.append(("loss", lambda: advA(d_fake, 0.0, slot(1), b.seed_D.samples(B, 2 * B) if g else None,
                                              1.0, True)))

        def losses_b(prog, g):
            # (:110) gen_loss_p2p, (:121) disc_loss_p2p
            prog.append(("loss", lambda: advB(p_fake, 1.0, slot(2), b.seed_PG if g else None)))
            prog.append(("loss", lambda: advB(p_real, 1.0, slot(4), b.seed_PD.samples(0, B) if g else None)))
            prog.append(("loss", lambda: advB(p_fake, 0.0, slot(4), b.seed_PD.samples(B, 2 * B) if g else None,
                                              1.0, True)))

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
        if do_dcgan:
            b.D.emit_transposes(ta, tdone)
            b.G.emit_transposes(ta, tdone)
            b.D.emit_backward(ta, b.seed_D, wgrad=True, tag="dloss", transposed=tdone)
            gin = b.D.emit_backward(ta, b.seed_G, nslice=(B, 2 * B), wgrad=False, input_grads=[d_in_layer],
                                    tag="gloss", transposed=tdone)
            b.G.emit_backward(ta, gin[d_in_layer], wgrad=True, transposed=tdone)
        if do_p2p:
            b.P.emit_transposes(tb, tdone)
            b.U.emit_transposes(tb, tdone)
            b.P.emit_backward(tb, b.seed_PD, wgrad=True, tag="dloss", transposed=tdone)
            gin = b.P.emit_backward(tb, b.seed_PG, nslice=(B, 2 * B), wgrad=False, input_grads=[i_b], tag="gloss",
                                    transposed=tdone)
            gu = gin[i_b]
            # (:115-117) recon loss and alpha * d recon / d U(X) added to the adversarial gradient
            tb.append(("recon", lambda: oB.recon_loss(b.U.out, b.y, slot(3), gu, self.alpha, l2, True)))
            b.U.emit_backward(tb, gu, wgrad=True, transposed=tdone)
        else:
            tb.append(("recon", lambda: oB.recon_loss(b.U.out, b.y, slot(3), None, 1.0, l2)))
        if self.side[0] is not None:        # the gradient streams rejoin before anything consumes the gradients
            ta.append(("join", lambda: dA.wait_for(self.side[0][0])))
            tb.append(("join", lambda: dB.wait_for(self.side[1][0])))
        b.train_compute = [ta, tb]
        # ---- exchange + update (:131-141) ----
        keys = (['dcgan_gen', 'dcgan_disc'] if do_dcgan else []) + (['p2p_gen', 'p2p_disc'] if do_p2p else [])
        b.exchange = []
        if self.exchange:
            # one communicator, every collective on stream A in one fixed order on all ranks: the DCGAN buckets
            # (stream A's own work, ready in stream order) go first and overlap the tail of the pix2pix stream;
            # then stream A waits for stream B, sums the pix2pix buckets and the losses, updates everything,
            # and stream B continues behind it
            for k in keys:
                st = self.stores[k]
                b.exchange.append(("allreduce_" + k, lambda st=st: oA.allreduce_sum(st.g, st.n_train), LANE_OF[k]))
            b.exchange.append(("allreduce_losses", lambda: oA.allreduce_sum(lo, 8), 1))
        gs = 1.0 / self.world
        hp = self.opt_spec.hp
        b.update = [[], []]
        for k in keys:
            st, hy = self.stores[k], self.hyper[k]
            lane = 0 if self.exchange else LANE_OF[k]
            o = self.ops[lane]
            if self.opt_spec.kind == 'rmsprop':
                b.update[lane].append(("rmsprop_" + k, lambda st=st, hy=hy, o=o: o.rmsprop(
                    st.w, st.g, st.opt_state['acc'], st.n_train, hy, hp['rho'], hp['epsilon'], gs)))
            else:
                b.update[lane].append(("adam_" + k, lambda st=st, hy=hy, o=o: o.adam(
                    st.w, st.g, st.opt_state['m'], st.opt_state['v'], st.n_train, hy, hp['beta1'], hp['beta2'],
                    hp['epsilon'], gs)))
                b.update[lane].append(("adam_tick_" + k, lambda hy=hy, o=o: o.adam_tick(hy)))
        b.graphs = {}
        b.calls = {}
        return b

    def built(self, B):
        if B not in self._built:
            self._built[B] = self._build(B)
        return self._built[B]

    # ---- running --------------------------------------------------------------------------------------
    def _upload(self, b, Z, X, Y):
        self.sync()                 # the previous step may still be reading the input buffers
        b.z.set(Z)
        b.x.set(X)
        b.y.set(Y)
        self.sync()

    def _run_lanes(self, b, name, lanes, wrap=None):
        """Run one launch list per stream: eager (interleaved so both streams fill) on the first call,
        captured into one HIP graph per stream on the second, replayed afterwards.  ``wrap(lane, entry)``
        replaces the plain call (used by bench.py to bracket kernels with HIP events; implies eager)."""
        if wrap is not None or not self.use_graph:
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
            self._run_lanes(b, 'loss', b.loss_prog)
            if self.exchange:
                self.sync()
                self.ops[0].allreduce_sum(self.losses_dev, 8)
        return self._read_losses()

    def enqueue_train(self, b, wrap=None):
        """one train step on the data already resident in b.z / b.x / b.y (asynchronous)"""
        dA, dB = self.devs
        if self.exchange:
            self._run_lanes(b, 'train_compute', b.train_compute, wrap)
            for e in b.exchange:                            # RCCL calls stay outside the captured graphs
                if e[2] == 0:
                    e[1]()
            if dB is not dA:
                dA.wait_for(dB)
            for e in b.exchange:
                if e[2] != 0:
                    e[1]()
            self._run_lanes(b, 'train_update', b.update, wrap)
            if dB is not dA:
                dB.wait_for(dA)
        else:
            if not hasattr(b, 'train_all'):
                b.train_all = [b.train_compute[0] + b.update[0], b.train_compute[1] + b.update[1]]
            self._run_lanes(b, 'train_all', b.train_all, wrap)

    def loss(self, Z, X, Y):
        b = self.built(int(np.shape(X)[0]))
        self._upload(b, Z, X, Y)
        self._run_lanes(b, 'loss', b.loss_prog)
        if self.exchange:
            self.sync()
            self.ops[0].allreduce_sum(self.losses_dev, 8)
        return self._read_losses()

    def profile_train(self, B):
        """[(label, ms, meta)] per program entry, stream by stream (synchronising; mutates parameters like a
        real step)."""
        b = self.built(B)
        out = []
        for lane in (0, 1):
            dev = self.devs[lane]
            for e in b.train_compute[lane] + b.update[lane]:
                d = e[3] if len(e) > 3 and e[3] is not None else dev
                d.timer_start(1)
                e[1]()
                d.timer_stop(1)
                out.append((e[0], d.timer_ms(1), e[2] if len(e) > 2 else None))
        if self.exchange:
            self.sync()
            for e in b.exchange:
                e[1]()
            self.sync()
        return out

    # ---- forward-only entry points (pix2pix.py:144-147) -------------------------------------------------
    def _infer_plan(self, key, B, deterministic):
        k = (key, B, deterministic)
        if k not in self._infer:
            lane = LANE_OF[key]
            plan = NetPlan(self.devs[lane], self.ops[lane], self.nets[key], B, self.stores[key], name=key + "_infer")
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


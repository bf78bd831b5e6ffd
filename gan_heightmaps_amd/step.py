"""The two-stage GAN step of the reference (/root/reference/pix2pix.py:87-147) as static device programs.

One ``GanStep`` owns the four networks' parameters in HBM and, per batch size, a set of NetPlans wired the
way ``Pix2Pix.__init__`` wires the Theano graph:
  * G(z) is evaluated once and written straight into the second half of the DCGAN discriminator's input
    batch, so D(X) and D(G(z)) run as ONE 2B-sample pass (D has no BatchNorm in any reference experiment;
    with bn=True the two passes are kept separate to preserve per-pass batch statistics),
  * U(X) is written straight into channels 1.. of the second half of the PatchGAN's concat buffer,
  * four gradient roots, each w.r.t. its own net's parameters, all at the pre-update parameters:
    D-loss backward on the 2B batch (weights + data gradients), G-loss backward through D on the fake half
    only (data gradients only), then through G; same for PatchGAN / U-Net with alpha * L1 added,
  * per net: (data-parallel: one RCCL all-reduce of the flat gradient buffer) then one optimiser kernel.
"""
import numpy as np

from . import layers as L
from .engine import NetPlan, ParamStore, run_program, time_program
from .device import Ops

TRAIN_KEYS = ['dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_recon', 'p2p_disc']


def _has_bn(layer):
    return any(isinstance(l, L.BatchNormLayer) for l in L.get_all_layers(layer))


class _Built:
    pass


class GanStep:
    def __init__(self, dev, dcgan_gen, dcgan_disc, p2p_gen, p2p_disc, alpha, lsgan, reconstruction, opt_spec,
                 train_mode='both', comm=None, use_graph=True):
        self.dev, self.ops = dev, Ops(dev)
        self.nets = {'dcgan_gen': dcgan_gen, 'dcgan_disc': dcgan_disc, 'p2p_gen': p2p_gen,
                     'p2p_disc': p2p_disc["out"]}
        self.p2p_disc_inputs = p2p_disc["inputs"]
        self.alpha, self.lsgan, self.reconstruction = float(alpha), bool(lsgan), reconstruction
        self.opt_spec, self.train_mode = opt_spec, train_mode
        self.comm = comm
        self.world = comm.world if comm is not None else 1
        self.use_graph = use_graph
        for k in ('dcgan_disc', 'p2p_disc'):
            if _has_bn(self.nets[k]):
                raise NotImplementedError("discriminators with BatchNorm: the batched real|fake pass would mix "
                                          "their batch statistics (no reference experiment enables it)")
        self.stores = {k: ParamStore(dev, L.get_all_params(v)) for k, v in self.nets.items()}
        # per-net optimiser state + hyper-parameter scalars [lr, t] in HBM
        self.hyper = {}
        lr = float(opt_spec.learning_rate.get_value()) if hasattr(opt_spec.learning_rate, 'get_value') \
            else float(opt_spec.learning_rate)
        for k, st in self.stores.items():
            self.hyper[k] = dev.tensor(np.array([lr, 0.0], np.float32))
            n = max(st.n_train, 1)
            if opt_spec.kind == 'rmsprop':
                st.opt_state = {'acc': dev.zeros((1, n, 1, 1))}
            elif opt_spec.kind == 'adam':
                st.opt_state = {'m': dev.zeros((1, n, 1, 1)), 'v': dev.zeros((1, n, 1, 1))}
            else:
                raise ValueError(opt_spec.kind)
        if hasattr(opt_spec.learning_rate, '_listeners'):
            opt_spec.learning_rate._listeners.append(self.set_lr)
        self.losses_dev = dev.zeros((1, 8, 1, 1))
        self._built = {}
        self._infer = {}

    def set_lr(self, lr):
        for k, h in self.hyper.items():
            cur = h.numpy().ravel()
            cur[0] = lr
            h.set(cur)

    # ---- building -------------------------------------------------------------------------------------
    def _build(self, B):
        dev, ops = self.dev, self.ops
        b = _Built()
        G, D, U, P = (self.nets[k] for k in ('dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc'))
        d_in_layer = [l for l in L.get_all_layers(D) if isinstance(l, L.InputLayer)][0]
        u_in_layer = [l for l in L.get_all_layers(U) if isinstance(l, L.InputLayer)][0]
        i_a, i_b = self.p2p_disc_inputs
        ca, H, W = d_in_layer.shape[1:]
        b.d_in = dev.empty((2 * B, ca, H, W))
        b.G = NetPlan(dev, ops, G, B, self.stores['dcgan_gen'], out_tensor=b.d_in.samples(B, 2 * B), name="G")
        b.D = NetPlan(dev, ops, D, 2 * B, self.stores['dcgan_disc'], inputs={d_in_layer: b.d_in}, name="D")
        b.P = NetPlan(dev, ops, P, 2 * B, self.stores['p2p_disc'], name="P")
        pa, pb = b.P.input_tensor(i_a), b.P.input_tensor(i_b)
        b.U = NetPlan(dev, ops, U, B, self.stores['p2p_gen'], out_tensor=pb.samples(B, 2 * B), name="U")
        b.z = b.G.input_nodes[0].out
        b.x = b.U.input_tensor(u_in_layer)
        b.y = dev.empty((B,) + tuple(pb.shape[1:]))
        b.B = B
        lo = self.losses_dev
        slot = lambda i: lo.channels(i, i + 1)
        adv = ops.lsgan_loss if self.lsgan else ops.bce_loss

        # ---- shared forward (pix2pix.py:92-101) ----
        fwd = []
        fwd.append(("x_to_d_in", lambda: ops.copy_view(b.x, b.d_in.samples(0, B))))
        fwd.append(("x_to_p_in0", lambda: ops.copy_view(b.x, pa.samples(0, B))))
        fwd.append(("x_to_p_in1", lambda: ops.copy_view(b.x, pa.samples(B, 2 * B))))
        fwd.append(("y_to_p_in", lambda: ops.copy_view(b.y, pb.samples(0, B))))
        b.G.emit_forward(fwd)
        b.D.emit_forward(fwd)
        b.U.emit_forward(fwd)
        b.P.emit_forward(fwd)
        d_out, p_out = b.D.out, b.P.out
        d_real, d_fake = d_out.samples(0, B), d_out.samples(B, 2 * B)
        p_real, p_fake = p_out.samples(0, B), p_out.samples(B, 2 * B)
        b.seed_D, b.seed_G = dev.empty(d_out.shape), dev.empty(d_fake.shape)
        b.seed_PD, b.seed_PG = dev.empty(p_out.shape), dev.empty(p_fake.shape)

        def losses(prog, with_grads):
            g = with_grads
            # (:107) gen_loss_dcgan, (:108) disc_loss_dcgan
            prog.append(("loss", lambda: adv(d_fake, 1.0, slot(0), b.seed_G if g else None)))
            prog.append(("loss", lambda: adv(d_real, 1.0, slot(1), b.seed_D.samples(0, B) if g else None)))
            prog.append(("loss", lambda: adv(d_fake, 0.0, slot(1), b.seed_D.samples(B, 2 * B) if g else None,
                                             1.0, True)))
            # (:110) gen_loss_p2p, (:121) disc_loss_p2p
            prog.append(("loss", lambda: adv(p_fake, 1.0, slot(2), b.seed_PG if g else None)))
            prog.append(("loss", lambda: adv(p_real, 1.0, slot(4), b.seed_PD.samples(0, B) if g else None)))
            prog.append(("loss", lambda: adv(p_fake, 0.0, slot(4), b.seed_PD.samples(B, 2 * B) if g else None,
                                             1.0, True)))

        # ---- loss_fn (:143): forward + losses, BN running stats still update ----
        b.loss_prog = list(fwd)
        losses(b.loss_prog, False)
        b.loss_prog.append(("recon", lambda: ops.recon_loss(b.U.out, b.y, slot(3), None, 1.0,
                                                            self.reconstruction == 'l2')))

        # ---- train_fn (:142) ----
        tr = list(fwd)
        losses(tr, True)
        do_dcgan = self.train_mode in ('both', 'dcgan')
        do_p2p = self.train_mode in ('both', 'p2p')
        tdone = set()      # conv weights whose transposed copy is already fresh in this program
        if do_dcgan:
            b.D.emit_backward(tr, b.seed_D, wgrad=True, tag="dloss", transposed=tdone)
            gin = b.D.emit_backward(tr, b.seed_G, nslice=(B, 2 * B), wgrad=False, input_grads=[d_in_layer],
                                    tag="gloss", transposed=tdone)
            b.G.emit_backward(tr, gin[d_in_layer], wgrad=True, transposed=tdone)
        if do_p2p:
            b.P.emit_backward(tr, b.seed_PD, wgrad=True, tag="dloss", transposed=tdone)
            gin = b.P.emit_backward(tr, b.seed_PG, nslice=(B, 2 * B), wgrad=False, input_grads=[i_b], tag="gloss",
                                    transposed=tdone)
            gu = gin[i_b]
            # (:115-117) recon loss and alpha * d recon / d U(X) added to the adversarial gradient
            tr.append(("recon", lambda: ops.recon_loss(b.U.out, b.y, slot(3), gu, self.alpha,
                                                       self.reconstruction == 'l2', True)))
            b.U.emit_backward(tr, gu, wgrad=True, transposed=tdone)
        else:
            tr.append(("recon", lambda: ops.recon_loss(b.U.out, b.y, slot(3), None, 1.0,
                                                       self.reconstruction == 'l2')))
        b.train_compute = tr
        # ---- exchange + update (:131-141) ----
        keys = (['dcgan_gen', 'dcgan_disc'] if do_dcgan else []) + (['p2p_gen', 'p2p_disc'] if do_p2p else [])
        b.exchange = []
        if self.world > 1:
            for k in keys:
                st = self.stores[k]
                b.exchange.append(("allreduce_" + k, lambda st=st: ops.allreduce_sum(st.g, st.n_train)))
            b.exchange.append(("allreduce_losses", lambda: ops.allreduce_sum(lo, 8)))
        upd = []
        gs = 1.0 / self.world
        hp = self.opt_spec.hp
        for k in keys:
            st, hy = self.stores[k], self.hyper[k]
            if self.opt_spec.kind == 'rmsprop':
                upd.append(("rmsprop_" + k, lambda st=st, hy=hy: ops.rmsprop(
                    st.w, st.g, st.opt_state['acc'], st.n_train, hy, hp['rho'], hp['epsilon'], gs)))
            else:
                upd.append(("adam_" + k, lambda st=st, hy=hy: ops.adam(
                    st.w, st.g, st.opt_state['m'], st.opt_state['v'], st.n_train, hy, hp['beta1'], hp['beta2'],
                    hp['epsilon'], gs)))
                upd.append(("adam_tick_" + k, lambda hy=hy: ops.adam_tick(hy)))
        b.update = upd
        b.graphs = {}
        b.calls = {'train': 0, 'loss': 0}
        return b

    def built(self, B):
        if B not in self._built:
            self._built[B] = self._build(B)
        return self._built[B]

    # ---- running --------------------------------------------------------------------------------------
    def _upload(self, b, Z, X, Y):
        b.z.set(Z)
        b.x.set(X)
        b.y.set(Y)

    def _run_segment(self, b, name, prog):
        """eager on the first call, captured into a HIP graph on the second, replayed afterwards"""
        if not self.use_graph or not prog:
            run_program(prog)
            return
        n = b.calls.get(name, 0)
        b.calls[name] = n + 1
        if n == 0:
            run_program(prog)
        else:
            if name not in b.graphs:
                self.dev.capture_begin()
                try:
                    run_program(prog)
                finally:
                    b.graphs[name] = self.dev.capture_end()
            self.dev.graph_launch(b.graphs[name])

    def _read_losses(self):
        v = self.losses_dev.numpy().ravel()[:5].astype(np.float64)
        if self.world > 1:
            v = v / self.world
        return [np.float32(x) for x in v]

    def train(self, Z, X, Y, read_losses=True):
        b = self.built(int(np.shape(X)[0]))
        self._upload(b, Z, X, Y)
        self.enqueue_train(b)
        return self._read_losses() if read_losses else None

    def enqueue_train(self, b):
        """one train step on the data already resident in b.z / b.x / b.y (asynchronous)"""
        if self.world > 1:
            self._run_segment(b, 'train_compute', b.train_compute)
            run_program(b.exchange)                      # RCCL calls stay outside the captured graphs
            self._run_segment(b, 'train_update', b.update)
        else:
            if 'train_all' not in b.__dict__:
                b.train_all = b.train_compute + b.update
            self._run_segment(b, 'train_all', b.train_all)

    def loss(self, Z, X, Y):
        b = self.built(int(np.shape(X)[0]))
        self._upload(b, Z, X, Y)
        self._run_segment(b, 'loss', b.loss_prog)
        if self.world > 1:
            self.ops.allreduce_sum(self.losses_dev, 8)
        return self._read_losses()

    def profile_train(self, B):
        """[(label, ms)] per program entry (synchronising; mutates parameters like a real step)."""
        b = self.built(B)
        return time_program(self.dev, b.train_compute + b.update)

    # ---- forward-only entry points (pix2pix.py:144-147) -------------------------------------------------
    def _infer_plan(self, key, B, deterministic):
        k = (key, B, deterministic)
        if k not in self._infer:
            plan = NetPlan(self.dev, self.ops, self.nets[key], B, self.stores[key], name=key + "_infer")
            prog = []
            plan.emit_forward(prog, deterministic=deterministic)
            self._infer[k] = (plan, prog)
        return self._infer[k]

    def generate(self, key, inp, deterministic=False):
        inp = np.ascontiguousarray(inp, np.float32)
        plan, prog = self._infer_plan(key, inp.shape[0], deterministic)
        plan.input_nodes[0].out.set(inp)
        run_program(prog)
        return plan.out.numpy()

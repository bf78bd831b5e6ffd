"""``Pix2Pix``: the reference's trainer surface (/root/reference/pix2pix.py:19-425) on the libghm backend.

Same constructor arguments, attributes (train_fn, loss_fn, gen_fn, gen_fn_det, z_fn, z_fn_det, dcgan, p2p, lr,
train_keys) and methods (save_model, load_model, train, generate_*); compiled functions take numpy float32
arrays and return lists of numpy scalars / arrays, synchronously.
Extra keyword arguments (device, comm, use_graph, seed, dtype) configure the MI355X backend; ``dtype`` is the
arithmetic of the convolution products: 'bf16x3' (default = the reference's floatX=float32 as six exact bf16 piece products
per multiply-add on the bf16 matrix cores: csrc/conv_split.hip, fp32-accurate, held to the fp32 parity bounds, 1.55x the speed),
'f32' (the same arithmetic on the fp32 matrix instruction v_mfma_f32_32x32x2_f32) or 'bf16' / 'f16' (operands rounded).
"""
import gzip
import os
import pickle
from time import time

import numpy as np

from . import init as _init
from . import layers as L
from .device import Device
from .step import GanStep, TRAIN_KEYS
from .updates import adam, shared, OptimizerSpec
from .util import convert_to_rgb, imsave, makedirs, plot_grid, writes as util_writes

floatX = _init.floatX


class _Py2CompatPickler(pickle._Pickler):
    """numpy >= 2 pickles an ndarray through the global ``numpy._core.multiarray._reconstruct``; the reference's
    environment (Python 2, numpy <= 1.16) only has ``numpy.core.multiarray``, which every numpy up to 2.x still
    resolves.  Write that spelling so a checkpoint saved here loads in the reference (pix2pix.py:174-186)."""

    def save_global(self, obj, name=None):
        mod = getattr(obj, '__module__', None) or ''
        if mod.startswith('numpy._core'):
            nm = name or getattr(obj, '__qualname__', obj.__name__)
            self.write(pickle.GLOBAL + ('numpy.core' + mod[len('numpy._core'):]).encode() + b'\n' +
                       nm.encode() + b'\n')
            self.memoize(obj)
            return
        pickle._Pickler.save_global(self, obj, name)


class Pix2Pix:
    def _print_network(self, l_out):
        for layer in L.get_all_layers(l_out):
            print(layer, layer.output_shape, "" if not hasattr(layer, 'nonlinearity') else layer.nonlinearity)
        print("# learnable params:", L.count_params(l_out, trainable=True))

    def __init__(self,
                 gen_fn_dcgan, disc_fn_dcgan,
                 gen_params_dcgan, disc_params_dcgan,
                 gen_fn_p2p, disc_fn_p2p,
                 gen_params_p2p, disc_params_p2p,
                 in_shp, latent_dim, is_a_grayscale, is_b_grayscale,
                 alpha=100, opt=adam, opt_args=None,
                 train_mode='both', reconstruction='l1', sampler=np.random.rand, lsgan=False, verbose=True,
                 device=None, comm=None, use_graph=True, seed=None, two_streams=True, force_exchange=False,
                 side_streams=None, dtype='bf16x3', bucket_mb=None, prefetch=True, exchange_mode=None):
        """Two-stage DCGAN / pix2pix GAN (see the reference docstring, pix2pix.py:32-64).
        gen_fn_dcgan(latent_dim, is_a_grayscale, **gen_params_dcgan) -> output layer
        disc_fn_dcgan(in_shp, is_a_grayscale, **disc_params_dcgan) -> output layer
        gen_fn_p2p(in_shp, is_a_grayscale, is_b_grayscale, **gen_params_p2p) -> output layer
        disc_fn_p2p(in_shp, is_a_grayscale, is_b_grayscale, **disc_params_p2p) -> {"inputs": [a, b], "out": layer}
        opt / opt_args: gan_heightmaps_amd.updates.{rmsprop, adam} and its kwargs; 'learning_rate' may be a
        shared scalar (default adam with shared(1e-3), pix2pix.py:30)."""
        assert train_mode in ['dcgan', 'p2p', 'both']
        assert reconstruction in ['l1', 'l2']
        if opt_args is None:
            opt_args = {'learning_rate': shared(floatX(1e-3))}
        self.is_a_grayscale = is_a_grayscale
        self.is_b_grayscale = is_b_grayscale
        self.latent_dim = latent_dim
        self.sampler = sampler
        self.in_shp = in_shp
        self.verbose = verbose
        self.train_mode = train_mode
        # train(): with host-array iterators the batch of step i+1 is drawn and uploaded (copy stream + page-locked staging)
        # while step i runs (GanStep.train_pipelined); False keeps the reference's strictly sequential upload -> step -> read
        self.prefetch = bool(prefetch)
        if seed is not None:
            _init.set_rng(np.random.RandomState(seed))
        # construction order fixes the RNG draw order of the initial weights (pix2pix.py:73-77)
        dcgan_gen = gen_fn_dcgan(latent_dim, is_a_grayscale, **gen_params_dcgan)
        dcgan_disc = disc_fn_dcgan(in_shp, is_a_grayscale, **disc_params_dcgan)
        p2p_gen = gen_fn_p2p(in_shp, is_a_grayscale, is_b_grayscale, **gen_params_p2p)
        p2p_disc = disc_fn_p2p(in_shp, is_a_grayscale, is_b_grayscale, **disc_params_p2p)
        if verbose:
            for label, net in [("dcgan gen:", dcgan_gen), ("dcgan disc:", dcgan_disc), ("p2p gen:", p2p_gen),
                               ("p2p disc:", p2p_disc["out"])]:
                print(label)
                self._print_network(net)
            print("train_mode: %s" % train_mode)
        self.dcgan = {'gen': dcgan_gen, 'disc': dcgan_disc}
        self.p2p = {'gen': p2p_gen, 'disc': p2p_disc["out"]}
        spec = opt(**opt_args)
        if not isinstance(spec, OptimizerSpec):
            raise TypeError("opt must be gan_heightmaps_amd.updates.rmsprop or .adam")
        self.lr = opt_args['learning_rate'] if 'learning_rate' in opt_args else spec.learning_rate
        if device is None:
            # one process per GPU: the launcher's LOCAL_RANK names this process' device (a communicator brings its own)
            device = Device(comm.dev.index) if comm is not None else Device(int(os.environ.get("LOCAL_RANK", "0")))
        self.device = device
        self.comm = comm
        self.engine = GanStep(self.device, dcgan_gen, dcgan_disc, p2p_gen, p2p_disc, alpha, lsgan, reconstruction,
                              spec, train_mode, comm=comm, use_graph=use_graph, two_streams=two_streams,
                              force_exchange=force_exchange, side_streams=side_streams, dtype=dtype,
                              bucket_mb=bucket_mb, exchange_mode=exchange_mode)
        self.train_keys = list(TRAIN_KEYS)
        eng = self.engine
        eng.broadcast_parameters()          # replicas start from rank 0's (possibly unseeded) initial weights
        self.train_fn = lambda Z, X, Y: eng.train(floatX(Z), floatX(X), floatX(Y))
        self._engine_train_fn = self.train_fn      # train() pipelines uploads only while train_fn is still the engine's own
        self.loss_fn = lambda Z, X, Y: eng.loss(floatX(Z), floatX(X), floatX(Y))
        self.gen_fn = lambda X: eng.generate('p2p_gen', X, False)
        self.gen_fn_det = lambda X: eng.generate('p2p_gen', X, True)
        self.z_fn = lambda Z: eng.generate('dcgan_gen', Z, False)
        self.z_fn_det = lambda Z: eng.generate('dcgan_gen', Z, True)

    def _is_writer(self):
        """files (results.txt, PNG dumps, checkpoints) are written by rank 0 only; every rank still runs the
        forward passes and iterator draws of the per-epoch dumps, which are part of the training trajectory"""
        comm = getattr(self, 'comm', None)
        return comm is None or comm.rank == 0

    # ---- checkpoint (pix2pix.py:158-186): gzip + pickle of get_all_param_values per net -------------------
    def save_model(self, filename):
        if not self._is_writer():
            return
        dd = {'dcgan': {'gen': L.get_all_param_values(self.dcgan['gen']),
                        'disc': L.get_all_param_values(self.dcgan['disc'])},
              'p2p': {'gen': L.get_all_param_values(self.p2p['gen']),
                      'disc': L.get_all_param_values(self.p2p['disc'])}}
        eng = getattr(self, 'engine', None)
        ls = eng.loss_scale_state() if hasattr(eng, 'loss_scale_state') else []
        if ls:          # fp16 only: the dynamic loss scale is training state (an extra key; the reference's loader ignores it)
            dd['loss_scale'] = [{k: float(v) for k, v in st.items()} for st in ls]
        with gzip.open(filename, "wb") as g:
            _Py2CompatPickler(g, 2).dump(dd)      # protocol 2 == py2 HIGHEST_PROTOCOL, readable by the reference

    def load_model(self, filename, mode='both'):
        assert mode in ['both', 'dcgan', 'p2p']
        with gzip.open(filename) as g:
            dd = pickle.load(g, encoding='latin1')      # genuine py2 checkpoints need latin1
        if mode in ('both', 'dcgan'):
            L.set_all_param_values(self.dcgan['gen'], dd['dcgan']['gen'])
            L.set_all_param_values(self.dcgan['disc'], dd['dcgan']['disc'])
        if mode in ('both', 'p2p'):
            L.set_all_param_values(self.p2p['gen'], dd['p2p']['gen'])
            L.set_all_param_values(self.p2p['disc'], dd['p2p']['disc'])
        if dd.get('loss_scale') and mode == 'both':
            self.engine.restore_loss_scale_state(dd['loss_scale'])

    # ---- training loop (pix2pix.py:187-275) ----------------------------------------------------------------
    def train(self, it_train, it_val, batch_size, num_epochs, out_dir, model_dir=None, save_every=10, resume=False,
              quick_run=False, validate_on_train_iterator=True, dump_images=True):
        """Same loop as the reference: per epoch N//batch_size train_fn steps then N//batch_size loss_fn steps,
        a CSV row of epoch means, then the per-epoch image dumps (a 4x4 grid of [A | U(A)] from ``it_val``, one batch of
        A->B pairs from each iterator, 20 DCGAN samples -- pix2pix.py:262-270; they advance the iterators, so they
        are part of the training trajectory; ``dump_images=False`` skips them), periodic checkpoints.  The
        reference's validation loop draws its batches from ``it_train`` (pix2pix.py:204);
        ``validate_on_train_iterator=True`` keeps that behaviour.  Checked event for event against the reference's
        own loop in tests/test_reference_trainloop.py."""
        def _next(it):
            return next(it) if hasattr(it, '__next__') else it.next()

        def _loop(fn, itr, src):
            rec = [[] for _ in self.train_keys]
            on_device = hasattr(src, 'next_into')          # gan_heightmaps_amd.data.Hdf5Iterator: batch stays in HBM
            eng = getattr(self, 'engine', None)
            own = fn is getattr(self, 'train_fn', None) and fn is getattr(self, '_engine_train_fn', None)   # (a replaced / wrapped train_fn is called as it is)
            if (own and not on_device and getattr(self, 'prefetch', False)
                    and hasattr(eng, 'train_pipelined')):
                # host arrays: the same draws in the same order, but batch i+1 is drawn, sampled and uploaded (copy stream,
                # page-locked staging) while step i runs; losses bit-identical to the call-by-call form
                def batches():
                    for _ in range(itr.N // batch_size):
                        X_batch, Y_batch = _next(src)
                        yield floatX(self.sampler(X_batch.shape[0], self.latent_dim)), floatX(X_batch), floatX(Y_batch)
                        if quick_run:
                            break
                try:
                    for results in eng.train_pipelined(batches()):
                        for i, r in enumerate(results):
                            rec[i].append(r)
                except BaseException:
                    eng.close_pipeline()        # an abandoned loop must not leave an upload in flight on the copy stream
                    raise
                return tuple(np.mean(elem) for elem in rec)
            if (own and on_device and getattr(self, 'prefetch', False)
                    and hasattr(eng, 'train_pipelined_from_iterator')):
                steps = 1 if quick_run else itr.N // batch_size
                try:
                    for results in eng.train_pipelined_from_iterator(src, lambda n: floatX(self.sampler(n, self.latent_dim)), steps):
                        for i, r in enumerate(results):
                            rec[i].append(r)
                except BaseException:
                    eng.close_pipeline()
                    raise
                return tuple(np.mean(elem) for elem in rec)
            for _ in range(itr.N // batch_size):
                if on_device:
                    results = self.engine.run_from_iterator(
                        src, lambda n: floatX(self.sampler(n, self.latent_dim)), train=fn is self.train_fn)
                else:
                    X_batch, Y_batch = _next(src)
                    Z_batch = floatX(self.sampler(X_batch.shape[0], self.latent_dim))
                    results = fn(Z_batch, X_batch, Y_batch)
                for i, r in enumerate(results):
                    rec[i].append(r)
                if quick_run:
                    break
            return tuple(np.mean(elem) for elem in rec)

        header = ["epoch"] + ["train_%s" % k for k in self.train_keys] + ["valid_%s" % k for k in self.train_keys] \
            + ["lr", "time", "mode"]
        writer = self._is_writer()
        if writer:
            os.makedirs(out_dir, exist_ok=True)
            if model_dir is not None:
                os.makedirs(model_dir, exist_ok=True)
        f = open("%s/results.txt" % out_dir if writer else os.devnull, "w" if not resume else "a")
        if not resume:
            f.write(",".join(header) + "\n")
            f.flush()
            if self.verbose:
                print(",".join(header))
        else:
            if self.verbose:
                print("loading weights from: %s" % resume)
            self.load_model(resume)
        for e in range(num_epochs):
            t0 = time()
            row = [str(e + 1)]
            row += [str(v) for v in _loop(self.train_fn, it_train, it_train)]
            row += [str(v) for v in _loop(self.loss_fn, it_val, it_train if validate_on_train_iterator else it_val)]
            lr = self.lr.get_value() if hasattr(self.lr, 'get_value') else self.lr
            row += [str(lr), str(time() - t0), self.train_mode]
            line = ",".join(row)
            if self.verbose:
                print(line)
            f.write(line + "\n")
            f.flush()
            self._check_loss_scale(e + 1)
            if dump_images:
                with util_writes(writer):
                    if self.train_mode in ['both', 'p2p']:
                        plot_grid("%s/out_%i.png" % (out_dir, e + 1), it_val, self.gen_fn,
                                  is_a_grayscale=self.is_a_grayscale, is_b_grayscale=self.is_b_grayscale)
                        self.generate_atob(it_train, 1, "%s/dump_train" % out_dir, deterministic=False)
                        self.generate_atob(it_val, 1, "%s/dump_valid" % out_dir, deterministic=False)
                    if self.train_mode in ['both', 'dcgan']:
                        self.generate_gz(num_examples=20, batch_size=batch_size, out_dir="%s/dump_a" % out_dir,
                                         deterministic=False)
            if model_dir is not None and (e + 1) % save_every == 0:
                self.save_model("%s/%i.model" % (model_dir, e + 1))
        f.close()

    def _check_loss_scale(self, epoch):
        """fp16 only: one line per epoch with the dynamic loss scale and the updates it skipped, and a warning when
        training has stalled on it -- a scale at its floor with every step still flagged means the FORWARD activations
        exceed the fp16 range, which no loss scale can fix (use bf16)"""
        eng = getattr(self, 'engine', None)
        ls = eng.loss_scale_state() if hasattr(eng, 'loss_scale_state') else []
        if not ls:
            return
        prev = getattr(self, '_ls_prev', [0] * len(ls))
        skipped = [st['skipped_steps'] - p for st, p in zip(ls, prev)]
        self._ls_prev = [st['skipped_steps'] for st in ls]
        if self.verbose:
            print("epoch %d: fp16 loss scale %s, updates skipped this epoch %s"
                  % (epoch, [st['scale'] for st in ls], skipped))
        for st, sk in zip(ls, skipped):
            if st['scale'] <= self.engine.ls_min and sk > 0:
                import warnings
                warnings.warn("fp16 loss scale is at its minimum (%g) and %d updates were skipped in epoch %d: the "
                              "activations overflow fp16; training is stalled -- use dtype='bf16'"
                              % (st['scale'], sk, epoch), RuntimeWarning)

    # ---- sampling helpers (pix2pix.py:276-425): forward-only, PNG output through util.imsave (PIL) --------
    @staticmethod
    def _next(itr):
        return next(itr) if hasattr(itr, '__next__') else itr.next()

    def generate_atob(self, itr, num_batches, out_dir, dont_predict=False, deterministic=True):
        """pix2pix samples (pix2pix.py:276-305): for every element of ``num_batches`` batches write
        ``<ctr>.a.png`` (the input A) and ``<ctr>.b.png`` (U(A), or the iterator's own B when
        ``dont_predict``)."""
        fn = self.gen_fn_det if deterministic else self.gen_fn
        makedirs(out_dir)
        ctr = 0
        for _ in range(num_batches):
            this_x, this_y = self._next(itr)
            pred_y = this_y if dont_predict else fn(this_x)
            for i in range(pred_y.shape[0]):
                imsave("%s/%i.a.png" % (out_dir, ctr), convert_to_rgb(this_x[i], is_grayscale=self.is_a_grayscale))
                imsave("%s/%i.b.png" % (out_dir, ctr), convert_to_rgb(pred_y[i], is_grayscale=self.is_b_grayscale))
                ctr += 1

    def generate_gz(self, num_examples, batch_size, out_dir, deterministic=True):
        """DCGAN samples g(z) (pix2pix.py:306-326): one draw of ``sampler(num_examples, latent_dim)``,
        ``num_examples // batch_size`` forward passes, ``<ctr>.png`` each."""
        makedirs(out_dir)
        fn = self.z_fn_det if deterministic else self.z_fn
        z = floatX(self.sampler(num_examples, self.latent_dim))
        ctr = 0
        for b in range(num_examples // batch_size):
            out = fn(z[b * batch_size:(b + 1) * batch_size])
            for i in range(out.shape[0]):
                imsave("%s/%i.png" % (out_dir, ctr), convert_to_rgb(out[i], is_grayscale=self.is_a_grayscale))
                ctr += 1

    def interpolation_grid(self, zsample1=None, zsample2=None, deterministic=True, mode='row'):
        """The decoded interpolation of generate_interpolation as an array [rows, cols, H, W, 3]:
        'row' = 1x6 with coefficients (0, .1, .3, .6, .9, 1), 'matrix' = 5x5 with linspace(0, 1, 25)
        (pix2pix.py:344-368).  All coefficients go through the generator as ONE batch (the reference
        runs 6 / 25 batch-1 calls; with deterministic BN the result per sample is the same)."""
        assert mode in ['row', 'matrix']
        fn = self.z_fn_det if deterministic else self.z_fn
        if zsample1 is None:
            zsample1 = floatX(self.sampler(1, self.latent_dim))[0]
        if zsample2 is None:
            zsample2 = floatX(self.sampler(1, self.latent_dim))[0]
        zsample1, zsample2 = np.asarray(zsample1), np.asarray(zsample2)
        if mode == 'row':
            rows, cols, coefs = 1, 6, np.asarray([0.0, 0.1, 0.3, 0.6, 0.9, 1.0], zsample1.dtype)
        else:
            rows, cols, coefs = 5, 5, np.linspace(0, 1, 25).astype(zsample1.dtype)
        z = (1 - coefs)[:, None] * zsample1[None] + coefs[:, None] * zsample2[None]
        if deterministic:
            out = fn(floatX(z))
        else:       # batch statistics would couple the samples: keep the reference's batch-1 calls
            out = np.concatenate([fn(floatX(z[i:i + 1])) for i in range(len(coefs))], axis=0)
        grid = np.zeros((rows, cols, self.in_shp, self.in_shp, 3), dtype=zsample1.dtype)
        for n in range(len(coefs)):
            grid[n // cols][n % cols] = convert_to_rgb(out[n], is_grayscale=self.is_a_grayscale)
        return grid

    def generate_interpolation(self, out_name, zsample1=None, zsample2=None, deterministic=True, mode='row',
                               figsize=(10, 10), cmap='gray'):
        """Write the interpolation grid between two prior samples as one figure (pix2pix.py:328-369)."""
        from . import image_grid
        grid = self.interpolation_grid(zsample1, zsample2, deterministic, mode)
        image_grid.write_image_grid(out_name, grid, figsize=figsize, cmap=cmap)

    def generate_interpolation_clip(self, num_samples, batch_size, out_dir, deterministic=True, min_max_norm=False,
                                    concat=False):
        """Frames of a long interpolation z1 -> z2 -> ... -> zn, 25 steps per leg, each decoded to a heightmap
        G(z) and textured by U(G(z)) (pix2pix.py:371-425).  The G -> U chain stays in HBM
        (GanStep.generate_chain); ``a_%04d.png``/``b_%04d.png`` or ``concat_%04d.png`` per frame."""
        os.makedirs(out_dir, exist_ok=True)
        zs = floatX(self.sampler(num_samples, self.latent_dim))
        coefs = np.linspace(0, 1, 25).astype(zs.dtype)
        legs = [(1 - coefs)[:, None] * zs[i][None] + coefs[:, None] * zs[i + 1][None] for i in range(len(zs) - 1)]
        all_z = np.concatenate(legs, axis=0).astype(zs.dtype) if legs else np.zeros((0, self.latent_dim), zs.dtype)
        ctr = 0
        for b in range(all_z.shape[0] // batch_size):
            z_out, p2p_out = self.engine.generate_chain(all_z[b * batch_size:(b + 1) * batch_size], deterministic)
            for i in range(z_out.shape[0]):
                a_img = z_out[i]
                if min_max_norm:
                    a_img = (a_img - np.min(a_img)) / (np.max(a_img) - np.min(a_img))
                a_img = convert_to_rgb(a_img, is_grayscale=self.is_a_grayscale)
                b_img = convert_to_rgb(p2p_out[i], is_grayscale=self.is_b_grayscale)
                d = '%04d' % ctr
                if concat:
                    imsave("%s/concat_%s.png" % (out_dir, d), np.concatenate([a_img, b_img], axis=1))
                else:
                    imsave("%s/a_%s.png" % (out_dir, d), a_img)
                    imsave("%s/b_%s.png" % (out_dir, d), b_img)
                ctr += 1

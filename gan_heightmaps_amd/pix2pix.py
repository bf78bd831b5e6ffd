"""``Pix2Pix``: the reference's trainer surface (/root/reference/pix2pix.py:19-425) on the libghm backend.

Same constructor arguments, attributes (train_fn, loss_fn, gen_fn, gen_fn_det, z_fn, z_fn_det, dcgan, p2p, lr,
train_keys) and methods (save_model, load_model, train, generate_*); compiled functions take numpy float32
arrays and return lists of numpy scalars / arrays, synchronously.
Extra keyword arguments (device, comm, use_graph, seed) configure the MI355X backend.
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

floatX = _init.floatX


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
                 device=None, comm=None, use_graph=True, seed=None, two_streams=True, force_exchange=False):
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
        self.device = device if device is not None else Device(0)
        self.engine = GanStep(self.device, dcgan_gen, dcgan_disc, p2p_gen, p2p_disc, alpha, lsgan, reconstruction,
                              spec, train_mode, comm=comm, use_graph=use_graph, two_streams=two_streams,
                              force_exchange=force_exchange)
        self.train_keys = list(TRAIN_KEYS)
        eng = self.engine
        self.train_fn = lambda Z, X, Y: eng.train(floatX(Z), floatX(X), floatX(Y))
        self.loss_fn = lambda Z, X, Y: eng.loss(floatX(Z), floatX(X), floatX(Y))
        self.gen_fn = lambda X: eng.generate('p2p_gen', X, False)
        self.gen_fn_det = lambda X: eng.generate('p2p_gen', X, True)
        self.z_fn = lambda Z: eng.generate('dcgan_gen', Z, False)
        self.z_fn_det = lambda Z: eng.generate('dcgan_gen', Z, True)

    # ---- checkpoint (pix2pix.py:158-186): gzip + pickle of get_all_param_values per net -------------------
    def save_model(self, filename):
        with gzip.open(filename, "wb") as g:
            pickle.dump({
                'dcgan': {'gen': L.get_all_param_values(self.dcgan['gen']),
                          'disc': L.get_all_param_values(self.dcgan['disc'])},
                'p2p': {'gen': L.get_all_param_values(self.p2p['gen']),
                        'disc': L.get_all_param_values(self.p2p['disc'])}
            }, g, 2)        # protocol 2 == py2 HIGHEST_PROTOCOL, readable by the reference

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

    # ---- training loop (pix2pix.py:187-275) ----------------------------------------------------------------
    def train(self, it_train, it_val, batch_size, num_epochs, out_dir, model_dir=None, save_every=10, resume=False,
              quick_run=False, validate_on_train_iterator=True, dump_images=False):
        """Same loop as the reference: per epoch N//batch_size train_fn steps then N//batch_size loss_fn steps,
        a CSV row of epoch means, periodic checkpoints.  The reference's validation loop draws its batches from
        ``it_train`` (pix2pix.py:204); ``validate_on_train_iterator=True`` keeps that behaviour."""
        def _next(it):
            return next(it) if hasattr(it, '__next__') else it.next()

        def _loop(fn, itr, src):
            rec = [[] for _ in self.train_keys]
            on_device = hasattr(src, 'next_into')          # gan_heightmaps_amd.data.Hdf5Iterator: batch stays in HBM
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
        os.makedirs(out_dir, exist_ok=True)
        if model_dir is not None:
            os.makedirs(model_dir, exist_ok=True)
        f = open("%s/results.txt" % out_dir, "w" if not resume else "a")
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
            if dump_images:
                if self.train_mode in ['both', 'p2p']:
                    self.generate_atob(it_train, 1, "%s/dump_train" % out_dir, deterministic=False)
                    self.generate_atob(it_val, 1, "%s/dump_valid" % out_dir, deterministic=False)
                if self.train_mode in ['both', 'dcgan']:
                    self.generate_gz(num_examples=20, batch_size=batch_size, out_dir="%s/dump_a" % out_dir,
                                     deterministic=False)
            if model_dir is not None and (e + 1) % save_every == 0:
                self.save_model("%s/%i.model" % (model_dir, e + 1))
        f.close()

    # ---- sampling helpers (pix2pix.py:276-326), PNG output through PIL when available ----------------------
    @staticmethod
    def _save_png(path, img_chw_01):
        try:
            from PIL import Image
        except ImportError:         # pragma: no cover
            np.save(path + ".npy", img_chw_01)
            return
        a = np.clip(img_chw_01, 0, 1)
        a = (a.transpose(1, 2, 0) * 255).astype(np.uint8)
        if a.shape[2] == 1:
            a = a[:, :, 0]
        Image.fromarray(a).save(path)

    @staticmethod
    def _to_01(img, is_grayscale):
        """util.convert_to_rgb range handling: grayscale is already [0,1], colour is [-1,1]"""
        return img if is_grayscale else (img + 1.0) / 2.0

    def generate_atob(self, itr, num_batches, out_dir, deterministic=False):
        os.makedirs(out_dir, exist_ok=True)
        fn = self.gen_fn_det if deterministic else self.gen_fn
        ctr = 0
        for _ in range(num_batches):
            this_x, this_y = next(itr) if hasattr(itr, '__next__') else itr.next()
            pred = fn(this_x)
            for i in range(pred.shape[0]):
                a = self._to_01(this_x[i], self.is_a_grayscale)
                b = self._to_01(pred[i], self.is_b_grayscale)
                if a.shape[0] != b.shape[0]:
                    a = np.repeat(a, 3, axis=0) if a.shape[0] == 1 else a
                    b = np.repeat(b, 3, axis=0) if b.shape[0] == 1 else b
                self._save_png("%s/%i.png" % (out_dir, ctr), np.concatenate([a, b], axis=2))
                ctr += 1

    def generate_gz(self, num_examples, batch_size, out_dir, deterministic=False):
        os.makedirs(out_dir, exist_ok=True)
        fn = self.z_fn_det if deterministic else self.z_fn
        ctr = 0
        for _ in range(max(num_examples // batch_size, 1)):
            z = floatX(self.sampler(batch_size, self.latent_dim))
            out = fn(z)
            for i in range(out.shape[0]):
                self._save_png("%s/%i.png" % (out_dir, ctr), self._to_01(out[i], self.is_a_grayscale))
                ctr += 1

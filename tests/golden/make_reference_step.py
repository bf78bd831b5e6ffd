#!/usr/bin/env python
"""Executes the REFERENCE's Pix2Pix.__init__ (pix2pix.py:24-157, loaded through lib2to3) -- the definition of
train_fn / loss_fn / gen_fn* / z_fn* -- on top of tests/golden/symtheano.py, and records what its compiled
functions return and do to the parameters:

    python tests/golden/make_reference_step.py        # build container only: needs /root/reference
    -> tests/golden/reference_step.npz

Reference code executed: the whole of __init__ (which outputs feed which loss, targets, alpha * recon, which
parameters every loss updates, `updates` order, the five outputs) and architectures/dcgan.py's generator and
discriminator.  The pix2pix nets come from this package's architectures/p2p.py here because the reference's
g_unet asserts in_shp == 512 (p2p.py:137) and this fixture is CPU-sized; their graph is pinned on the reference at
full size by reference_graph.json.  Not reference code: the arithmetic of every op, BN's update rule and the
optimiser formulas (oracle/ops.py via symtheano) -- numerics of Theano itself stay unpinned.

The test (tests/test_reference_step.py) runs oracle/step.py -- the oracle's own, independently written wiring --
on the same seed and inputs and must land on the same losses, outputs and post-step parameters.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

CFG_OVER = dict(in_shp=32, latent_dim=24, gen_dcgan=dict(nch=32, div=[2, 2, 4]), disc_dcgan=dict(nch=32, div=[4, 2, 2]),
                gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))
SEED, BATCH = 5, 3
VARIANTS = {"rmsprop_both": dict(opt="rmsprop", train_mode="both", lsgan=True, reconstruction="l1"),
            "adam_p2p_l2_bce": dict(opt="adam", train_mode="p2p", lsgan=False, reconstruction="l2",
                                    disc_dcgan=dict(nch=32, div=[4, 2, 2], nonlinearity="sigmoid"),
                                    disc_p2p=dict(nf=4, mul_factor=[1, 2], act="sigmoid")),
            "rmsprop_dcgan": dict(opt="rmsprop", train_mode="dcgan", lsgan=True, reconstruction="l1")}


def load_reference(sym):
    import make_reference_graph as G
    import make_reference_iterator as I
    import make_reference_trainloop as T
    G.install_shims({})
    th = sys.modules["theano"]
    th.function = sym.function
    th.tensor.fmatrix = sym.placeholder
    th.tensor.tensor4 = sym.placeholder
    th.tensor.abs_ = sym.abs_
    sys.modules["lasagne.layers"].get_output = sym.get_output
    sys.modules["lasagne.objectives"].squared_error = sym.squared_error
    sys.modules["lasagne.objectives"].binary_crossentropy = sym.binary_crossentropy
    sys.modules["lasagne.updates"].rmsprop = sym.rmsprop
    sys.modules["lasagne.updates"].adam = sym.adam
    sys.modules["util"] = I.load_reference_util()
    sys.modules["keras_ports"] = types.ModuleType("keras_ports")
    sys.modules["keras_ports"].ReduceLROnPlateau = object
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "architectures"))
    P = T.load_py2(os.path.join(REF, "pix2pix.py"), "reference_pix2pix")
    import importlib
    ref_dcgan = importlib.import_module("architectures.dcgan")
    return P, ref_dcgan


def build(P, ref_dcgan, sym, cfg):
    from gan_heightmaps_amd import init as INIT
    from gan_heightmaps_amd import nonlinearities as NL
    from gan_heightmaps_amd import updates as UP
    from gan_heightmaps_amd.architectures import p2p
    nl = {'linear': NL.linear, 'tanh': NL.tanh, 'sigmoid': NL.sigmoid}
    g, d, u, p = cfg['gen_dcgan'], cfg['disc_dcgan'], cfg['gen_p2p'], cfg['disc_p2p']
    INIT.set_rng(np.random.RandomState(SEED))
    return P.Pix2Pix(
        gen_fn_dcgan=ref_dcgan.default_generator, disc_fn_dcgan=ref_dcgan.default_discriminator,
        gen_params_dcgan=dict(nch=g['nch'], h=g['h'], initial_size=g['initial_size'], final_size=cfg['in_shp'],
                              div=g['div'], bilinear_upsample=g['bilinear_upsample']),
        disc_params_dcgan=dict(nch=d['nch'], h=d['h'], div=d['div'], bn=d['bn'], nonlinearity=nl[d['nonlinearity']],
                               pool_mode=d['pool_mode']),
        gen_fn_p2p=p2p.g_unet, disc_fn_p2p=p2p.discriminator,
        gen_params_p2p=dict(nf=u['nf'], act=nl[u['act']], bilinear_upsample=u['bilinear_upsample']),
        disc_params_p2p=dict(nf=p['nf'], bn=p['bn'], act=nl[p['act']], mul_factor=p['mul_factor']),
        in_shp=cfg['in_shp'], latent_dim=cfg['latent_dim'], is_a_grayscale=cfg['is_a_grayscale'],
        is_b_grayscale=cfg['is_b_grayscale'], alpha=cfg['alpha'], lsgan=cfg['lsgan'],
        reconstruction=cfg['reconstruction'], opt=sym.rmsprop if cfg['opt'] == 'rmsprop' else sym.adam,
        opt_args={'learning_rate': UP.shared(np.float32(cfg['lr']))}, train_mode=cfg['train_mode'], verbose=False)


def all_values(m):
    from gan_heightmaps_amd import layers as L
    return {"%s/%s/%03d" % (a, b, i): v for a in ("dcgan", "p2p") for b in ("gen", "disc")
            for i, v in enumerate(L.get_all_param_values(getattr(m, a)[b]))}


def main():
    import symtheano as sym
    from oracle import step as S
    P, ref_dcgan = load_reference(sym)
    out = {}
    for name, over in VARIANTS.items():
        cfg = S.default_cfg(**dict(CFG_OVER, **over))
        m = build(P, ref_dcgan, sym, cfg)
        for step in range(2):
            Z, X, Y = S.synthetic_batch(BATCH, cfg, seed=100 + step)
            out["%s/train%d" % (name, step)] = np.asarray(m.train_fn(Z, X, Y), np.float64)
            if step == 0:
                for k, v in all_values(m).items():
                    out["%s/after0/%s" % (name, k)] = v
        Z, X, Y = S.synthetic_batch(BATCH, cfg, seed=200)
        out[name + "/loss"] = np.asarray(m.loss_fn(Z, X, Y), np.float64)
        out[name + "/z_fn_det"] = np.asarray(m.z_fn_det(Z), np.float64)
        out[name + "/gen_fn_det"] = np.asarray(m.gen_fn_det(X), np.float64)
        out[name + "/z_fn"] = np.asarray(m.z_fn(Z), np.float64)
        out[name + "/gen_fn"] = np.asarray(m.gen_fn(X), np.float64)
        for k, v in all_values(m).items():
            out["%s/final/%s" % (name, k)] = v
        print(name, out[name + "/train0"], out[name + "/train1"], out[name + "/loss"])
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(HERE, "reference_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

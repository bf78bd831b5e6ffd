#!/usr/bin/env python
"""Writes a checkpoint with the REFERENCE's own Pix2Pix.save_model (pix2pix.py:158-166, executed via the in-memory
lib2to3 load of make_reference_trainloop.py) for four small networks built from this package's architecture files.

    python tests/golden/make_reference_checkpoint.py        # build container only: needs /root/reference
    -> tests/golden/reference_checkpoint.model   (gzip + pickle, as the reference writes it)
       tests/golden/reference_checkpoint.npz     (the same parameter values, plainly, to compare after loading)

Python 3's pickle.HIGHEST_PROTOCOL (5) is what the executed code picks here; a Python-2 run of the reference writes
protocol 2 with byte strings -- Pix2Pix.load_model reads both (encoding='latin1').
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

SMALL = dict(in_shp=32, latent_dim=12,
             gen_dcgan=dict(nch=16, div=[2, 2, 4], initial_size=4), disc_dcgan=dict(nch=16, div=[4, 2, 2]),
             gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))


def build_nets(seed):
    from gan_heightmaps_amd import init as INIT
    from gan_heightmaps_amd.architectures import dcgan, p2p
    from gan_heightmaps_amd.nonlinearities import linear, tanh
    INIT.set_rng(np.random.RandomState(seed))
    s = SMALL
    g = dcgan.default_generator(s["latent_dim"], True, num_repeats=0, final_size=s["in_shp"], **s["gen_dcgan"])
    d = dcgan.default_discriminator(s["in_shp"], True, num_repeats=0, bn=False, nonlinearity=linear, **s["disc_dcgan"])
    u = p2p.g_unet(s["in_shp"], True, False, act=tanh, num_repeats=0, bilinear_upsample=True, **s["gen_p2p"])
    pd = p2p.discriminator(s["in_shp"], True, False, bn=False, num_repeats=0, act=linear, **s["disc_p2p"])
    return {"dcgan": {"gen": g, "disc": d}, "p2p": {"gen": u, "disc": pd["out"]}}


def load_reference_pix2pix():
    import make_reference_graph as G
    import make_reference_iterator as I
    import make_reference_trainloop as T
    G.install_shims({})
    sys.modules["util"] = I.load_reference_util()
    sys.modules["keras_ports"] = types.ModuleType("keras_ports")
    sys.modules["keras_ports"].ReduceLROnPlateau = object
    sys.path.insert(0, REF)
    return T.load_py2(os.path.join(REF, "pix2pix.py"), "reference_pix2pix")


def main():
    from gan_heightmaps_amd import layers as L
    P = load_reference_pix2pix()
    nets = build_nets(123)
    m = P.Pix2Pix.__new__(P.Pix2Pix)
    m.dcgan, m.p2p = nets["dcgan"], nets["p2p"]
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(HERE, "reference_checkpoint.model")
    m.save_model(path)
    flat = {}
    for a in ("dcgan", "p2p"):
        for b in ("gen", "disc"):
            for i, v in enumerate(L.get_all_param_values(nets[a][b])):
                flat["%s/%s/%03d" % (a, b, i)] = v
    np.savez_compressed(os.path.splitext(path)[0] + ".npz", **flat)
    print("wrote", path, os.path.getsize(path), "bytes,", len(flat), "tensors")


if __name__ == "__main__":
    main()

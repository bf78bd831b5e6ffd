#!/usr/bin/env python
"""How far apart are the fp32 MFMA mode and the split-fp32 mode on the real 512x512 nets, per data seed?  (no oracle: GPU only)
    python tools/mode_agreement.py [batch] [seed ...]
Prints rel-L2 of G(z), U(X) and of every net's gradients between the two modes after ONE train_fn call from the same parameters.
A figure far above 1e-6 on one seed and not on the others is a near-tie of the nets flipped by a 1e-7 rounding difference (the
small-batch BatchNorm chains of the generators amplify it), not an arithmetic error."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import step as ostep
from gan_heightmaps_amd import device, layers as L
from gan_heightmaps_amd.experiments import make_model


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seeds = [int(s) for s in sys.argv[2:]] or [9, 10, 11, 12]
dev = device.Device(0)
cfg = ostep.default_cfg()
nets = [('dcgan', 'gen', 'dcgan_gen'), ('dcgan', 'disc', 'dcgan_disc'), ('p2p', 'gen', 'p2p_gen'), ('p2p', 'disc', 'p2p_disc')]
for seed in seeds:
    Z, X, Y = ostep.synthetic_batch(B, cfg, seed=seed)
    res = {}
    for dt in ('f32', 'bf16x3'):
        m = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False, dtype=dt)
        gz = m.gen_fn_det(X)[0] if False else None
        losses = m.train_fn(Z, X, Y)
        g = {k: np.concatenate([m.engine.stores[k].download_grad(p).ravel() for p in L.get_all_params(getattr(m, a)[b], trainable=True)])
             for a, b, k in nets}
        res[dt] = (np.asarray(losses), g)
        del m
    print("batch %d data seed %d: losses %.1e  " % (B, seed, rel(res['bf16x3'][0], res['f32'][0])) +
          "  ".join("%s %.2e" % (k, rel(res['bf16x3'][1][k], res['f32'][1][k])) for _, _, k in nets), flush=True)

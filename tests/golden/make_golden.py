"""Generates the committed golden vectors under tests/golden/ from the float64 numpy oracle.

    python tests/golden/make_golden.py

Provenance: the reference (Python-2 Theano/Lasagne) cannot run in this environment and holds no golden vectors
of its own, so these are produced by oracle/ after it has been cross-checked against torch-CPU
autograd (tests/test_oracle_vs_torch.py) and the reference-held structural known answers
(tests/test_known_answers.py).  They pin the oracle against silent drift and give the GPU tests fixed targets.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ops as O  # noqa: E402
from oracle import step as S  # noqa: E402

SMALL = dict(in_shp=32, latent_dim=24,
             gen_dcgan=dict(nch=16, div=[2, 2, 4]), disc_dcgan=dict(nch=16, div=[4, 2, 2]),
             gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))

OP_CASES = {
    # name: (N, C, H, W, K, k, stride, pad)
    "conv5_same": (2, 6, 9, 9, 8, 5, 1, 2),
    "conv3_s2": (2, 5, 8, 8, 7, 3, 2, 1),
    "conv3_s1": (1, 4, 7, 6, 3, 3, 1, 1),
    "conv2_valid": (3, 8, 2, 2, 6, 2, 1, 0),
}
DECONV_CASES = {"deconv_k2s1": (2, 6, 1, 1, 5, 2, 1), "deconv_k2s2": (2, 5, 4, 4, 3, 2, 2)}


def make_ops():
    rng = np.random.RandomState(2024)
    out = {}
    for name, (N, C, H, W, K, k, s, pad) in OP_CASES.items():
        x, Wt, b = rng.randn(N, C, H, W), rng.randn(K, C, k, k) * 0.3, rng.randn(K)
        y = O.conv2d_fwd(x, Wt, b, s, pad)
        dy = rng.randn(*y.shape)
        dx, dW, db = O.conv2d_vjp(x, Wt, dy, s, pad)
        for k_, v in dict(x=x, W=Wt, b=b, y=y, dy=dy, dx=dx, dW=dW, db=db).items():
            out["%s.%s" % (name, k_)] = v.astype(np.float32)
    for name, (N, Ci, h, w, Co, k, s) in DECONV_CASES.items():
        x, Wt, b = rng.randn(N, Ci, h, w), rng.randn(Ci, Co, k, k) * 0.3, rng.randn(Co)
        y = O.deconv2d_fwd(x, Wt, b, s)
        dy = rng.randn(*y.shape)
        dx, dW, db = O.deconv2d_vjp(x, Wt, dy, s)
        for k_, v in dict(x=x, W=Wt, b=b, y=y, dy=dy, dx=dx, dW=dW, db=db).items():
            out["%s.%s" % (name, k_)] = v.astype(np.float32)
    # batchnorm incl. the [4, C, 1, 1] bottleneck case
    for name, shape in {"bn_4d": (4, 6, 5, 5), "bn_bottleneck": (4, 16, 1, 1), "bn_dense": (4, 12)}.items():
        x, beta, gamma = rng.randn(*shape) * 1.5 + 0.3, rng.randn(shape[1]), rng.rand(shape[1]) + 0.5
        y, mu, inv = O.bn_train_fwd(x, beta, gamma)
        dy = rng.randn(*shape)
        dx, dbeta, dgamma = O.bn_train_vjp(x, gamma, mu, inv, dy)
        rm, ri = O.bn_running_update(np.zeros(shape[1]), np.ones(shape[1]), mu, inv)
        for k_, v in dict(x=x, beta=beta, gamma=gamma, y=y, mu=mu, inv=inv, dy=dy, dx=dx, dbeta=dbeta, dgamma=dgamma,
                          run_mean=rm, run_inv=ri).items():
            out["%s.%s" % (name, k_)] = v.astype(np.float32)
    # theano bilinear x2, odd sizes included, from the LITERAL algorithm
    for name, hw in {"bilinear_5x3": (5, 3), "bilinear_2x2": (2, 2), "bilinear_1x1": (1, 1)}.items():
        x = rng.randn(2, 3, *hw)
        out[name + ".x"] = x.astype(np.float32)
        out[name + ".y"] = O.bilinear_theano_literal(x, 2).astype(np.float32)
        g = rng.randn(2, 3, 2 * hw[0], 2 * hw[1])
        out[name + ".g"] = g.astype(np.float32)
        out[name + ".dx"] = O.bilinear_up2_vjp(g).astype(np.float32)
    return out


def make_step(opt, lr):
    cfg = S.default_cfg(opt=opt, lr=lr, **SMALL)
    st = S.init_state(cfg, seed=7, dtype=np.float32)
    rec = {"losses": [], "grad_norms": [], "param_sums": [], "param_abs_sums": []}
    for it in range(3):
        Z, X, Y = S.synthetic_batch(4, cfg, seed=100 + it)
        # the device resyncs nothing: pure float64 oracle trajectory with float32 parameter storage
        res = S.train_step(st, Z, X, Y, dtype=np.float64)
        rec["losses"].append(res['losses'])
        rec["grad_norms"].append([np.sqrt(sum(float((g ** 2).sum()) for g in res['grads'][k])) for k in S.NET_ORDER])
        rec["param_sums"].append([sum(float(np.asarray(p, np.float64).sum()) for p in st['params'][k[0]][k[1]])
                                  for k in S.NET_ORDER])
        rec["param_abs_sums"].append([sum(float(np.abs(np.asarray(p, np.float64)).sum())
                                          for p in st['params'][k[0]][k[1]]) for k in S.NET_ORDER])
    return {k: np.asarray(v, np.float64) for k, v in rec.items()}


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **make_ops())
    np.savez_compressed(os.path.join(HERE, "step_rmsprop.npz"), **make_step('rmsprop', 1e-4))
    np.savez_compressed(os.path.join(HERE, "step_adam.npz"), **make_step('adam', 1e-3))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))

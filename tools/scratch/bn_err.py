import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D
dev = D.Device(0); ops = D.Ops(dev)
rng = np.random.RandomState(0)
for (N, C, H, W, mu, sd) in [(16, 16, 64, 64, 0, 1), (16, 16, 64, 64, 5, 0.1), (16, 32, 8, 8, 0, 1), (16, 32, 16, 16, 1, 1), (4, 64, 256, 256, 0.3, 1), (16, 1024, 1, 1, 0, 1), (3, 5, 6, 6, 0, 1)]:
    x = (rng.randn(N, C, H, W) * sd + mu).astype(np.float32)
    g = rng.randn(N, C, H, W).astype(np.float32)
    X = dev.tensor(x); G = dev.tensor(g)
    vec = lambda a: dev.tensor(np.asarray(a, np.float32).reshape(1, C, 1, 1))
    m, iv = dev.empty((1, C, 1, 1)), dev.empty((1, C, 1, 1))
    ws = dev.alloc(ops.bn_workspace(C))
    ops.bn_stats(X, m, iv, ws)
    x64 = x.astype(np.float64)
    mr = x64.mean((0, 2, 3)); vr = x64.var((0, 2, 3)); ir = 1 / np.sqrt(vr + 1e-4)
    em = np.abs(m.numpy().ravel() - mr).max() / (np.abs(mr).max() + 1e-30); ei = np.abs(iv.numpy().ravel() / ir - 1).max()
    gam = rng.rand(C) + 0.5; bet = rng.randn(C)
    Y = dev.empty((N, C, H, W)); ops.bn_apply(X, Y, m, iv, vec(gam), vec(bet), 'linear')
    dx = dev.empty((N, C, H, W)); dg, db = dev.empty((1, C, 1, 1)), dev.empty((1, C, 1, 1))
    ops.bn_backward(G, Y, X, dx, m, iv, vec(gam), dg, db, ws, 'linear')
    xh = (x64 - mr[None, :, None, None]) * ir[None, :, None, None]
    g64 = g.astype(np.float64)
    dbr = g64.sum((0, 2, 3)); dgr = (g64 * xh).sum((0, 2, 3))
    cnt = N * H * W
    dxr = (gam * ir)[None, :, None, None] * (g64 - dbr[None, :, None, None] / cnt - xh * dgr[None, :, None, None] / cnt)
    rel = lambda a, b: np.linalg.norm(a.ravel() - b.ravel()) / np.linalg.norm(b.ravel())
    print((N, C, H, W, mu, sd), "mean %.1e inv %.1e  dbeta %.1e dgamma %.1e dx %.1e" % (em, ei, rel(db.numpy(), dbr), rel(dg.numpy(), dgr), rel(dx.numpy(), dxr)))

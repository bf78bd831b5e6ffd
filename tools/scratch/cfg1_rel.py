import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_gpu_step as T
from oracle import step as ostep
from gan_heightmaps_amd import device as D
dev = D.Device(0)
over = dict(in_shp=64, latent_dim=100, train_mode='dcgan',
            gen_dcgan=dict(nch=64, div=[2, 2, 4, 4]), disc_dcgan=dict(nch=64, div=[8, 4, 2, 1]),
            gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))
cfg = ostep.default_cfg(**over)
B, seed = 16, 7
model = T.build_model(cfg, seed, dev)
state = ostep.init_state(cfg, seed, np.float32)
Z, X, Y = ostep.synthetic_batch(B, cfg, seed=100)
ref = ostep.train_step(state, Z, X, Y, dtype=np.float64)
got = model.train_fn(Z, X, Y)
mg = T.model_grads(model)
for key in ref['grads']:
    for i, (g, r) in enumerate(zip(mg[key], ref['grads'][key])):
        print(key, i, g.shape, "norm %.3e rel %.2e" % (np.linalg.norm(r), T.rel(g.ravel(), r.ravel())))

"""The DCGAN generator's gradient as a per-sample multiple of the discriminator-loss pass in fp16, in confident-
discriminator states: shortcut (GHM_RANK_ONE_F16=1) against the two separate passes, both against the float64 oracle.
    gpurun -- python tools/f16_rank_one_probe.py"""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import step as ostep
from tests.test_gpu_step import SMALL, build_model, model_grads, rel, _confident_discriminator_state
from tests.test_gpu_lp import LP_STEP
from gan_heightmaps_amd import device, layers as L
dev = device.Device(0)
cfg = ostep.default_cfg(**LP_STEP)      # 128 x 128 nets in which every low-precision kernel family runs
Z, X, Y = ostep.synthetic_batch(4, cfg, seed=310)
for target in (1e-2, 1e-4, 1e-6):
    st, d = _confident_discriminator_state(cfg, 7, Z, X, Y, target)
    ref = ostep.train_step(ostep.clone_state(st), Z, X, Y, dtype=np.float64)
    gref = np.concatenate([g.ravel() for g in ref['grads'][('dcgan', 'gen')]])
    out = {}
    for name, env in (("shortcut", {"GHM_RANK_ONE_F16": "1"}), ("two passes", {})):
        for k in ("GHM_RANK_ONE_F16",):
            os.environ.pop(k, None)
        os.environ.update(env)
        m = build_model(cfg, 7, dev, dtype='f16')
        L.set_all_param_values(m.dcgan['disc'], st['params']['dcgan']['disc'])
        m.train_fn(Z, X, Y)
        g = np.concatenate([x.ravel() for x in model_grads(m)[('dcgan', 'gen')]]) / m.engine.loss_scale
        out[name] = (rel(g, gref), float(np.mean(g == 0)), "per_sample_ratio" in [e[0] for e in m.engine.built(4).train_compute[0]])
    print("D(G(z)) ~ %g: |gref| %.3e" % (target, np.linalg.norm(gref)), out)

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D
from gan_heightmaps_amd.experiments import make_model
dev = D.Device(0)
rng = np.random.RandomState(1)
Z = rng.rand(4, 1000).astype(np.float32); X = rng.rand(4, 1, 512, 512).astype(np.float32)
Y = (rng.rand(4, 3, 512, 512) * 2 - 1).astype(np.float32)
for name in ('test1_nobn', 'test1_nobn_finetunep2p_bilin', 'test1_nobn_bilin_both'):
    m = make_model(name, device=dev, seed=0, verbose=False, use_graph=False)
    l = m.train_fn(Z, X, Y)
    eng = m.engine; b = eng.built(4)
    for _ in range(3): eng.enqueue_train(b)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(10): eng.enqueue_train(b)
    eng.sync(); dt = (time.perf_counter() - t0) / 10
    print("%-32s mode=%-5s losses=%s  %.2f ms/step %.1f img/s" % (name, m.train_mode, np.round(l, 4), dt * 1e3, 4 / dt))
    g = m.gen_fn_det(X); z = m.z_fn_det(Z)
    assert np.isfinite(g).all() and np.isfinite(z).all() and g.shape == (4, 3, 512, 512) and z.shape == (4, 1, 512, 512)
    del m

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D, experiments as E
from gan_heightmaps_amd.pix2pix import Pix2Pix
dev = D.Device(0)
kw = E.experiment_kwargs('test1_nobn_bilin_both')
kw.update(in_shp=1024, device=dev, seed=0, verbose=False, use_graph=False)
kw['gen_params_dcgan'] = {'num_repeats': 0, 'div': [2, 2, 4, 4, 8, 8, 8, 8], 'final_size': 1024}
kw['disc_params_dcgan'] = dict(kw['disc_params_dcgan'], div=[8, 8, 4, 4, 4, 2, 2, 2], nch=1024)
try:
    m = Pix2Pix(**kw)
except Exception as e:
    print("construction failed:", repr(e)); raise
B = 4
rng = np.random.RandomState(0)
Z = rng.rand(B, 1000).astype(np.float32); X = rng.rand(B, 1, 1024, 1024).astype(np.float32)
Y = (rng.rand(B, 3, 1024, 1024) * 2 - 1).astype(np.float32)
l = m.train_fn(Z, X, Y); print("losses", l)
eng = m.engine; b = eng.built(B)
for _ in range(2): eng.enqueue_train(b)
eng.sync(); t0 = time.perf_counter()
for _ in range(5): eng.enqueue_train(b)
eng.sync(); dt = (time.perf_counter() - t0) / 5
print("1024x1024 B=4 fp32: %.2f ms/step %.1f img/s" % (dt * 1e3, B / dt))

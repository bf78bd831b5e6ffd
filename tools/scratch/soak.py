import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D
from gan_heightmaps_amd.experiments import make_model, get_iterators
dev = D.Device(0)
m = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False, use_graph=False)
it_train, it_val = get_iterators(None, 4, True, False, True, in_shp=512, n_synthetic=16, device=dev)
eng = m.engine
t0 = time.perf_counter(); hist = []
for s in range(400):
    r = eng.run_from_iterator(it_train, lambda n: np.random.rand(n, 1000).astype(np.float32), train=True)
    hist.append([float(v) for v in r])
    if s % 100 == 99:
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        print("steps %3d-%3d: %.2f ms/step  losses %s" % (s - 99, s, dt * 10, np.round(hist[-1], 4)), flush=True)
h = np.array(hist)
assert np.isfinite(h).all()
print("all finite; disc losses moved from", np.round(h[0], 3), "to", np.round(h[-1], 3))

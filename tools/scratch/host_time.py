import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D
from gan_heightmaps_amd.experiments import make_model
dev = D.Device(0)
for graph in (False, True):
    m = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False, use_graph=graph)
    eng = m.engine
    rng = np.random.RandomState(1)
    Z = rng.rand(4, 1000).astype(np.float32); X = rng.rand(4, 1, 512, 512).astype(np.float32)
    Y = (rng.rand(4, 3, 512, 512) * 2 - 1).astype(np.float32)
    b = eng.built(4); eng._upload(b, Z, X, Y)
    for _ in range(3): eng.enqueue_train(b)
    eng.sync()
    t0 = time.perf_counter(); 
    for _ in range(10): eng.enqueue_train(b)
    t1 = time.perf_counter(); eng.sync(); 
    for d in eng.devs: d.sync()
    t2 = time.perf_counter()
    print("graph=%s: host enqueue %.2f ms/step, wall %.2f ms/step" % (graph, (t1-t0)*100, (t2-t0)*100))
    del m

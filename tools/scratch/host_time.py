import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D
from gan_heightmaps_amd.experiments import make_model
from oracle import step as S
dev = D.Device(0)
for graph in (False, True):
    m = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False, use_graph=graph)
    eng = m.engine
    cfg = S.default_cfg(); Z, X, Y = S.synthetic_batch(4, cfg, seed=1)
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

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device as D
dev = D.Device(0); ops = D.Ops(dev)
x = dev.empty((8, 64, 512, 512)); y = dev.empty((8, 64, 512, 512))
nb = 8*64*512*512*4
def t(fn, reps=20):
    for _ in range(3): fn()
    dev.sync(); dev.timer_start(0)
    for _ in range(reps): fn()
    dev.timer_stop(0); return dev.timer_ms(0)/reps
ms = t(lambda: dev.memset_zero(y.ptr, nb)); print("memset 537MB   %.3f ms  %.2f TB/s write" % (ms, nb/ms/1e9))
ms = t(lambda: ops.act_fwd(x, y, 'lrelu', 0.2)); print("act_fwd r+w    %.3f ms  %.2f TB/s total" % (ms, 2*nb/ms/1e9))
ms = t(lambda: dev.d2d(y.ptr, x.ptr, nb)); print("d2d copy       %.3f ms  %.2f TB/s total" % (ms, 2*nb/ms/1e9))
s = dev.empty((1, 64, 1, 1))
ms = t(lambda: ops.channel_sum(x, s)); print("channel_sum r  %.3f ms  %.2f TB/s read" % (ms, nb/ms/1e9))

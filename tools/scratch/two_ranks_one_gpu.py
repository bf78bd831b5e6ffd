"""Can RCCL form a 2-rank communicator with both ranks on the one visible GPU?  (functional N=2 check only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gan_heightmaps_amd import device, dist
rank, local_rank, world = dist.env_rank_world()
dev = device.Device(0)
comm = dist.Comm(dev, rank, world)
x = dev.tensor(np.full((1, 8, 1, 1), rank + 1, np.float32))
from gan_heightmaps_amd._lib import call
import ctypes as C
call("ghm_allreduce_sum", dev.h, C.c_void_p(x.ptr), 8)
dev.sync()
print("rank", rank, "sum", x.numpy().ravel()[:2], flush=True)

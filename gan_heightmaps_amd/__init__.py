"""gan-heightmaps on MI355X: the DCGAN + pix2pix train step behind the reference's callable surface,
executed by hand-written gfx950 HIP kernels in libghm.so (include/ghm.h)."""
__all__ = ["device", "_lib"]

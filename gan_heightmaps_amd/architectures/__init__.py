"""Architecture definitions with the reference's callable surface
(/root/reference/architectures/{dcgan,p2p,layers}.py), written against gan_heightmaps_amd.layers."""
from . import dcgan, p2p, layers  # noqa: F401

"""The identity the DCGAN stage of the step relies on (gan_heightmaps_amd/step.py `_per_sample_scalar_head`), checked
on the float64 ORACLE alone -- no device code: with a discriminator that returns one scalar per sample and has no BatchNorm,

    d gen_loss_dcgan / d G(z)_n  ==  ( seed_G[n] / seed_D[n] ) * d disc_loss_dcgan / d G(z)_n,
    seed_X[n] = d X_loss / d D(G(z))_n

(pix2pix.py:107-108: both losses read the same D(G(z))), for the LSGAN and the BCE losses; with BatchNorm in the discriminator the
samples are coupled through the batch statistics and the identity does NOT hold (the step keeps the separate pass there)."""
import numpy as np
import pytest

from oracle import step as ostep
from oracle import tape as T

SMALL = dict(in_shp=32, latent_dim=24,
             gen_dcgan=dict(nch=16, div=[2, 2, 4]),
             disc_dcgan=dict(nch=16, div=[4, 2, 2]),
             gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))


def _both_passes(cfg, seed, B=4):
    state = ostep.init_state(cfg, seed, np.float32)
    Z, X, Y = ostep.synthetic_batch(B, cfg, seed=100)
    fw = ostep.forward(state, Z, X, Y, dtype=np.float64)
    T.backward(fw['gen_loss_dcgan'], stop_at=[fw['gz']])
    g_gen, s_gen = fw['gz'].g.copy(), fw['d_fake'].g.copy()
    T.backward(fw['disc_loss_dcgan'], stop_at=[fw['gz']])
    g_disc, s_disc = fw['gz'].g.copy(), fw['d_fake'].g.copy()
    return g_gen, s_gen.reshape(B), g_disc, s_disc.reshape(B)


@pytest.mark.parametrize("lsgan", [True, False])
def test_generator_cotangent_is_a_per_sample_multiple_of_the_discriminator_loss_cotangent(lsgan):
    over = dict(SMALL, lsgan=lsgan)
    if not lsgan:
        over.update(disc_dcgan=dict(nch=16, div=[4, 2, 2], nonlinearity='sigmoid'),
                    disc_p2p=dict(nf=4, mul_factor=[1, 2], act='sigmoid'))
    g_gen, s_gen, g_disc, s_disc = _both_passes(ostep.default_cfg(**over), seed=7)
    assert np.all(s_disc != 0) and np.linalg.norm(g_gen) > 0
    ratio = (s_gen / s_disc).reshape(-1, 1, 1, 1)
    assert np.linalg.norm(ratio * g_disc - g_gen) <= 1e-12 * np.linalg.norm(g_gen)


def test_a_dead_final_relu_zeroes_both_cotangents():
    """D's last convolution keeps lasagne's default rectify (dcgan.py:50); where it is dead at every position of a sample both
    seeds' backward passes are exactly zero -- the case `ghm_scale_samples` answers with 0 instead of 0 / 0 (seed 11: d == 0)"""
    g_gen, s_gen, g_disc, s_disc = _both_passes(ostep.default_cfg(**dict(SMALL, lsgan=True)), seed=11)
    dead = s_disc == 0
    assert dead.any()
    assert np.all(g_gen[dead] == 0) and np.all(g_disc[dead] == 0)
    live = ~dead
    if live.any():
        ratio = (s_gen[live] / s_disc[live]).reshape(-1, 1, 1, 1)
        assert np.linalg.norm(ratio * g_disc[live] - g_gen[live]) <= 1e-12 * max(np.linalg.norm(g_gen[live]), 1e-300)


def test_batchnorm_in_the_discriminator_breaks_the_identity():
    cfg = ostep.default_cfg(**dict(SMALL, disc_dcgan=dict(nch=16, div=[4, 2, 2], bn=True)))
    g_gen, s_gen, g_disc, s_disc = _both_passes(cfg, seed=7)
    ratio = (s_gen / s_disc).reshape(-1, 1, 1, 1)
    assert np.linalg.norm(ratio * g_disc - g_gen) > 1e-3 * np.linalg.norm(g_gen)

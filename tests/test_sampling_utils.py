"""f3: sampling / interpolation utilities after the path (pix2pix.py:276-425, util.py:69-116,
image_grid.py).  CPU: the host arithmetic and file naming with stand-in forward functions."""
import os

import numpy as np
import pytest

from gan_heightmaps_amd import util
from gan_heightmaps_amd.pix2pix import Pix2Pix


def test_convert_to_rgb_ranges_and_layout():
    g = np.linspace(-0.5, 1.5, 12, dtype=np.float32).reshape(1, 3, 4)
    out = util.convert_to_rgb(g, is_grayscale=True)
    assert out.shape == (3, 4, 3)
    assert np.array_equal(out[..., 0], np.clip(g[0], 0, 1)) and np.array_equal(out[..., 0], out[..., 2])
    c = np.stack([np.full((2, 2), -1.0), np.zeros((2, 2)), np.full((2, 2), 3.0)]).astype(np.float32)
    out = util.convert_to_rgb(c, is_grayscale=False)
    assert out.shape == (2, 2, 3)
    assert np.allclose(out[0, 0], [0.0, 0.5, 1.0])
    # one channel in the tanh range is also replicated then rescaled (util.py:78-83)
    out = util.convert_to_rgb(np.zeros((1, 2, 2), np.float32), is_grayscale=False)
    assert np.allclose(out, 0.5)
    with pytest.raises(Exception):
        util.convert_to_rgb(np.zeros((2, 4, 4)))
    with pytest.raises(Exception):
        util.convert_to_rgb(np.zeros((4, 4)))


def test_compose_imgs_side_by_side():
    a = np.full((1, 4, 5), 0.25, np.float32)
    b = np.zeros((3, 4, 5), np.float32)
    out = util.compose_imgs(a, b, is_a_grayscale=True, is_b_grayscale=False)
    assert out.shape == (4, 10, 3)
    assert np.allclose(out[:, :5], 0.25) and np.allclose(out[:, 5:], 0.5)
    with pytest.raises(Exception):
        util.compose_imgs(a, np.zeros((3, 4, 6), np.float32))


def test_imsave_roundtrip(tmp_path):
    img = np.random.RandomState(0).rand(8, 9, 3)
    util.imsave(str(tmp_path / "x.png"), img)
    back = util.imread(str(tmp_path / "x.png"))
    assert back.dtype == np.uint8 and back.shape == (8, 9, 3)
    assert np.array_equal(back, np.rint(img * 255).astype(np.uint8))


class _StubEngine:
    def __init__(self, m):
        self.m = m
        self.calls = []

    def generate_chain(self, Z, deterministic=True):
        self.calls.append(Z.copy())
        a = self.m.z_fn_det(Z)
        return a, self.m.gen_fn_det(a)


def _stub_model(latent=5, shp=8):
    m = Pix2Pix.__new__(Pix2Pix)
    m.latent_dim, m.in_shp = latent, shp
    m.is_a_grayscale, m.is_b_grayscale = True, False
    rs = np.random.RandomState(3)
    m.sampler = rs.rand
    proj = rs.rand(latent, shp * shp).astype(np.float32) / latent
    m.z_calls = []

    def z_fn(Z):
        m.z_calls.append(np.array(Z))
        return (np.asarray(Z, np.float32) @ proj).reshape(-1, 1, shp, shp)
    m.z_fn = m.z_fn_det = z_fn
    m.gen_fn = m.gen_fn_det = lambda X: np.concatenate([2 * X - 1, -(2 * X - 1), 0 * X], axis=1)
    m.engine = _StubEngine(m)
    return m


def test_generate_gz_counts_and_names(tmp_path):
    m = _stub_model()
    m.generate_gz(num_examples=7, batch_size=3, out_dir=str(tmp_path / "gz"))
    names = sorted(os.listdir(tmp_path / "gz"))
    assert names == ["%d.png" % i for i in range(6)]        # 7 // 3 batches of 3
    assert [c.shape[0] for c in m.z_calls] == [3, 3]
    img = util.imread(str(tmp_path / "gz" / "0.png"))
    assert img.shape == (8, 8, 3)
    assert np.array_equal(img[..., 0], util.to_uint8(m.z_fn(m.z_calls[0][:1])[0, 0]))


def test_generate_atob_pairs(tmp_path):
    m = _stub_model()
    X = np.random.RandomState(1).rand(4, 1, 8, 8).astype(np.float32)
    Y = np.random.RandomState(2).rand(4, 3, 8, 8).astype(np.float32) * 2 - 1

    def batches():
        while True:
            yield X[:2], Y[:2]
            yield X[2:], Y[2:]
    m.generate_atob(batches(), 2, str(tmp_path / "ab"))
    assert sorted(os.listdir(tmp_path / "ab")) == sorted(
        ["%d.%s.png" % (i, s) for i in range(4) for s in "ab"])
    got = util.imread(str(tmp_path / "ab" / "3.b.png"))
    assert np.array_equal(got, util.to_uint8(util.convert_to_rgb(m.gen_fn_det(X[3:4])[0], False)))
    m.generate_atob(batches(), 1, str(tmp_path / "gt"), dont_predict=True)
    got = util.imread(str(tmp_path / "gt" / "1.b.png"))
    assert np.array_equal(got, util.to_uint8(util.convert_to_rgb(Y[1], False)))


def test_interpolation_grid_coefficients():
    m = _stub_model()
    z1 = np.zeros(5, np.float32)
    z2 = np.ones(5, np.float32)
    g = m.interpolation_grid(z1, z2, mode='row')
    assert g.shape == (1, 6, 8, 8, 3)
    assert np.allclose(m.z_calls[-1][:, 0], [0.0, 0.1, 0.3, 0.6, 0.9, 1.0])
    g = m.interpolation_grid(z1, z2, mode='matrix')
    assert g.shape == (5, 5, 8, 8, 3)
    assert np.allclose(m.z_calls[-1][:, 0], np.linspace(0, 1, 25))
    # end points decode to the pure samples, row-major fill
    assert np.allclose(g[0, 0], util.convert_to_rgb(m.z_fn(z1[None])[0], True))
    assert np.allclose(g[4, 4], util.convert_to_rgb(m.z_fn(z2[None])[0], True))
    assert np.allclose(g[1, 0], util.convert_to_rgb(m.z_fn((5 / 24.0) * z2[None])[0], True), atol=1e-6)


def test_generate_interpolation_writes_figure(tmp_path):
    m = _stub_model()
    out = tmp_path / "fig" / "interp.png"
    m.generate_interpolation(str(out), mode='row', figsize=(6, 1))
    assert out.exists() and out.stat().st_size > 0


def test_interpolation_clip_frames(tmp_path):
    m = _stub_model()
    m.generate_interpolation_clip(num_samples=3, batch_size=10, out_dir=str(tmp_path / "clip"))
    # 2 legs x 25 coefficients = 50 latent vectors, 5 batches of 10
    assert len(m.engine.calls) == 5
    allz = np.concatenate(m.engine.calls)
    assert allz.shape == (50, 5)
    assert np.allclose(allz[24], allz[25])          # end of leg 1 == start of leg 2 (both are z2)
    names = sorted(os.listdir(tmp_path / "clip"))
    assert names == sorted(["%s_%04d.png" % (s, i) for i in range(50) for s in "ab"])
    m2 = _stub_model()
    m2.generate_interpolation_clip(num_samples=2, batch_size=5, out_dir=str(tmp_path / "cat"), concat=True,
                                   min_max_norm=True)
    names = sorted(os.listdir(tmp_path / "cat"))
    assert names == ["concat_%04d.png" % i for i in range(25)]
    img = util.imread(str(tmp_path / "cat" / "concat_0000.png"))
    assert img.shape == (8, 16, 3)
    assert img[:, :8].min() == 0 and img[:, :8].max() == 255         # min-max normalised heightmap half


def test_plot_grid(tmp_path):
    m = _stub_model()
    X = np.random.RandomState(1).rand(1, 1, 8, 8).astype(np.float32)
    Y = np.zeros((1, 3, 8, 8), np.float32)

    def batches():
        while True:
            yield X, Y
    out = tmp_path / "grid.png"
    util.plot_grid(str(out), batches(), m.gen_fn_det, True, False, N=2)
    assert out.exists()
    util.plot_grid(str(out), batches(), None, True, False, N=2)

"""One call of the reference's compiled ``train_fn`` / ``loss_fn`` restated in numpy
(TEST INFRASTRUCTURE ONLY; parity numerics unpinned -- see oracle/__init__.py).

Follows /root/reference/pix2pix.py:87-147:
  * one shared forward of the four nets (:92-101),
  * five losses (:102-121),
  * four gradient roots, each w.r.t. ITS OWN net's trainable params only (:123-135), all
    taken at the pre-update parameters (one merged update dict => simultaneous update),
  * lasagne rmsprop / adam per net (:131-141; choice at experiments.py:116-117),
  * BatchNorm running mean / inv_std updates fire in every non-deterministic function,
    including loss_fn (lasagne default_updates).
"""
import copy

import numpy as np

from . import nets, ops
from . import tape as T

TRAIN_KEYS = ['dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_recon', 'p2p_disc']    # pix2pix.py:157


def default_cfg(**over):
    """kwargs of experiments.py:102-119 (test1_nobn_bilin_both)."""
    cfg = dict(
        in_shp=512, latent_dim=1000, is_a_grayscale=True, is_b_grayscale=False,
        gen_dcgan=dict(nch=512, h=5, initial_size=4, div=[2, 2, 4, 4, 8, 8, 8], bilinear_upsample=False),
        disc_dcgan=dict(nch=512, h=5, div=[8, 4, 4, 4, 2, 2, 2], bn=False, nonlinearity='linear',
                        pool_mode='max'),
        gen_p2p=dict(nf=64, act='tanh', bilinear_upsample=True),
        disc_p2p=dict(nf=64, bn=False, act='linear', mul_factor=[1, 2, 4, 8]),
        alpha=100.0, lsgan=True, reconstruction='l1', opt='rmsprop', lr=1e-4, train_mode='both',
    )
    for k, v in over.items():
        if isinstance(v, dict) and k in cfg:
            cfg[k] = dict(cfg[k], **v)
        else:
            cfg[k] = v
    return cfg


def specs(cfg):
    g, d, u, p = cfg['gen_dcgan'], cfg['disc_dcgan'], cfg['gen_p2p'], cfg['disc_p2p']
    ag, bg = cfg['is_a_grayscale'], cfg['is_b_grayscale']
    return {
        ('dcgan', 'gen'): nets.dcgan_gen_spec(cfg['latent_dim'], ag, g['nch'], g['h'], g['initial_size'], g['div']),
        ('dcgan', 'disc'): nets.dcgan_disc_spec(cfg['in_shp'], ag, d['nch'], d['h'], d['div'], d['bn']),
        ('p2p', 'gen'): nets.unet_spec(cfg['in_shp'], ag, bg, u['nf'], u['bilinear_upsample']),
        ('p2p', 'disc'): nets.patchgan_spec(cfg['in_shp'], ag, bg, p['nf'], p['mul_factor'], p['bn']),
    }


NET_ORDER = [('dcgan', 'gen'), ('dcgan', 'disc'), ('p2p', 'gen'), ('p2p', 'disc')]   # pix2pix.py:73-77


def init_state(cfg, seed=0, dtype=np.float32):
    """Seeded GlorotUniform init in layer-construction order (DCGAN G, DCGAN D, U-Net,
    PatchGAN; pix2pix.py:73-77).  The reference draws from the unseeded global numpy RNG."""
    rng = np.random.RandomState(seed)
    sp = specs(cfg)
    params = {'dcgan': {}, 'p2p': {}}
    for key in NET_ORDER:
        params[key[0]][key[1]] = sp[key].init(rng, dtype)
    return {'params': params, 'opt': {}, 'cfg': cfg}


def synthetic_batch(B, cfg, seed=0, dtype=np.float32):
    """Synthetic (Z, X, Y) with the reference's value ranges: Z~U[0,1)
    (pix2pix.py:31,206), X = uint8/255 (util.py:34), Y = (uint8-127.5)/127.5 (util.py:35)."""
    H = cfg['in_shp']
    ca = 1 if cfg['is_a_grayscale'] else 3
    cb = 1 if cfg['is_b_grayscale'] else 3
    Z = np.random.RandomState(seed).rand(B, cfg['latent_dim']).astype(dtype)
    xa = np.random.RandomState(seed + 1).randint(0, 256, (B, ca, H, H)).astype(dtype)
    yb = np.random.RandomState(seed + 2).randint(0, 256, (B, cb, H, H)).astype(dtype)
    X = xa / 255.0 if cfg['is_a_grayscale'] else (xa - 127.5) / 127.5
    Y = yb / 255.0 if cfg['is_b_grayscale'] else (yb - 127.5) / 127.5
    return Z, X.astype(dtype), Y.astype(dtype)


def _adv(x, target, lsgan):
    fn = ops.squared_error_mean if lsgan else ops.bce_mean
    return T.scalar_loss(x, lambda v: fn(v, target))


def forward(state, Z, X, Y, dtype=np.float64, deterministic=False):
    """Shared forward of pix2pix.py:92-121.  -> dict of tape nodes."""
    cfg = state['cfg']
    g, d, u, p = cfg['gen_dcgan'], cfg['disc_dcgan'], cfg['gen_p2p'], cfg['disc_p2p']
    P = {k: [T.leaf(np.asarray(a, dtype)) for a in state['params'][k[0]][k[1]]] for k in NET_ORDER}
    z, x, y = T.leaf(np.asarray(Z, dtype)), T.leaf(np.asarray(X, dtype)), T.leaf(np.asarray(Y, dtype))
    fw = {'P': P}
    # dcgan (:92-95): G(z) is evaluated once and shared by D(G(z))
    fw['gz'], cur_g = nets.dcgan_gen_fwd(P[('dcgan', 'gen')], z, g['nch'], g['h'], g['initial_size'], g['div'],
                                         g['bilinear_upsample'], deterministic)
    dk = dict(in_shp=cfg['in_shp'], h=d['h'], div=d['div'], bn=d['bn'], nonlinearity=d['nonlinearity'],
              pool_mode=d['pool_mode'])
    fw['d_real'], _ = nets.dcgan_disc_fwd(P[('dcgan', 'disc')], x, **dk)
    fw['d_fake'], cur_d = nets.dcgan_disc_fwd(P[('dcgan', 'disc')], fw['gz'], **dk)
    # p2p (:98-101): U(X) shared by Dp(X, U(X)) and the reconstruction loss
    pk = dict(act=p['act'], mul_factor=p['mul_factor'], bn=p['bn'])
    fw['p_real'], _ = nets.patchgan_fwd(P[('p2p', 'disc')], x, y, **pk)
    fw['ux'], cur_u = nets.unet_fwd(P[('p2p', 'gen')], x, cfg['in_shp'], u['act'], u['bilinear_upsample'],
                                    deterministic)
    fw['p_fake'], cur_p = nets.patchgan_fwd(P[('p2p', 'disc')], x, fw['ux'], **pk)
    ls = cfg['lsgan']
    fw['gen_loss_dcgan'] = _adv(fw['d_fake'], 1.0, ls)                                      # :107
    fw['disc_loss_dcgan'] = T.add(_adv(fw['d_real'], 1.0, ls), _adv(fw['d_fake'], 0.0, ls))  # :108
    fw['gen_loss_p2p'] = _adv(fw['p_fake'], 1.0, ls)                                        # :110
    rec = ops.l2_mean if cfg['reconstruction'] == 'l2' else ops.l1_mean                     # :112-115
    fw['recon_loss'] = T.scalar_loss(fw['ux'], lambda v: rec(v, y.v))
    fw['gen_total_p2p'] = T.add(fw['gen_loss_p2p'], T.scale(fw['recon_loss'], cfg['alpha']))  # :117
    fw['disc_loss_p2p'] = T.add(_adv(fw['p_real'], 1.0, ls), _adv(fw['p_fake'], 0.0, ls))   # :121
    # discriminators with BatchNorm are evaluated twice (real, fake): each call attaches its own default_update to the
    # same running-statistics storage, both computed from the old value, so one survives -- unspecified which in
    # Theano; the later get_output call (the fake pass) is assumed, here and in the HIP path
    fw['bn_stats'] = {('dcgan', 'gen'): cur_g.bn_stats, ('p2p', 'gen'): cur_u.bn_stats,
                      ('dcgan', 'disc'): cur_d.bn_stats, ('p2p', 'disc'): cur_p.bn_stats}
    fw['inputs'] = (z, x, y)
    return fw


def losses_of(fw):
    return [float(fw[k].v) for k in ('gen_loss_dcgan', 'disc_loss_dcgan', 'gen_loss_p2p', 'recon_loss',
                                     'disc_loss_p2p')]                                     # order :142


def gradients(fw, state, nets_wanted=NET_ORDER):
    """Four independent T.grad roots (:132-135).  -> {net key: [grad per trainable param]}"""
    sp = specs(state['cfg'])
    roots = {('dcgan', 'gen'): ('gen_loss_dcgan', ()),
             ('dcgan', 'disc'): ('disc_loss_dcgan', ('gz',)),     # D-loss does not reach G's params
             ('p2p', 'gen'): ('gen_total_p2p', ()),
             ('p2p', 'disc'): ('disc_loss_p2p', ('ux',))}
    out = {}
    for key in nets_wanted:
        root, stops = roots[key]
        T.backward(fw[root], stop_at=[fw[s] for s in stops])
        gs = []
        for node, tr in zip(fw['P'][key], sp[key].trainable):
            if tr:
                gs.append(np.zeros_like(node.v) if node.g is None else node.g.copy())
        out[key] = gs
    return out


def _apply_opt(state, key, grads, cfg, dtype):
    sp = specs(cfg)[key]
    plist = state['params'][key[0]][key[1]]
    tr_idx = [i for i, t in enumerate(sp.trainable) if t]
    st = state['opt'].setdefault(key, None)
    if cfg['opt'] == 'rmsprop':
        if st is None:
            st = {'acc': [np.zeros(plist[i].shape, dtype) for i in tr_idx]}
        for j, i in enumerate(tr_idx):
            p, a = ops.rmsprop_step(np.asarray(plist[i], dtype), grads[j], st['acc'][j], cfg['lr'])
            plist[i], st['acc'][j] = p.astype(plist[i].dtype), a
    elif cfg['opt'] == 'adam':
        if st is None:
            st = {'m': [np.zeros(plist[i].shape, dtype) for i in tr_idx],
                  'v': [np.zeros(plist[i].shape, dtype) for i in tr_idx], 't': 0}
        t_new = st['t']
        for j, i in enumerate(tr_idx):
            p, m, v, t_new = ops.adam_step(np.asarray(plist[i], dtype), grads[j], st['m'][j], st['v'][j],
                                           st['t'], cfg['lr'])
            plist[i], st['m'][j], st['v'][j] = p.astype(plist[i].dtype), m, v
        st['t'] = t_new
    else:
        raise ValueError(cfg['opt'])
    state['opt'][key] = st


def _apply_bn_running(state, fw):
    for key, stats in fw['bn_stats'].items():
        plist = state['params'][key[0]][key[1]]
        for mean_idx, (mu, inv) in stats.items():
            m, s = ops.bn_running_update(np.asarray(plist[mean_idx], mu.dtype),
                                         np.asarray(plist[mean_idx + 1], mu.dtype), mu, inv)
            plist[mean_idx] = m.astype(plist[mean_idx].dtype)
            plist[mean_idx + 1] = s.astype(plist[mean_idx + 1].dtype)


def train_step(state, Z, X, Y, dtype=np.float64, update=True, want=()):
    """train_fn (update=True, pix2pix.py:142) or loss_fn (update=False, :143).
    Mutates ``state`` in place.  -> dict(losses=[5], grads={...} if computed, plus ``want``)."""
    cfg = state['cfg']
    fw = forward(state, Z, X, Y, dtype)
    res = {'losses': losses_of(fw)}
    for k in want:
        res[k] = fw[k].v.copy()
    if update:
        mode = cfg['train_mode']
        keys = [k for k in NET_ORDER if mode == 'both' or k[0] == mode]      # :131-141
        grads = gradients(fw, state, keys)
        res['grads'] = grads
        for key in keys:
            _apply_opt(state, key, grads[key], cfg, dtype)
    _apply_bn_running(state, fw)
    return res


def clone_state(state):
    return copy.deepcopy(state)

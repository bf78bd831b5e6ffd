"""CPU oracle for the gan-heightmaps hot path (TEST INFRASTRUCTURE ONLY).

This package is a numpy restatement of the arithmetic that the reference's compiled
Theano ``train_fn`` (/root/reference/pix2pix.py:142) executes for one (Z, X, Y) minibatch.
It exists so that the HIP path can be checked against something; it is never the thing
that is shipped or measured (only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it).

PARITY STATUS: **numerics unpinned**.  The reference is Python-2 source on top of
Theano + Lasagne 0.2.dev1, neither of which is vendored under /root/reference nor
installable here (no network), and the reference has no tests or golden vectors for this
path.  The op semantics below therefore follow the published
behaviour of those libraries, anchored on the reference's own call
sites.  What IS pinned by reference-held known answers (tests/test_oracle_known_answers.py):
parameter counts 22,882,243 (g_unet.ipynb:481) and 391,009 (g_unet.ipynb:558), the
per-layer output shapes (g_unet.ipynb:416-480, 547-557) and the [W, b] / [beta, gamma,
mean, inv_std] parameter order.  The vjp of every op is cross-checked against
finite differences and against torch-CPU autograd (tests/test_oracle_vs_torch.py).
"""

"""CPU oracle for the gan-heightmaps hot path (TEST INFRASTRUCTURE ONLY).

This package is a numpy restatement of the arithmetic that the reference's compiled
Theano ``train_fn`` (/root/reference/pix2pix.py:142) executes for one (Z, X, Y) minibatch.
It exists so that the HIP path can be checked against something; it is never the thing
that is shipped or measured (only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it).

PARITY STATUS: **op numerics unpinned; structure, wiring and host logic pinned on the reference's own code**.
The reference is Python-2 source on top of Theano + Lasagne 0.2.dev1, neither of which is vendored under
/root/reference nor installable here (no network), and the reference has no tests or golden vectors for this
path.  The arithmetic of each op (convolution, BatchNorm, bilinear up-sampling, pooling
gradients, RMSprop/Adam ...) therefore follows the published behaviour of those libraries
and is cross-checked against finite differences and torch-CPU autograd (tests/test_oracle_vs_torch.py) -- that
part stays unpinned.  Everything that IS reference code was executed in the build container and its outputs are
committed as fixtures under tests/golden/ together with the scripts that made them:
  * experiments.py + architectures/{dcgan,p2p,layers}.py  -> the layer graphs        (reference_graph.json)
  * pix2pix.py __init__ (loss / gradient / update wiring)  -> train_fn & co results    (reference_step*.npz)
  * pix2pix.py train / generate_* / save_model / load_model -> events, files, formats  (reference_trainloop.json,
                                                                                        reference_checkpoint.model)
  * util.py iterate_hdf5 / convert_to_rgb / compose_imgs    -> batches, seeds, images   (reference_iterator.npz)
and oracle/step.py, oracle/nets.py, oracle/keras_aug.py reproduce them (tests/test_reference_*.py).  Reference-held
known answers (tests/test_known_answers.py): parameter counts 22,882,243 (g_unet.ipynb:481) and 391,009
(g_unet.ipynb:558), per-layer output shapes (g_unet.ipynb:416-480, 547-557), parameter order.
"""

"""CPU oracle for the U-Net train-step hot path of igmsalinas/unet-rir.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``unet-rir_amd/``)
imports this directory; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

PARITY UNPINNED.  The reference is TensorFlow/Keras (dl_models/u_net.py:20-31)
and neither TensorFlow nor Keras is installed here (ordinary
ModuleNotFoundError, not a permission denial), so the reference cannot be run
to produce golden vectors, and it ships no tests, fixtures or known-answer
vectors of its own.  The arithmetic lives in
an unpinned third-party dependency (TensorFlow 2.x / Keras, est. 2.6-2.11 from
API usage).  What pins results instead:

  1. ``np_ops``    - direct-loop NumPy fp64 definitions of every op, written
                     from the published TF/Keras semantics (SAME padding,
                     Conv2DTranspose as the adjoint of the SAME strided conv,
                     BatchNormalization eps=1e-3/momentum=0.99, Keras Adam).
  2. ``torch_ref`` - an independent ``torch.nn.functional`` restatement of the
                     whole graph + loss + Adam step (fp32 or fp64), checked
                     op-by-op against (1) in tests/test_oracle.py.
  3. ``tests/golden/*.npz`` - vectors generated from (2) by
                     tests/golden/make_golden.py with platform-independent
                     inputs (``detrand``).

Every function cites the reference file:line it restates.
"""

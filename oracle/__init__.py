"""CPU oracle for the BLSTM / CTC / attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the thing timed as the CPU
baseline -- never as a fallback for the HIP path.

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.decoders`` (greedy + prefix beam search) is PINNED: it is checked
  against golden vectors produced by the reference's own numpy decoders
  (``/root/reference/models/ctc/decoders/*.py``) with the committed generator
  ``tests/golden/make_golden.py``.
* ``oracle.splice`` (frame stacking / splicing) is PINNED the same way against
  ``/root/reference/utils/io/inputs/{splicing,frame_stacking}.py``.
* Everything whose arithmetic lives in TensorFlow 1.x (LSTMBlockCell,
  dynamic_rnn masking, tf.nn.ctc_loss, conv2d SAME, optimizers ...) is
  "PARITY UNPINNED": TensorFlow 1.2/1.3 (requirements.txt:11) is not
  installable here and the reference's tests hold no golden numbers
  (models/test/test_ctc.py:225-233 only loops until LER < 0.1).  Those parts
  restate the published op semantics and are
  cross-checked against an independent second implementation
  (torch.nn.LSTM, torch.nn.functional.ctc_loss, finite differences).
"""

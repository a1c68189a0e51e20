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
* ``oracle.ctc`` (tf.nn.ctc_loss: loss and gradient) and the peephole-free cell of ``oracle.lstm`` are PINNED
  to TensorFlow's own known-answer tests: the constants of ``ctc_loss_op_test.py::testBasic`` (two losses, 60
  gradient entries) and of ``lstm_ops_test.py::testLSTMBlockCell`` (``tests/golden/tf_known_answers.py``) are
  reproduced to their printed precision; the greedy decoder also reproduces
  ``ctc_decoder_ops_test.py::testCTCGreedyDecoder``; the convolution layout / SAME-pool padding conventions of
  ``oracle.vgg``, Adagrad and clip_by_norm of ``oracle.optim`` reproduce the constants of TensorFlow's
  conv_ops / pooling_ops / adagrad / clip_ops tests.
* The label maps, the dataset iterators, the LR controller and the sparse-label helpers of the host side are
  PINNED to outputs recorded from the reference's own classes (``tests/golden/*.json``, ``datasets_v1.npz``).
* The rest of what lives in TensorFlow 1.x (peepholes and cell clip of LSTMBlockCell,
  dynamic_rnn masking, conv2d SAME, the attention decoder, optimizers ...) is
  "PARITY UNPINNED": TensorFlow 1.2/1.3 (requirements.txt:11) is not
  installable here and the reference's tests hold no golden numbers
  (models/test/test_ctc.py:225-233 only loops until LER < 0.1).  Those parts
  restate the published op semantics and are
  cross-checked against an independent second implementation
  (torch.nn.LSTM, torch.nn.functional.ctc_loss, finite differences).
"""

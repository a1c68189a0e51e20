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
* Round 5: ``oracle.attention`` (attention layer, decoder step, dynamic_decode, bridge, sequence loss, joint loss),
  the peephole / clip / projection cell and the encoders of ``oracle.lstm``, ``oracle.model`` (CTC model over blstm /
  lstm / bottleneck / vgg_blstm / LSTMCell + num_proj / gru / bgru), ``oracle.vgg`` and the per-variable clip are
  PINNED to the REFERENCE'S OWN CODE AS EXECUTED: ``tests/golden/make_golden_tfshim.py`` imports the unchanged files
  (models/attention/decoders/attention_layer.py, attention_decoder.py, dynamic_decoder.py, models/attention/bridge.py,
  attention_seq2seq.py, joint_ctc_attention.py, models/ctc/ctc.py, models/encoders/core/*.py,
  models/recurrent/layers/lstm.py, models/model_base.py) over an eager float64 stand-in for the TensorFlow ops they call
  (``tests/golden/tf_shim``; autograd gives the gradients of the reference's own forward graph), and
  ``tests/test_oracle_tfshim.py`` (65 cases) holds the oracle to the recorded numbers at 1e-11 (values) / 1e-9
  (gradients).  Adadelta is pinned to the recurrence of TensorFlow's adadelta_test.py, merge_repeated on a path with
  repeats to the worked example of TensorFlow's documentation of the op.
* What remains a restatement of TensorFlow-internal semantics INSIDE that stand-in (TF 1.2/1.3 is
  not installable here): dynamic_rnn's zero-output / state-copy rule and reverse_sequence, SAME padding,
  TrainingHelper / GreedyEmbeddingHelper, sequence_loss, LSTMBlockCell's straight-through clip in the gradient.  Each is
  a few lines, checked where TensorFlow publishes constants (lstm_ops_test, core_rnn_cell_test, conv / pooling tests,
  ctc tests) and against the reference's Python LSTMCell; tf.nn.ctc_loss inside the stand-in is
  torch.nn.functional.ctc_loss -- a third implementation beside oracle.ctc and the HIP kernel.
"""

"""Known-answer vectors of TensorFlow's OWN unit tests for the two ops the hot path takes from TensorFlow
(tensorflow==1.2/1.3, the reference's pinned dependency), so that the oracle -- and through it the HIP
kernels -- is pinned to TensorFlow's results, not only to an independent restatement:

  * tf.nn.ctc_loss: tensorflow/python/kernel_tests/ctc_loss_op_test.py, CTCLossTest.testBasic -- two utterances,
    depth 6 (blank = 5), 5 frames each; expected loss and the full gradient w.r.t. the (log-probability) inputs.
  * tf.contrib.rnn.LSTMBlockCell (no peephole, forget_bias 1): tensorflow/contrib/rnn/python/kernel_tests/
    lstm_ops_test.py, LSTMBlockCellTest.testLSTMBlockCell (the same numbers as rnn_cell_test.py testBasicLSTMCell):
    two stacked cells of 2 units, every weight 0.5, zero bias, x = [1, 1], every state entry 0.1.

TensorFlow is not installable here; the numbers are the constants printed in those test files.  They validate
themselves: the oracle, written from the op semantics, reproduces both losses to 3e-6 (the printed precision) and
all 60 gradient entries to 5e-7 (tests/test_oracle.py) -- which no independent implementation does by accident."""
import numpy as np

CTC_DEPTH = 6           # classes incl. blank (index 5 = depth - 1, TensorFlow's convention)

CTC_TARGETS_0 = [0, 1, 2, 1, 0]
CTC_LOSS_0 = 3.34211
CTC_PROBS_0 = np.asarray(
    [[0.633766, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553],
     [0.111121, 0.588392, 0.278779, 0.0055756, 0.00569609, 0.010436],
     [0.0357786, 0.633813, 0.321418, 0.00249248, 0.00272882, 0.0037688],
     [0.0663296, 0.643849, 0.280111, 0.00283995, 0.0035545, 0.00331533],
     [0.458235, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]], dtype=np.float64)
CTC_GRAD_0 = np.asarray(
    [[-0.366234, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553],
     [0.111121, -0.411608, 0.278779, 0.0055756, 0.00569609, 0.010436],
     [0.0357786, 0.633813, -0.678582, 0.00249248, 0.00272882, 0.0037688],
     [0.0663296, -0.356151, 0.280111, 0.00283995, 0.0035545, 0.00331533],
     [-0.541765, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]], dtype=np.float64)

CTC_TARGETS_1 = [0, 1, 1, 0]
CTC_LOSS_1 = 5.42262
CTC_PROBS_1 = np.asarray(
    [[0.30176, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508],
     [0.24082, 0.397533, 0.0557226, 0.0546814, 0.0557528, 0.19549],
     [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, 0.202456],
     [0.280884, 0.429522, 0.0326593, 0.0339046, 0.0326856, 0.190345],
     [0.423286, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]], dtype=np.float64)
CTC_GRAD_1 = np.asarray(
    [[-0.69824, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508],
     [0.24082, -0.602467, 0.0557226, 0.0546814, 0.0557528, 0.19549],
     [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, -0.797544],
     [0.280884, -0.570478, 0.0326593, 0.0339046, 0.0326856, 0.190345],
     [-0.576714, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]], dtype=np.float64)

CTC_CASES = [(CTC_PROBS_0, CTC_TARGETS_0, CTC_LOSS_0, CTC_GRAD_0), (CTC_PROBS_1, CTC_TARGETS_1, CTC_LOSS_1, CTC_GRAD_1)]

# LSTMBlockCell x 2, num_units 2, all weights 0.5, x = [[1, 1]], c = h = 0.1 everywhere
LSTM_WEIGHT, LSTM_X, LSTM_STATE = 0.5, [1.0, 1.0], 0.1
LSTM_C0, LSTM_H0 = [0.68967271, 0.68967271], [0.44848421, 0.44848421]
LSTM_C1, LSTM_H1 = [0.39897051, 0.39897051], [0.24024698, 0.24024698]

# tf.nn.ctc_greedy_decoder: tensorflow/python/kernel_tests/ctc_decoder_ops_test.py, testCTCGreedyDecoder -- depth 4
# (blank = 3), 6 padded frames, two utterances of 4 and 5 frames; decoded labels and the negative log probability of
# the best path (sum of -log of the per-frame maxima).
GREEDY_SEQ_LEN = [4, 5]
GREEDY_PROBS = np.asarray(
    [[[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.4, 0.6], [0.0, 0.0, 0.4, 0.6], [0.0, 0.9, 0.1, 0.0], [0.0, 0.0, 0.0, 0.0],
      [0.0, 0.0, 0.0, 0.0]],
     [[0.1, 0.9, 0.0, 0.0], [0.0, 0.9, 0.1, 0.0], [0.0, 0.0, 0.1, 0.9], [0.0, 0.9, 0.1, 0.1], [0.9, 0.1, 0.0, 0.0],
      [0.0, 0.0, 0.0, 0.0]]], dtype=np.float64)                                   # [B, T, depth]
GREEDY_DECODED = [[0, 1], [1, 1, 0]]
GREEDY_NEG_LOG_PROB = [float(np.sum(-np.log([1.0, 0.6, 0.6, 0.9]))), float(np.sum(-np.log([0.9] * 5)))]

# tf.nn.ctc_beam_search_decoder: tensorflow/python/kernel_tests/ctc_decoder_ops_test.py, testCTCDecoderBeamSearch ("one
# batch, two beams -- hibernating beam search") -- depth 6 (blank = 5), one utterance of 5 frames (a sixth, random frame
# and two zero frames lie beyond seq_len), beam_width = 2, top_paths = 2, merge_repeated = False.  The test feeds
# log(probabilities) + 2.0 ("arbitrary offset -- this is fine": the decoder works on frame-normalised scores) and expects
# the two beams [1, 0] and [0, 1, 0] with log_probability 0.584855 / 0.389139.  Those numbers are POSITIVE because the
# TF1 kernel normalises every frame by its MAXIMUM, not by its log-sum-exp: log_probability = log p(path | x) +
# sum_t -log max_c softmax(x_t)_c  (here -3.582118 + 4.166973 and -3.777833 + 4.166973).  Restated, not a copied file.
BEAM_PROBS = np.asarray(
    [[0.30999, 0.309938, 0.0679938, 0.0673362, 0.0708352, 0.173908],
     [0.215136, 0.439699, 0.0370931, 0.0393967, 0.0381581, 0.230517],
     [0.199959, 0.489485, 0.0233221, 0.0251417, 0.0233289, 0.238763],
     [0.279611, 0.452966, 0.0204795, 0.0209126, 0.0194803, 0.20655],
     [0.51286, 0.288951, 0.0243026, 0.0220788, 0.0219297, 0.129878],
     [0.155251, 0.164444, 0.173517, 0.176138, 0.169979, 0.160671]], dtype=np.float64)   # [T = 6, depth]; seq_len = 5
BEAM_SEQ_LEN, BEAM_WIDTH, BEAM_BLANK, BEAM_LOGIT_OFFSET, BEAM_PADDED_FRAMES = 5, 2, 5, 2.0, 8
BEAM_DECODED = [[1, 0], [0, 1, 0]]                    # beam 0, beam 1
BEAM_LOG_PROB = [0.584855, 0.389139]                  # TF1's max-normalised log-probabilities of the two beams


# merge_repeated on a path WITH repeats: the worked example in the documentation of tf.nn.ctc_beam_search_decoder /
# tf.nn.ctc_greedy_decoder (tensorflow/python/ops/ctc_ops.py): "if consecutive entries in a beam are the same, only the
# first of these is emitted.  That is, when the sequence is `A B B * B * B` (where '*' is the blank label), the return
# value is: `A B` if merge_repeated = True; `A B B B` if merge_repeated = False."  A = 0, B = 1, blank = 2; the frame
# posteriors put 0.9 on the documented sequence, so it is the best alignment and A B B B the best labelling.
MERGE_DOC_FRAMES = [0, 1, 1, 2, 1, 2, 1]              # A B B * B * B
MERGE_DOC_DEPTH, MERGE_DOC_BLANK, MERGE_DOC_PEAK = 3, 2, 0.9
MERGE_DOC_MERGED, MERGE_DOC_UNMERGED = [0, 1], [0, 1, 1, 1]


def merge_doc_probs():
    p = np.full((len(MERGE_DOC_FRAMES), MERGE_DOC_DEPTH), (1.0 - MERGE_DOC_PEAK) / (MERGE_DOC_DEPTH - 1))
    p[np.arange(len(MERGE_DOC_FRAMES)), MERGE_DOC_FRAMES] = MERGE_DOC_PEAK
    return p


def beam_max_normaliser(probs, seq_len):
    """sum_t -log max_c p[t, c]: what TF1's frame-max normalisation adds to log p(path | x)."""
    return float(np.sum(-np.log(np.asarray(probs)[:seq_len].max(1))))


# tf.train.AdagradOptimizer: tensorflow/python/training/adagrad_test.py, doTestBasic -- learning rate 3.0,
# initial_accumulator_value 0.1, constant gradients, 3 steps.
ADAGRAD_LR, ADAGRAD_STEPS = 3.0, 3
ADAGRAD_VAR0, ADAGRAD_GRAD0, ADAGRAD_OUT0 = [1.0, 2.0], [0.1, 0.1], [-1.6026098728179932, -0.6026098728179932]
ADAGRAD_VAR1, ADAGRAD_GRAD1, ADAGRAD_OUT1 = [3.0, 4.0], [0.01, 0.01], [2.715679168701172, 3.715679168701172]

# The other tf.train optimizers, from the expectations TensorFlow's own tests of them spell out (TF r1.2 / r1.3,
# tensorflow/python/training/*_test.py; two variables [1, 2] and [3, 4], constant gradients).  Restated from those
# tests' arithmetic -- the expressions below are theirs, evaluated here -- not copied files.
#  * gradient_descent_test.py testBasic: learning rate 3.0, gradients 0.1 / 0.01, one step.
SGD_LR, SGD_VAR, SGD_GRAD = 3.0, [[1.0, 2.0], [3.0, 4.0]], [[0.1, 0.1], [0.01, 0.01]]
SGD_OUT = [[1.0 - 3.0 * 0.1, 2.0 - 3.0 * 0.1], [3.0 - 3.0 * 0.01, 4.0 - 3.0 * 0.01]]
#  * momentum_test.py testBasic: learning rate 2.0, momentum 0.9, gradients 0.1 / 0.01, two steps; the accumulator
#    holds 0.1 then 0.9 * 0.1 + 0.1 (accum = momentum * accum + grad; var -= lr * accum).
MOM_LR, MOM_MOMENTUM, MOM_VAR, MOM_GRAD = 2.0, 0.9, [[1.0, 2.0], [3.0, 4.0]], [[0.1, 0.1], [0.01, 0.01]]
MOM_OUT_STEP1 = [[1.0 - (0.1 * 2.0), 2.0 - (0.1 * 2.0)], [3.0 - (0.01 * 2.0), 4.0 - (0.01 * 2.0)]]
MOM_OUT_STEP2 = [[1.0 - (0.1 * 2.0) - ((0.9 * 0.1 + 0.1) * 2.0), 2.0 - (0.1 * 2.0) - ((0.9 * 0.1 + 0.1) * 2.0)],
                 [2.98 - ((0.9 * 0.01 + 0.01) * 2.0), 3.98 - ((0.9 * 0.01 + 0.01) * 2.0)]]
MOM_ACCUM_STEP2 = [[0.9 * 0.1 + 0.1] * 2, [0.9 * 0.01 + 0.01] * 2]
#  * momentum_test.py testNesterovMomentum's numpy reference (_update_nesterov_momentum_numpy):
#    var += accum * lr * momentum; accum = accum * momentum + g; var -= lr * accum; var -= accum * lr * momentum.


def nesterov_reference(var, accum, g, lr, momentum):
    var = var + accum * lr * momentum
    accum = accum * momentum + g
    var = var - lr * accum
    var = var - accum * lr * momentum
    return var, accum


#  * rmsprop_test.py testWithoutMomentum: learning rate 2.0, decay 0.9, momentum 0, epsilon 1.0; gradients
#    [0.1, 0.2] / [0.01, 0.2]; the rms slot STARTS AT ONE (0.901 = 0.9 * 1 + 0.1 * 0.1^2) and epsilon sits INSIDE
#    the root (sqrt(0.901 + 1.0)); two steps.
RMS_LR, RMS_DECAY, RMS_EPS = 2.0, 0.9, 1.0
RMS_VAR, RMS_GRAD = [[1.0, 2.0], [3.0, 4.0]], [[0.1, 0.2], [0.01, 0.2]]
RMS_SLOT_STEP1 = [[0.901, 0.904], [0.90001, 0.904]]
RMS_OUT_STEP1 = [[1.0 - (0.1 * 2.0 / np.sqrt(0.901 + 1.0)), 2.0 - (0.2 * 2.0 / np.sqrt(0.904 + 1.0))],
                 [3.0 - (0.01 * 2.0 / np.sqrt(0.90001 + 1.0)), 4.0 - (0.2 * 2.0 / np.sqrt(0.904 + 1.0))]]
RMS_SLOT_STEP2 = [[0.901 * 0.9 + 0.001, 0.904 * 0.9 + 0.004], [0.90001 * 0.9 + 1e-5, 0.904 * 0.9 + 0.004]]
RMS_OUT_STEP2 = [[RMS_OUT_STEP1[0][0] - (0.1 * 2.0 / np.sqrt(0.901 * 0.9 + 0.001 + 1.0)),
                  RMS_OUT_STEP1[0][1] - (0.2 * 2.0 / np.sqrt(0.904 * 0.9 + 0.004 + 1.0))],
                 [RMS_OUT_STEP1[1][0] - (0.01 * 2.0 / np.sqrt(0.90001 * 0.9 + 1e-5 + 1.0)),
                  RMS_OUT_STEP1[1][1] - (0.2 * 2.0 / np.sqrt(0.904 * 0.9 + 0.004 + 1.0))]]
#  * adam_test.py testBasic's numpy reference (adam_update_numpy), default hyper-parameters, gradients 0.1 / 0.01,
#    three steps: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); m = beta1 m + (1 - beta1) g; v = beta2 v + (1 - beta2) g^2;
#    param -= lr_t * m / (sqrt(v) + epsilon).


def adam_reference(param, g, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m_t = beta1 * m + (1 - beta1) * g
    v_t = beta2 * v + (1 - beta2) * g * g
    return param - alpha_t * m_t / (np.sqrt(v_t) + epsilon), m_t, v_t


ADAM_VAR, ADAM_GRAD, ADAM_STEPS = [[1.0, 2.0], [3.0, 4.0]], [[0.1, 0.1], [0.01, 0.01]], 3

#  * adadelta_test.py doTestBasic: variables [1, 2] / [3, 4], the SAME constant gradient for both, for grad in
#    {0.2, 0.1, 0.01} x lr in {1.0, 0.5, 0.1}; rho 0.95, epsilon 1e-8, four updates.  The test carries its own scalar
#    recurrence and asserts after every step that both slots and both variables follow it:
#        accum        = accum * rho + grad^2 * (1 - rho)
#        update       = sqrt(accum_update + epsilon) * (1 / sqrt(accum + epsilon)) * grad
#        accum_update = accum_update * rho + update^2 * (1 - rho)
#        tot_update  += update * lr ;  var == var_init - tot_update
ADADELTA_GRADS, ADADELTA_LRS, ADADELTA_RHO, ADADELTA_EPS, ADADELTA_STEPS = [0.2, 0.1, 0.01], [1.0, 0.5, 0.1], 0.95, 1e-8, 4
ADADELTA_VAR = [[1.0, 2.0], [3.0, 4.0]]


def adadelta_reference(grad, lr, steps, rho=0.95, epsilon=1e-8):
    """-> per step (accum, accum_update, tot_update), the test's scalars."""
    accum = accum_update = tot_update = 0.0
    out = []
    for _ in range(steps):
        accum = accum * rho + (grad ** 2) * (1 - rho)
        update = ((accum_update + epsilon) ** 0.5 * (accum + epsilon) ** (-0.5) * grad)
        accum_update = accum_update * rho + (update ** 2) * (1.0 - rho)
        tot_update += update * lr
        out.append((accum, accum_update, tot_update))
    return out

# tf.clip_by_norm: tensorflow/python/kernel_tests/clip_ops_test.py, testClipByNormClipped / NotClipped
CLIP_X = [[-3.0, 0.0, 0.0], [4.0, 0.0, 0.0]]
CLIP_NORM_CLIPPED, CLIP_ANS_CLIPPED = 4.0, [[-2.4, 0.0, 0.0], [3.2, 0.0, 0.0]]
CLIP_NORM_NOT_CLIPPED = 6.0

# tf.nn.conv2d (NHWC input, HWIO filter): tensorflow/python/kernel_tests/conv_ops_test.py, testConv2D2x2Filter and
# testConv2D1x2Filter -- input [1,2,3,3] and filter filled with 1, 2, 3, ... in row-major order, stride 1, VALID.
CONV_IN_SHAPE = (1, 2, 3, 3)
CONV_2X2_FILTER_SHAPE, CONV_2X2_OUT = (2, 2, 3, 3), [2271.0, 2367.0, 2463.0, 2901.0, 3033.0, 3165.0]
CONV_1X2_FILTER_SHAPE, CONV_1X2_OUT = (1, 2, 3, 3), [231.0, 252.0, 273.0, 384.0, 423.0, 462.0, 690.0, 765.0, 840.0,
                                                     843.0, 936.0, 1029.0]
# tf.nn.max_pool 2x2 stride 2 SAME on an odd width: tensorflow/python/kernel_tests/pooling_ops_test.py,
# testMaxPoolSamePadding -- input [1,2,3,3] = 1..18: the extra column is padded AFTER.
POOL_SAME_OUT = [13.0, 14.0, 15.0, 16.0, 17.0, 18.0]

# tensorflow/contrib/rnn/python/kernel_tests/core_rnn_cell_test.py::testGRUCell (TF 1.x): every kernel entry 0.5
# (variable_scope initializer), gate bias 1 and candidate bias 0 (the cell's own initializers), h = [0.1, 0.1]:
#   x = [1, 1]    -> new h = [0.175991, 0.175991]
#   x = [1, 1, 1] -> new h = [0.156736, 0.156736]
GRU_CASES = [
    dict(x=[1.0, 1.0], h=[0.1, 0.1], kernel=0.5, out=[0.175991, 0.175991]),
    dict(x=[1.0, 1.0, 1.0], h=[0.1, 0.1], kernel=0.5, out=[0.156736, 0.156736]),
]


"""CTC model -- mirror of models/ctc/ctc.py:15-398 (class CTC) on the HIP path.

Same constructor arguments, attributes (name, num_classes = +1 blank, *_pl_list,
summaries_*), and methods: create_placeholders, compute_loss -> (total_loss,
logits[T,B,C]), train (ModelBase), decoder(logits, inputs_seq_len, beam_width),
posteriors(logits), compute_ler(decode_op, labels).

TF1 is deferred graph + session; this is eager: compute_loss RUNS encoder -> FC ->
CTC forward and keeps the activations; train(loss, optimizer, lr) runs backward,
per-variable clip, weight decay and the optimizer update for that forward.
"""
import numpy as np
import torch

from ... import ops
from ..._lib import ASR_BF16, ASR_F32
from ...utils.io.labels.sparsetensor import dense_to_flat, sparse_to_flat
from ...utils.evaluation.edit_distance import compute_ler as _ler
from ...utils.parameter import ParamStore
from ..encoders.load_encoder import load
from ..model_base import ModelBase


def truncated_normal(rng, stddev, shape):
    """tf.truncated_normal_initializer(stddev): N(0, s^2) resampled outside +-2s."""
    x = rng.normal(0.0, stddev, size=shape)
    bad = np.abs(x) > 2 * stddev
    while bad.any():
        x[bad] = rng.normal(0.0, stddev, size=int(bad.sum()))
        bad = np.abs(x) > 2 * stddev
    return x


class Placeholder(object):
    """Eager stand-in for tf.placeholder: a named slot the driver fills before a step."""

    def __init__(self, name, dtype=None, shape=None):
        self.name, self.dtype, self.shape = name, dtype, shape
        self.value = None

    def feed(self, value):
        self.value = value
        return self


def _not_enough_time(n):
    return ValueError('Not enough time for target transition sequence (%d utterance(s))' % n)


class CTC(ModelBase):
    """Connectionist Temporal Classification (CTC) network (models/ctc/ctc.py:15-57 docstring).

    Extra keyword arguments of the HIP build (not in the reference): `dtype` ('f32' exact
    fp32 MFMA path | 'bf16' operands with fp32 accumulate), `device`, `seed`.
    """
    head_scope = 'output'      # variable scope of the output FC (ctc.py:216-224)

    def __init__(self, encoder_type, input_size, num_units, num_layers, num_classes,
                 lstm_impl='LSTMBlockCell', use_peephole=True, splice=1, num_stack=1,
                 parameter_init=0.1, clip_grad_norm=None, clip_activation=None, num_proj=None,
                 weight_decay=0.0, bottleneck_dim=None, time_major=True,
                 dtype='f32', device='cuda:0', seed=0):
        super(CTC, self).__init__()
        assert input_size % 3 == 0, 'input_size must be divisible by 3 (+ delta, acceleration coefficients).'
        assert splice % 2 == 1, 'splice must be the odd number'
        if clip_grad_norm is not None:
            assert float(clip_grad_norm) > 0, 'clip_grad_norm must be larger than 0.'
        assert float(weight_decay) >= 0, 'weight_decay must not be a negative value.'

        self.encoder_type = encoder_type
        self.input_size = input_size
        self.splice = splice
        self.num_stack = num_stack
        self.num_units = num_units
        # the reference does int(num_proj) before the None check (ctc.py:93-98, TypeError on the
        # default); None / 0 both mean "no projection" here
        self.num_proj = int(num_proj) if num_proj not in (None, 0, '0') else None
        self.num_layers = num_layers
        self.bottleneck_dim = bottleneck_dim
        self.num_classes = num_classes + 1  # + blank
        self.lstm_impl = lstm_impl
        self.use_peephole = use_peephole
        self.parameter_init = parameter_init
        self.clip_grad_norm = clip_grad_norm
        self.clip_activation = clip_activation
        self.weight_decay = weight_decay
        self.summaries_train = []
        self.summaries_dev = []
        self.inputs_pl_list = []
        self.labels_pl_list = []
        self.inputs_seq_len_pl_list = []
        self.keep_prob_pl_list = []
        self.time_major = time_major
        self.name = encoder_type + '_ctc'
        self.dtype = ops.dtype_id(dtype)
        self._requested_dtype = self.dtype               # the projected layers' operand dtype (rnn_util.LSTMPLayer)
        if encoder_type in ('gru', 'bgru') or (lstm_impl == 'LSTMCell' and self.num_proj is not None):
            # the GRU recurrences are fp32 and the projected-LSTM layers take and return fp32: the heads follow
            self.dtype = ASR_F32
        self.device = torch.device(device)
        self._dropout_calls = 0
        self.seed = seed

        self.bottleneck_dim = int(bottleneck_dim) if bottleneck_dim not in (None, 0) else None

        self.encoder = self._create_encoder(encoder_type, input_size, splice, num_stack, num_units, num_layers,
                                            lstm_impl, use_peephole, parameter_init, clip_activation)

        # variables: encoder, then the output head(s) (ctc.py:216-224)
        rng = np.random.RandomState(seed)
        self.store = ParamStore(self.device)
        enc_dim = self.encoder.build(self.store, input_size * num_stack * splice, rng)
        self._declare_heads(rng, enc_dim, parameter_init)
        self.store.finalize()
        self._tape = None

    def _create_encoder(self, encoder_type, input_size, splice, num_stack, num_units, num_layers, lstm_impl,
                        use_peephole, parameter_init, clip_activation):
        """ctc.py:124-170: constructor keywords differ per encoder family."""
        if encoder_type in ['blstm', 'lstm']:
            return load(encoder_type)(
                num_units=num_units, num_proj=self.num_proj, num_layers=num_layers,
                lstm_impl=lstm_impl, use_peephole=use_peephole, parameter_init=parameter_init,
                clip_activation=clip_activation, time_major=True, dtype=self._requested_dtype)
        if encoder_type in ['vgg_blstm', 'vgg_lstm', 'cldnn_wang']:
            return load(encoder_type)(
                input_size=input_size, splice=splice, num_stack=num_stack, num_units=num_units,
                num_proj=self.num_proj, num_layers=num_layers, lstm_impl=lstm_impl,
                use_peephole=use_peephole, parameter_init=parameter_init,
                clip_activation=clip_activation, time_major=True, dtype=self._requested_dtype)
        if encoder_type in ['bgru', 'gru']:              # ctc.py:150-155
            return load(encoder_type)(num_units=num_units, num_layers=num_layers, parameter_init=parameter_init,
                                      time_major=True)
        load(encoder_type)  # ValueError for unknown keys, as load_encoder.py:53-56
        raise NotImplementedError

    def _declare_heads(self, rng, enc_dim, parameter_init):
        if self.bottleneck_dim:   # ctc.py:201-216: FC(relu) named by its variable scope, then dropout
            self.store.declare('bottleneck/weights', (enc_dim, self.bottleneck_dim),
                               truncated_normal(rng, parameter_init, (enc_dim, self.bottleneck_dim)))
            self.store.declare('bottleneck/biases', (self.bottleneck_dim,), np.zeros(self.bottleneck_dim))
            enc_dim = self.bottleneck_dim
        self.store.declare(self.head_scope + '/weights', (enc_dim, self.num_classes),
                           truncated_normal(rng, parameter_init, (enc_dim, self.num_classes)))
        self.store.declare(self.head_scope + '/biases', (self.num_classes,), np.zeros(self.num_classes))

    # ------------------------------------------------------------------ graph pieces
    def _build(self, inputs, inputs_seq_len, keep_prob, is_training):
        """ctc.py:175-238: encoder -> [T*B, 2H] -> output FC -> logits [T,B,C] (time-major)."""
        rng_state = None
        if is_training and keep_prob is not None and float(keep_prob) < 1.0:
            self._dropout_calls += 1
            rng_state = (self.seed, self._dropout_calls << 40)
        # operand-dtype shadow of the variables the heads multiply with: refreshed on the encoder's side stream right
        # after its weight images and before its dropout masks (the main stream waits for the mask events, so it is
        # complete long before the output FC)
        if hasattr(self.encoder, 'side_extra'):
            self.encoder.side_extra = self._refresh_head_shadow
        self.encoder.want_f32_outputs = False            # the heads consume the operand copy
        self.encoder(inputs, inputs_seq_len, float(keep_prob) if keep_prob is not None else 1.0,
                     is_training, rng_state=rng_state)
        ops.wait_event(getattr(self.encoder, 'side_extra_event', None))
        sh = self.store.shadow(self.dtype)               # refreshed by side_extra where the encoder offers it
        x_op = self._enc_operand()                       # [T,Bp,E]
        T, Bp, E = x_op.shape
        enc = x_op
        self._bn = None
        if self.bottleneck_dim:
            # bottleneck FC + ReLU (+ dropout on the hidden-output connection), ctc.py:201-216
            h = ops.gemm(x_op.view(T * Bp, E), sh['bottleneck/weights'], bias=self.store['bottleneck/biases'],
                         relu=True)
            mask = None
            hd = h
            if rng_state is not None:
                mask = ops.dropout_mask(h.shape, float(keep_prob), rng_state[0] + 11, rng_state[1], enc.device)
                hd = ops.apply_mask(h, mask)
            self._bn = dict(x=x_op, h=h, hd=hd, mask=mask)
            x_op, E = hd.view(T, Bp, self.bottleneck_dim), self.bottleneck_dim
        self._head_in = x_op
        logits = torch.empty((T, Bp, self.num_classes), dtype=torch.float32, device=enc.device)
        ops.gemm(x_op.view(T * Bp, E), sh[self.head_scope + '/weights'], bias=self.store[self.head_scope + '/biases'],
                 out=logits.view(T * Bp, self.num_classes))
        return logits

    def _refresh_head_shadow(self):
        """Operand-dtype copies of the head weights (cast per variable on first use after an update)."""
        sh = self.store.shadow(self.dtype)
        if sh is not self.store:
            for n in self.store.names:
                if n.endswith('/weights') and not n.startswith(('VGG', 'CNN', 'bridge', 'fc')):
                    sh[n]

    def _enc_operand(self):
        """Encoder output in the MFMA operand dtype (what the output FC consumes)."""
        return self.encoder._out_op

    def create_placeholders(self):
        """ctc.py:240-254."""
        self.inputs_pl_list.append(Placeholder('input', np.float32,
                                               [None, None, self.input_size * self.num_stack * self.splice]))
        self.labels_pl_list.append(Placeholder('labels'))
        self.inputs_seq_len_pl_list.append(Placeholder('inputs_seq_len', np.int32, [None]))
        self.keep_prob_pl_list.append(Placeholder('keep_prob', np.float32))

    @staticmethod
    def _labels_to_flat(labels, batch_size):
        if isinstance(labels, (list, tuple)) and len(labels) == 3 and np.asarray(labels[0]).ndim == 2 \
                and np.asarray(labels[0]).shape[1] == 2 and np.asarray(labels[2]).shape == (2,):
            return sparse_to_flat(labels, batch_size)
        if torch.is_tensor(labels):
            labels = labels.cpu().numpy()
        return dense_to_flat(np.asarray(labels), -1)

    def compute_loss(self, inputs, labels, inputs_seq_len, keep_prob, scope=None,
                     softmax_temperature=1, is_training=True):
        """ctc.py:256-323.  inputs [B,T,input_size] fp32; labels: the SparseTensor triple of
        list2sparsetensor or a dense [B,Lmax] array padded -1; inputs_seq_len [B].
        Returns (total_loss 0-dim cuda tensor, logits [T,B,num_classes])."""
        dev = self.device
        # host-side view of the lengths for encoders that plan on the host (VGG valid-frame packing, GRU tmax): taken
        # from what the caller handed over, never by reading a device copy back (that would drain the stream each step)
        self.encoder._lens_host = ops.host_ints(inputs_seq_len)
        inputs = ops.to_device(inputs, torch.float32, dev)
        B = inputs.shape[0]
        flat, offsets, max_len = self._labels_to_flat(labels, B)
        Bp = B + (-B) % 16                               # the encoder pads the batch to whole 16-utterance tiles
        if Bp > B:
            offsets = np.concatenate([offsets, np.full(Bp - B, offsets[-1], dtype=np.int32)])
        # frame counts and labels go up FIRST and together (one pinned staging buffer, one async copy on the upload
        # stream), i.e. before the forward is enqueued: behind it the labels would sit on the critical path between the
        # output FC and the CTC kernels
        inputs_seq_len, off_d, flat_d = ops.upload_ints(
            dev, [inputs_seq_len, offsets, flat if len(flat) else np.zeros(1, np.int32)])
        logits = self._build(inputs, inputs_seq_len, keep_prob, is_training)
        T, Bp, C = logits.shape
        ctc_in = logits
        inv_temp = 1.0 / float(softmax_temperature)
        if softmax_temperature != 1:
            ctc_in = ops.scale_(logits.clone(), inv_temp)
        ctc_losses, grad, ninf = ops.ctc_loss(ctc_in, flat_d, off_d, self.encoder.seq_len_padded,
                                              max_len, grad_scale=inv_temp / B, want_grad=is_training)
        ctc_loss = ctc_losses[:B].mean()                                   # ctc.py:298
        total_loss = ctc_loss
        if self.weight_decay > 0:
            l2 = torch.zeros((), dtype=torch.float32, device=dev)
            ops.weight_decay(None, self.store.flat, self.store.plan, self.store.decay_mask,
                             self.weight_decay, l2_out=l2)
            total_loss = ctc_loss + l2                                      # ctc.py:280-302
        self.ctc_losses = ctc_losses[:B]
        self.num_infeasible = ninf
        # tf.nn.ctc_loss(ignore_longer_outputs_than_inputs=False) fails the step (ctc.py:289); here the counter is watched
        # asynchronously while training (raises at most ops.DeferredCheck.DEPTH = 4 steps late, the rows themselves contribute 0 loss / 0 gradient) and
        # checked at once in evaluation
        ops.defer_zero_check(ninf, _not_enough_time, blocking=not is_training)
        self._tape = dict(dlogits=grad, B=B) if is_training else None
        total_loss._asr_model = self
        return total_loss, logits[:, :B]

    # ------------------------------------------------------------------ backward
    def _backward(self):
        if self._tape is None:
            raise RuntimeError('train()/compute_gradients() needs a preceding compute_loss(is_training=True)')
        tape, st = self._tape, self.store
        dlogits = tape['dlogits']
        T, Bp, C = dlogits.shape
        x_op = self._head_in                             # what the output FC consumed
        E = x_op.shape[2]
        sh = st.shadow(self.dtype)
        dl2d = dlogits.view(T * Bp, C)
        dl_op = ops.cast_from_f32(dl2d, ASR_BF16) if self.dtype == ASR_BF16 else dl2d
        denc = ops.gemm(dl_op, sh[self.head_scope + '/weights'], transB=True, out_dtype=ASR_F32)
        with ops.side_lane(dlogits.device, keep=(x_op, dl_op, dl2d)):   # joined by encoder.backward
            ops.gemm(x_op.view(T * Bp, E), dl_op, transA=True, out=st.g(self.head_scope + '/weights'))
            ops.colsum(dl2d, out=st.g(self.head_scope + '/biases'))
        if self._bn is not None:
            bn = self._bn
            dpre = ops.relu_bwd(denc, bn['h'], bn['mask'])                  # operand dtype
            xe = bn['x']
            Ee = xe.shape[2]
            ops.gemm(xe.view(T * Bp, Ee), dpre, transA=True, out=st.g('bottleneck/weights'))
            ops.colsum(dpre, out=st.g('bottleneck/biases'))
            denc = ops.gemm(dpre, sh['bottleneck/weights'], transB=True, out_dtype=ASR_F32)
            E = Ee
            self._bn = None
        self.encoder.backward(denc.view(T, Bp, E), **self._encoder_backward_extra())
        if self.weight_decay > 0:
            ops.weight_decay(st.grad, st.flat, st.plan, st.decay_mask, self.weight_decay)
        self._tape = None

    def _encoder_backward_extra(self):
        """Further gradients entering the encoder (the sub-task head of MultitaskCTC)."""
        return {}

    # ------------------------------------------------------------------ decode / eval
    def decoder(self, logits, inputs_seq_len, beam_width=1, merge_repeated=True):
        """ctc.py:325-352.  Returns the decoded labels as the SparseTensor triple
        [indices int64 [n,2], values int32 [n], dense_shape int64 [2]] (host numpy), i.e. what
        sess.run(decode_op) hands to sparsetensor2list in the reference.
        merge_repeated (beam_width > 1 only; extension keyword): the reference calls
        tf.nn.ctc_beam_search_decoder(logits, seq_len, beam_width=...) with TensorFlow's default
        merge_repeated=True (ctc.py:344-346), under which consecutive equal labels of the OUTPUT beam are collapsed to
        their first occurrence -- 'a a' can never be emitted.  True (default) reproduces that
        call; False returns the prefix beam search result as it is, the semantics of the reference's numpy
        BeamSearchDecoder (models/ctc/decoders/beam_search_decoder.py) the device search is pinned to."""
        assert isinstance(beam_width, int), "beam_width must be integer."
        assert beam_width >= 1, "beam_width must be >= 1"
        logits = logits.contiguous()
        seq = torch.as_tensor(inputs_seq_len, dtype=torch.int32, device=logits.device)
        if beam_width == 1:
            lab, n = ops.ctc_greedy_decode(logits, seq)
        else:
            lab, n, _ = ops.ctc_beam_decode(logits, seq, beam_width=beam_width)
        lab, n = lab.cpu().numpy(), n.cpu().numpy()
        if beam_width > 1 and merge_repeated:
            lab, n = lab.copy(), n.copy()
            for b in range(lab.shape[0]):
                row = lab[b, :int(n[b])]
                keep = np.ones(len(row), dtype=bool)
                keep[1:] = row[1:] != row[:-1]
                kept = row[keep]
                lab[b, :len(kept)] = kept
                n[b] = len(kept)
        indices, values = [], []
        for b in range(lab.shape[0]):
            for i in range(int(n[b])):
                indices.append([b, i])
                values.append(lab[b, i])
        dense_shape = [lab.shape[0], int(n.max()) if len(n) else 0]
        return [np.array(indices, dtype=np.int64).reshape(-1, 2), np.array(values, dtype=np.int32),
                np.array(dense_shape, dtype=np.int64)]

    def posteriors(self, logits, blank_prior=1):
        """ctc.py:354-380: softmax over classes on the batch-major flattening [B*T, C]."""
        lb = logits.transpose(0, 1).contiguous()
        return ops.softmax_rows(lb.view(-1, self.num_classes))

    def compute_ler(self, decode_op, labels):
        """ctc.py:382-398: mean normalised edit distance between decode result and labels
        (both SparseTensor triples)."""
        B = int(np.asarray(decode_op[2])[0])
        from ...utils.io.labels.sparsetensor import sparse_to_flat
        hv, ho, _ = sparse_to_flat(decode_op, B)
        rv, ro, _ = sparse_to_flat(labels, B)
        hyps = [hv[ho[b]:ho[b + 1]] for b in range(B)]
        refs = [rv[ro[b]:ro[b + 1]] for b in range(B)]
        return _ler(hyps, refs)

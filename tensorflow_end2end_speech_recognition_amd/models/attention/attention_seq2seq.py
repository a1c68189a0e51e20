"""Attention-based encoder-decoder -- mirror of models/attention/attention_seq2seq.py:25-725
(class AttentionSeq2Seq) on the HIP path.

Same constructor arguments and methods: create_placeholders, compute_loss(inputs, labels,
inputs_seq_len, labels_seq_len, keep_prob_encoder, keep_prob_decoder, keep_prob_embedding) ->
(total_loss, logits [B,T_out,C+2], decoder_outputs_train, decoder_outputs_infer), train
(ModelBase), decode(train, infer) -> (ids_train, ids_infer), compute_ler.

Graph restated (eager): BLSTM encoder -> InitialStateBridge (bridge.py:128-151) -> teacher-forced
AttentionDecoder loop (attention_decoder.py:256-295, dynamic_decoder.py:148-197) -> masked
sequence loss (:625-637).  The recurrence (cell + attention) runs step by step; everything that
does not feed back (attentional vector, output layer, all weight gradients) is batched over the
decoder steps in single GEMMs.

Reference quirk Q1 is reproduced by default (it IS the reference graph): location / hybrid attention see a
zero "previous alpha", so their location features reduce to the W_filter bias; `filter` and
W_filter/weights therefore receive zero gradient, W_keys of 'location' too (Q6).
`prev_alpha='carry'` (an extension keyword, default 'zeros') runs the recurrence attention_layer.py:191-265 was
written to express: the previous step's weights go through conv1d([201|200,1,10], SAME) -> W_filter inside the
energy kernel (csrc/attention.hip asr_att_loc_energy_*), with gradients into `filter`, W_filter and, through the
previous step's softmax, everything upstream.
"""
import os as _os

import numpy as np
import torch

from ... import ops
from ..._lib import ASR_BF16, ASR_F32
from ...utils.evaluation.edit_distance import compute_ler as _ler
from ...utils.parameter import ParamStore
from ..ctc.ctc import Placeholder, truncated_normal
from ..encoders.load_encoder import load as load_encoder
from ..model_base import ModelBase
from .bridge import InitialStateBridge
from .decoders import attention_layer as AL
from .decoders.attention_decoder import (AttentionDecoder, AttentionDecoderOutput, GreedyEmbeddingHelper,  # noqa: F401
                                         LSTMDecoderCell, TrainingHelper)
from .decoders.attention_layer import AttentionLayer

D = 'attention_decoder/decoder/'
AT = D + 'attention_layer/'


ATT_BWD_BF16 = _os.environ.get('ASR_ATT_BWD_BF16', '1') != '0'


class AttentionSeq2Seq(ModelBase):

    def __init__(self, input_size, encoder_type, encoder_num_units, encoder_num_layers, encoder_num_proj,
                 attention_type, attention_dim, decoder_type, decoder_num_units, decoder_num_layers,
                 embedding_dim, num_classes, sos_index, eos_index, max_decode_length,
                 lstm_impl='LSTMBlockCell', use_peephole=True, splice=1, parameter_init=0.1,
                 clip_grad_norm=5.0, clip_activation_encoder=50, clip_activation_decoder=50,
                 weight_decay=0.0, time_major=True, sharpening_factor=1.0, logits_temperature=1.0,
                 sigmoid_smoothing=False, name='attention', dtype='f32', device='cuda:0', seed=0,
                 _extra_vars=None, prev_alpha='zeros'):
        super(AttentionSeq2Seq, self).__init__()
        assert input_size % 3 == 0, 'input_size must be divisible by 3 (+ delta, double delta features).'
        assert splice % 2 == 1, 'splice must be the odd number'
        assert clip_grad_norm > 0, 'clip_grad_norm must be larger than 0.'
        assert weight_decay >= 0, 'weight_decay must not be a negative value.'
        AL.check_attention_type(attention_type)
        if prev_alpha not in ('zeros', 'carry'):
            raise ValueError("prev_alpha is 'zeros' (the reference's effective graph) or 'carry'")
        # carried weights only exist for the two types that have location features
        self.carry_alpha = prev_alpha == 'carry' and attention_type in AL.HAS_FILTER
        if decoder_type != 'lstm':
            raise TypeError('decoder_type is "lstm" or "gru".') if decoder_type != 'gru' else \
                NotImplementedError('GRU decoder (crashes in the reference too, attention_seq2seq.py:364-365)')
        if encoder_type not in ('blstm',):
            raise NotImplementedError
        self.input_size, self.splice = input_size, splice
        self.encoder_type = encoder_type
        self.encoder_num_units, self.encoder_num_proj = encoder_num_units, encoder_num_proj
        self.encoder_num_layers = encoder_num_layers
        self.lstm_impl, self.use_peephole = lstm_impl, use_peephole
        self.attention_type, self.attention_dim = attention_type, attention_dim
        self.sharpening_factor, self.sigmoid_smoothing = sharpening_factor, sigmoid_smoothing
        self.decoder_type, self.decoder_num_units = decoder_type, decoder_num_units
        self.decdoder_num_layers = decoder_num_layers
        self.embedding_dim = embedding_dim
        self.num_classes = num_classes + 2
        self.sos_index, self.eos_index = sos_index, eos_index
        self.max_decode_length = max_decode_length
        self.logits_temperature = logits_temperature
        self.use_beam_search = False
        self.parameter_init = parameter_init
        self.clip_grad_norm = clip_grad_norm
        self.clip_activation_encoder = clip_activation_encoder
        self.clip_activation_decoder = clip_activation_decoder
        self.weight_decay = weight_decay
        self.time_major = time_major
        self.name = name
        self.summaries_train, self.summaries_dev = [], []
        self.inputs_pl_list, self.labels_pl_list = [], []
        self.inputs_seq_len_pl_list, self.labels_seq_len_pl_list = [], []
        self.keep_prob_encoder_pl_list, self.keep_prob_decoder_pl_list = [], []
        self.keep_prob_embedding_pl_list = []
        self.labels_st_true_pl_list, self.labels_st_pred_pl_list = [], []
        self.dtype = ops.dtype_id(dtype)
        self.device = torch.device(device)
        self.seed = seed
        self._calls = 0

        rng = np.random.RandomState(seed)
        st = self.store = ParamStore(self.device)
        self.encoder = load_encoder(encoder_type)(
            num_units=encoder_num_units, num_proj=None, num_layers=encoder_num_layers, lstm_impl=lstm_impl,
            use_peephole=use_peephole, parameter_init=parameter_init, clip_activation=clip_activation_encoder,
            time_major=True, dtype=self.dtype)
        E2 = self.enc_dim = self.encoder.build(st, input_size * splice, rng, scope_prefix='encoder/')
        H, U, A, Em, C2 = encoder_num_units, decoder_num_units, attention_dim, embedding_dim, self.num_classes
        u = lambda *s: rng.uniform(-parameter_init, parameter_init, size=s)
        tn = lambda *s: truncated_normal(rng, parameter_init, s)
        st.declare('output_embedding/W_embedding', (C2, Em), u(C2, Em))
        st.declare('bridge/fully_connected/weights', (4 * H, 2 * U), tn(4 * H, 2 * U))
        st.declare('bridge/fully_connected/biases', (2 * U,), np.zeros(2 * U))
        self.dec_in_dim = Em + E2 + U
        st.declare(D + 'lstm_cell/kernel', (self.dec_in_dim, 4 * U), u(self.dec_in_dim, 4 * U))
        st.declare(D + 'lstm_cell/bias', (4 * U,), np.zeros(4 * U))
        if use_peephole:
            for n in ('w_i_diag', 'w_f_diag', 'w_o_diag'):
                st.declare(D + 'lstm_cell/' + n, (U,), u(U))
        at = attention_type
        self.att_mode = 0 if at in AL.ADDITIVE else 1
        if at == 'luong_dot' and E2 != U:
            raise ValueError('encoder_num_units and decoder_num_units must be the same size.')
        self.key_dim = {'luong_dot': E2, 'luong_general': U}.get(at, A)
        if at == 'luong_concat':
            st.declare(AT + 'W_concat/weights', (E2 + U, A), tn(E2 + U, A))
        elif at in AL.HAS_QUERY_FC:
            st.declare(AT + 'W_query/weights', (U, A), tn(U, A))
        if at in AL.HAS_KEYS_FC:
            st.declare(AT + 'W_keys/weights', (E2, self.key_dim), tn(E2, self.key_dim))
            if at not in ('dot_product', 'luong_general'):
                st.declare(AT + 'W_keys/biases', (A,), np.zeros(A))
        if at in AL.HAS_FILTER:
            taps = 201 if at == 'location' else 200
            st.declare(AT + 'filter', (taps, 1, 10),
                       truncated_normal(rng, 0.1 if at == 'location' else parameter_init, (taps, 1, 10)))
            st.declare(AT + 'W_filter/weights', (10, A), tn(10, A))
            st.declare(AT + 'W_filter/biases', (A,), np.zeros(A))
        if at in AL.ADDITIVE:
            lim = np.sqrt(3.0 / A)      # tf.get_variable default (glorot_uniform) for a [A] vector
            st.declare(AT + 'v_a', (A,), rng.uniform(-lim, lim, size=(A,)))
        st.declare(D + 'attentional_vector/weights', (U + E2, U), tn(U + E2, U))
        st.declare(D + 'output_layer/weights', (U, C2), tn(U, C2))
        st.declare(D + 'output_layer/biases', (C2,), np.zeros(C2))
        for name_, shape, init in (_extra_vars(rng, E2) if _extra_vars else []):
            st.declare(name_, shape, init)
        st.finalize()
        self._tape = None

    # ------------------------------------------------------------------ helpers
    def create_placeholders(self):
        for lst, nm in ((self.inputs_pl_list, 'input'), (self.labels_pl_list, 'labels'),
                        (self.inputs_seq_len_pl_list, 'inputs_seq_len'),
                        (self.labels_seq_len_pl_list, 'labels_seq_len'),
                        (self.keep_prob_encoder_pl_list, 'keep_prob_encoder'),
                        (self.keep_prob_decoder_pl_list, 'keep_prob_decoder'),
                        (self.keep_prob_embedding_pl_list, 'keep_prob_embedding')):
            lst.append(Placeholder(nm))

    def _w_cell(self):
        """The decoder cell's kernel as the products see it: itself in an fp32 model; rounded to bf16 (straight-through:
        the gradient lands on the fp32 variable) in a bf16-operand model -- the per-step products stream 13 MB of it at
        cfg D widths, half of that as bf16 (asr_lstm_cell_gemm_*_h); the copy made here keeps every other path that
        multiplies with it (class surface, shapes the skinny kernels do not take) on the same values."""
        W = self.store[D + 'lstm_cell/kernel']
        return W.to(torch.bfloat16).to(torch.float32) if self.dtype == ASR_BF16 else W

    def _peep(self):
        st = self.store
        if not self.use_peephole:
            return None
        return torch.stack([st[D + 'lstm_cell/w_i_diag'], st[D + 'lstm_cell/w_f_diag'],
                            st[D + 'lstm_cell/w_o_diag']]).contiguous()

    # key / query projection matrices (and their gradient views); luong_concat slices W_concat
    def _wk(self, grad=False):
        st = self.store
        get = st.g if grad else st.__getitem__
        if self.attention_type == 'luong_concat':
            return get(AT + 'W_concat/weights')[:2 * self.encoder_num_units]
        return get(AT + 'W_keys/weights')

    def _wq(self, grad=False):
        st = self.store
        get = st.g if grad else st.__getitem__
        if self.attention_type == 'luong_concat':
            return get(AT + 'W_concat/weights')[2 * self.encoder_num_units:]
        return get(AT + 'W_query/weights')

    def _encode(self, inputs, inputs_seq_len, keep_prob_encoder, is_training):
        rs = None
        if is_training and float(keep_prob_encoder) < 1.0:
            self._calls += 1
            rs = (self.seed, self._calls << 40)
        _, final = self.encoder(inputs, inputs_seq_len, float(keep_prob_encoder), is_training, rng_state=rs)
        enc = self.encoder._out_tm.contiguous()            # [T,Bp,E2] fp32, time-major
        lay = self.encoder.layers[-1]
        return enc, self.encoder.seq_len_padded

    def _keys(self, enc):
        st, at = self.store, self.attention_type
        T, Bp, E2 = enc.shape
        if at in AL.USES_KEYS:
            b = st[AT + 'W_keys/biases'] if (AT + 'W_keys/biases') in st.views else None
            return ops.gemm(enc.view(T * Bp, E2), self._wk(), bias=b).view(T, Bp, self.key_dim)
        if at == 'luong_dot':
            return enc
        return None                                          # 'location': keys unused (Q6)

    def _query(self, s):
        st, at = self.store, self.attention_type
        if at in AL.HAS_QUERY_FC:
            b = st[AT + 'W_filter/biases'] if at in AL.HAS_FILTER else None
            return ops.gemm(s, self._wq(), bias=b)
        return s                                             # luong_*: the decoder state itself

    def _bridge(self, final_c, final_h, B):
        """final_c/final_h [2,Bp,H] -> (bi [Bp,4H], c0, h0) through InitialStateBridge (bridge.py:128-151: the final
        (c, h) of the forward then the backward direction, flattened and concatenated, one FC, split)."""
        class _Enc(object):
            final_state = ((final_c[0], final_h[0]), (final_c[1], final_h[1]))
        U = self.decoder_num_units
        br = InitialStateBridge(_Enc, (U, U), self.parameter_init, store=self.store)
        c0, h0 = br()
        return br.bridge_input(), c0, h0

    def attention_layer(self, time_major_inputs=True):
        """The model's AttentionLayer (attention_seq2seq.py:413-430), bound to its variables."""
        return AttentionLayer(self.attention_type, self.attention_dim, self.parameter_init, self.sharpening_factor,
                              self.sigmoid_smoothing, mode='infer', store=self.store,
                              prev_alpha='carry' if self.carry_alpha else 'zeros', time_major_inputs=time_major_inputs)

    # ------------------------------------------------------------------ forward
    def compute_loss(self, inputs, labels, inputs_seq_len, labels_seq_len, keep_prob_encoder,
                     keep_prob_decoder, keep_prob_embedding, scope=None, is_training=True, **joint):
        dev, st = self.device, self.store
        # nothing below may drain the stream (pageable uploads and device read-backs do): the host has to run ahead of the
        # device for the step to be device-bound -- ops.to_device stages through pinned memory, ops.host_ints remembers
        # the host copy of a device vector it has seen
        self.encoder._lens_host = ops.host_ints(inputs_seq_len)
        inputs = ops.to_device(inputs, torch.float32, dev)
        isl = ops.to_device(inputs_seq_len, torch.int32, dev)
        labels_np = (ops.host_ints(labels) if torch.is_tensor(labels) else np.asarray(labels)).astype(np.int64)
        lsl_np = (ops.host_ints(labels_seq_len) if torch.is_tensor(labels_seq_len) else np.asarray(labels_seq_len)).astype(np.int64)
        B = inputs.shape[0]
        enc, seq_p = self._encode(inputs, isl, keep_prob_encoder, is_training)
        T, Bp, E2 = enc.shape
        # joint model: the CTC head (logits GEMM, alpha / beta recursions -- one wave per utterance, ~1.6 ms of latency
        # at T = 1600 -- and the gradient) needs only the encoder output, so it runs on the side lane beside the decoder
        # loop instead of after it; the main stream waits for it where the two losses meet
        lam = joint.get('lambda_weight')
        ctc_pending = ctc_event = None
        if lam is not None:
            with ops.side_lane(dev, keep=(enc, seq_p), lane=1):
                ctc_pending = self._ctc_head_launch(enc, seq_p, joint['ctc_labels'], B, lam, is_training)
                ctc_event = ops.stream_event()
        U, Em, C2 = self.decoder_num_units, self.embedding_dim, self.num_classes
        To = int(lsl_np.max()) - 1
        Lmax = labels_np.shape[1]
        # host-side step tables: ids fed at step k, targets, live mask, loss weights  ([To,Bp])
        ids_in = np.full((To, Bp), self.eos_index, dtype=np.int32)
        tgt = np.zeros((To, Bp), dtype=np.int32)
        live = np.zeros((To, Bp), dtype=np.float32)
        ids_in[:, :B] = labels_np[:, :To].T
        tgt[:, :B] = labels_np[:, 1:To + 1].T
        live[:, :B] = (np.arange(To)[:, None] < (lsl_np - 1)[None, :]).astype(np.float32)
        ids_d = ops.to_device(ids_in, torch.int32, dev)
        tgt_d = ops.to_device(tgt, torch.int32, dev)
        live_d = ops.to_device(live, torch.float32, dev)
        emb = ops.embedding_gather(st['output_embedding/W_embedding'], ids_d.view(-1)).view(To, Bp, Em)
        emb_mask = None
        if is_training and float(keep_prob_embedding) < 1.0:
            self._calls += 1
            emb_mask = ops.dropout_mask(emb.shape, keep_prob_embedding, self.seed + 1, self._calls << 40, dev)
            emb = ops.apply_mask(emb, emb_mask)
        cf, hf = self.encoder._final_ch
        bi, c, h = self._bridge(cf, hf, B)
        keys = self._keys(enc)
        # what the per-step context / d-alpha kernels stream: the bf16 operand copy the encoder keeps (same
        # values as enc, half the bytes) for bf16-operand models, enc itself otherwise
        enc_att = self.encoder._out_op.contiguous() if self.dtype == ASR_BF16 else enc
        peep = self._peep()
        Din = self.dec_in_dim
        dec_in = torch.empty((To, Bp, Din), dtype=torch.float32, device=dev)
        av_in = torch.empty((To, Bp, U + E2), dtype=torch.float32, device=dev)
        ctx = torch.zeros((Bp, E2), dtype=torch.float32, device=dev)
        W_cell, b_cell = self._w_cell(), st[D + 'lstm_cell/bias']
        v = st[AT + 'v_a'] if self.att_mode == 0 else None
        alpha_all = torch.empty((To, Bp, T), dtype=torch.float32, device=dev)   # one slab: d_enc GEMMs read it strided
        use_ddrop = is_training and float(keep_prob_decoder) < 1.0
        dec_in[:, :, :Em].copy_(emb.view(To, Bp, Em))          # the embedded inputs of all steps at once
        dmask_all = None
        if use_ddrop:                                            # one launch for the masks of every step
            self._calls += 1
            dmask_all = ops.dropout_mask((To, Bp, U), keep_prob_decoder, self.seed + 2, self._calls << 40, dev)
        # sigmoid smoothing (attention_layer.py:92-96): the per-step normaliser is kept for the backward
        snorm_all = torch.empty((To, Bp), dtype=torch.float32, device=dev) if self.sigmoid_smoothing else None
        alpha_zero = torch.zeros((Bp, T), dtype=torch.float32, device=dev) if self.carry_alpha else None
        dec_in[0, :, Em:Em + E2].copy_(ctx)
        dec_in[0, :, Em + E2:].copy_(h)
        # the To decoder steps (cell-input GEMM -> cell -> query FC -> energies -> softmax + context, each kernel writing
        # where the next one reads) are issued by ONE native call: a host loop over ~9 launches per step is host-bound
        A = self.key_dim                                     # width of keys / query (== U for the luong_* types)
        has_q = self.attention_type in AL.HAS_QUERY_FC
        c_all = torch.empty((To + 1, Bp, U), dtype=torch.float32, device=dev)
        h_all = torch.empty((To + 1, Bp, U), dtype=torch.float32, device=dev)
        c_all[0].copy_(c)
        h_all[0].copy_(h)
        loop = dict(To=To, B=Bp, T=T, U=U, Em=Em, E2=E2, A=A, att_mode=self.att_mode, has_query_fc=int(has_q),
                    carry_alpha=int(self.carry_alpha), taps=int(st[AT + 'filter'].shape[0]) if self.carry_alpha else 0,
                    enc_dtype=ops.dtype_id(enc_att.dtype), forget_bias=1.0,
                    cell_clip=float(self.clip_activation_decoder or 0.0), sharpening=float(self.sharpening_factor),
                    W_cell=W_cell, b_cell=b_cell, cell_bf16=(self.dtype == ASR_BF16), peep=peep,
                    W_q=self._wq() if has_q else None,
                    b_q=st[AT + 'W_filter/biases'] if (has_q and self.attention_type in AL.HAS_FILTER) else None,
                    v=v, keys=keys, enc=enc_att, seq_len=seq_p,
                    filt=st[AT + 'filter'] if self.carry_alpha else None,
                    wfil=st[AT + 'W_filter/weights'] if self.carry_alpha else None, alpha_zero=alpha_zero,
                    live=live_d, dmask=dmask_all, dec_in=dec_in, av_in=av_in, alpha_all=alpha_all, snorm_all=snorm_all,
                    gates_all=torch.empty((To, Bp, 4 * U), dtype=torch.float32, device=dev),
                    craw_all=torch.empty((To, Bp, U), dtype=torch.float32, device=dev), c_all=c_all, h_all=h_all,
                    qz_all=torch.empty((To, Bp, A), dtype=torch.float32, device=dev))
        ops.att_decoder_fwd(loop)
        av = ops.tanh_fwd(ops.gemm(av_in.view(To * Bp, U + E2), st[D + 'attentional_vector/weights']))
        logits2d = ops.gemm(av, st[D + 'output_layer/weights'], bias=st[D + 'output_layer/biases'])
        logits2d = ops.apply_mask(logits2d, live_d.view(-1, 1).expand(To * Bp, C2).contiguous())  # impute_finished
        inv_t = 1.0 / float(self.logits_temperature)
        lt = ops.scale_(logits2d.clone(), inv_t) if self.logits_temperature != 1.0 else logits2d
        wsum = float(live.sum())
        seq_scale = (1.0 - lam) if lam is not None else 1.0
        row_loss, dlogits = ops.seq_xent(lt, tgt_d.view(-1), live_d.view(-1), 1e-10,
                                         seq_scale * inv_t / (wsum + 1e-12), want_grad=is_training)
        seq_loss = row_loss.sum() / (wsum + 1e-12)
        total = seq_loss
        ctc_logits = None
        ctc_tape = None
        if lam is not None:
            ops.wait_event(ctc_event)
            if not is_training:
                # no backward pass will join the side lane: release the encoder outputs it holds for the CTC head now
                # (a dev-loss loop would otherwise pin one [T,B,2H] tensor per evaluated batch)
                ops.join_side(dev)
            ctc_logits, ctc_mean, ctc_tape = self._ctc_head_finish(ctc_pending, B)
            total = (1.0 - lam) * seq_loss + lam * ctc_mean
        if self.weight_decay > 0:
            l2 = torch.zeros((), dtype=torch.float32, device=dev)
            ops.weight_decay(None, st.flat, st.plan, st.decay_mask, self.weight_decay, l2_out=l2)
            total = total + l2
        self.sequence_loss = seq_loss
        logits_bm = logits2d.view(To, Bp, C2)[:, :B].transpose(0, 1)        # [B,To,C2] (batch-major)
        ids_train = ops.argmax_rows(logits2d).view(To, Bp)[:, :B].t()
        # dynamic_decode(impute_finished=True): every emitted field is zero once a row has finished
        lv = live_d[:, :B].t().unsqueeze(2)                                  # [B,To,1]
        alphas = alpha_all[:, :B].transpose(0, 1) * lv
        out_train = AttentionDecoderOutput(logits=logits_bm, predicted_ids=ids_train * live_d[:, :B].t().int(),
                                           decoder_output=av.view(To, Bp, U)[:, :B].transpose(0, 1) * lv,
                                           attention_weights=alphas,
                                           context_vector=av_in[:, :B, U:].transpose(0, 1) * lv)
        inf_args = (inputs, isl)
        out_infer = AttentionDecoderOutput(lazy=lambda: self._decode_infer(*inf_args))
        if is_training:
            self._tape = dict(B=B, To=To, enc=enc, seq_p=seq_p, keys=keys, dec_in=dec_in, av_in=av_in, av=av,
                              loop=loop, dlogits=dlogits, ids=ids_d, emb_mask=emb_mask, live=live_d, bi=bi,
                              peep=peep, ctc=ctc_tape, alpha_all=alpha_all, enc_att=enc_att)
        else:
            self._tape = None
        total._asr_model = self
        if lam is not None:
            return total, logits_bm, ctc_logits, out_train, out_infer
        return total, logits_bm, out_train, out_infer

    def _ctc_head_launch(self, *a, **k):
        raise NotImplementedError

    def _ctc_head_finish(self, *a, **k):
        raise NotImplementedError

    # ------------------------------------------------------------------ backward
    def _backward(self):
        if self._tape is None:
            raise RuntimeError('train()/compute_gradients() needs a preceding compute_loss(is_training=True)')
        tp, st, dev = self._tape, self.store, self.device
        enc, seq_p, keys, dec_in, av_in, av = tp['enc'], tp['seq_p'], tp['keys'], tp['dec_in'], tp['av_in'], tp['av']
        loop, To, live = tp['loop'], tp['To'], tp['live']
        T, Bp, E2 = enc.shape
        U, Em, C2, A = self.decoder_num_units, self.embedding_dim, self.num_classes, self.key_dim
        at = self.attention_type
        st.grad.zero_()
        # ---- output layer + attentional vector, all steps at once
        dlogits = tp['dlogits']
        W_out, W_av = st[D + 'output_layer/weights'], st[D + 'attentional_vector/weights']
        dav_pre = ops.tanh_bwd(ops.gemm(dlogits, W_out, transB=True), av)
        # bf16-operand model: the batched products of the BACKWARD pass outside the loop (everything below that multiplies
        # two [steps x utterances]- or [frames x utterances]-long arrays) round their operands to bf16 like the encoder's
        # dx / weight-gradient products do, and run on the bf16 matrix pipe instead of the 1/16-rate fp32 one: 2 ms of
        # the cfg-D-shaped step (profiles/r05_cfgD_timeline.md).  The forward keeps its fp32 products (the oracle's
        # rounding points, oracle/attention.py).  ASR_ATT_BWD_BF16=0: fp32 as in an fp32 model.
        b16 = self.dtype == ASR_BF16 and ATT_BWD_BF16
        dav16 = dav_pre.to(torch.bfloat16) if b16 else None
        # weight gradients are nobody's input until the optimizer: they go to the side lane (joined by encoder.backward
        # / the end of this function) and run beside the reverse loop and the encoder's BPTT
        with ops.side_lane(dev, keep=(av, dlogits, av_in, dav_pre, dav16), lane=1):
            ops.gemm(av, dlogits, transA=True, out=st.g(D + 'output_layer/weights'))
            ops.colsum(dlogits, out=st.g(D + 'output_layer/biases'))
            if b16:
                ops.gemm(av_in.view(To * Bp, U + E2).to(torch.bfloat16), dav16, transA=True,
                         out=st.g(D + 'attentional_vector/weights'))
            else:
                ops.gemm(av_in.view(To * Bp, U + E2), dav_pre, transA=True, out=st.g(D + 'attentional_vector/weights'))
        # gradient of the attentional vector's inputs, as two contiguous arrays (cell output | context): a step's rows
        # are then the buffers the loop works in, not slices that have to be copied out first
        if b16:
            W16 = W_av.to(torch.bfloat16)
            dav_cell = ops.gemm(dav16, W16[:U], transB=True, out_dtype=torch.float32).view(To, Bp, U)
            dav_ctx = ops.gemm(dav16, W16[U:], transB=True, out_dtype=torch.float32).view(To, Bp, E2)
        else:
            dav_cell = ops.gemm(dav_pre, W_av[:U], transB=True).view(To, Bp, U)
            dav_ctx = ops.gemm(dav_pre, W_av[U:], transB=True).view(To, Bp, E2)
        # ---- the recurrence, backwards
        # d_enc starts as the CTC head's part (joint model) -- one GEMM that needs nothing from the decoder, issued on
        # side lane 2 so that it runs beside the reverse loop; the first accumulation into d_enc waits for it
        ctc_denc_event, ctc_denc_done = None, False
        if tp['ctc'] is not None and at != 'luong_dot':
            denc = torch.empty_like(enc)
            with ops.side_lane(dev, keep=(enc, denc), lane=2):
                self._ctc_head_backward(tp['ctc'], enc, denc, accumulate=False)
                ctc_denc_event, ctc_denc_done = ops.stream_event(), True
        else:
            denc = torch.zeros_like(enc)
        dkeys = None
        if at in AL.USES_KEYS:
            dkeys = torch.zeros_like(keys)
        elif at == 'luong_dot':
            dkeys = denc
        dpre_all = torch.empty((To, Bp, 4 * U), dtype=torch.float32, device=dev)
        dqz_all = torch.empty((To, Bp, A), dtype=torch.float32, device=dev)
        dv_all = torch.empty_like(dqz_all) if self.att_mode == 0 else None
        dpeep_all = torch.empty((To, Bp, 3 * U), dtype=torch.float32, device=dev) if self.use_peephole else None
        cell_out_all = av_in[:, :, :U]
        d_in_all = torch.empty((To, Bp, Em + E2 + U), dtype=torch.float32, device=dev)   # d loss / d cell input
        dctx_all = torch.empty((To, Bp, E2), dtype=torch.float32, device=dev)
        dwfil_rows = dfilt_rows = None
        if self.carry_alpha:
            filt, wfil = st[AT + 'filter'], st[AT + 'W_filter/weights']
            dwfil_rows = torch.empty((Bp,) + tuple(wfil.shape), dtype=torch.float32, device=dev)
            dfilt_rows = torch.empty((Bp, filt.shape[0], filt.shape[2]), dtype=torch.float32, device=dev)
        dc_next = torch.empty((Bp, U), dtype=torch.float32, device=dev)
        dh_next = torch.empty((Bp, U), dtype=torch.float32, device=dev)
        # the reverse loop, native as the forward one: per step  d ctx = attentional-vector part + the next step's
        # cell-input part -> softmax / context backward -> energy backward (d_enc += alpha (x) dctx is NOT done per
        # step: alpha and dctx of all steps are kept and contracted once per utterance below) -> query FC backward ->
        # dropout mask -> cell backward -> cell-input GEMM backward; everything lands in the rows of per-step arrays
        loop.update(dav_cell=dav_cell, dav_ctx=dav_ctx, dctx_all=dctx_all, dpre_all=dpre_all, dqz_all=dqz_all,
                    dv_all=dv_all, dpeep_all=dpeep_all, d_in_all=d_in_all, dkeys=dkeys, dwfil_rows=dwfil_rows,
                    dfilt_rows=dfilt_rows, dc0=dc_next, dh0=dh_next)
        ops.att_decoder_bwd(loop)
        # ---- d_enc[:, b, :] += alpha_b^T [T, To] . dctx_b [To, 2H]   (context path of all steps at once)
        alpha_all = tp['alpha_all']
        ops.wait_event(ctc_denc_event)
        if b16:
            # (rows of 8-element vectors: the frame axis padded to a multiple of 8 so that every utterance's block is aligned)
            Tp8 = (T + 7) // 8 * 8
            a16 = torch.zeros((To, Bp, Tp8), dtype=torch.bfloat16, device=dev)
            a16[:, :, :T].copy_(alpha_all)
            c16 = dctx_all.to(torch.bfloat16)
            for b in range(tp['B']):
                ops.gemm(a16[:, b, :T], c16[:, b, :], transA=True, out=denc[:, b, :], accumulate=True)
        else:
            for b in range(tp['B']):
                ops.gemm(alpha_all[:, b, :], dctx_all[:, b, :], transA=True, out=denc[:, b, :], accumulate=True)
        dq2d = dqz_all.view(To * Bp, -1)
        dk2d = dkeys.view(T * Bp, -1) if at in AL.USES_KEYS else None
        dk16 = dk2d.to(torch.bfloat16) if (b16 and dk2d is not None) else None
        if dk16 is not None:
            ops.gemm(dk16, self._wk().to(torch.bfloat16), transB=True, out=denc.view(T * Bp, E2), accumulate=True)
        elif dk2d is not None:
            ops.gemm(dk2d, self._wk(), transB=True, out=denc.view(T * Bp, E2), accumulate=True)
        # bridge
        dinit = torch.cat([dc_next, dh_next], dim=1).contiguous()
        dbi = ops.gemm(dinit, st['bridge/fully_connected/weights'], transB=True)
        H = self.encoder_num_units
        dcf = torch.stack([dbi[:, :H], dbi[:, 2 * H:3 * H]]).contiguous()
        dhf = torch.stack([dbi[:, H:2 * H], dbi[:, 3 * H:]]).contiguous()
        # ---- weight gradients of everything inside the loop, batched over the steps -- on the side lane
        side_keep = [t for t in (dec_in, dpre_all, dpeep_all, dqz_all, dv_all, dwfil_rows, dfilt_rows, dkeys, enc,
                                 d_in_all, av_in, dinit, tp['bi'], tp['ids'], tp['emb_mask'], dk16, tp['enc_att'])
                     if t is not None]
        with ops.side_lane(dev, keep=side_keep, lane=1):
            if b16:     # 84 GFLOP at cfg D: 1.38 ms on the fp32 pipe beside the top encoder layer's BPTT
                ops.gemm(dec_in.view(To * Bp, -1).to(torch.bfloat16), dpre_all.view(To * Bp, 4 * U).to(torch.bfloat16),
                         transA=True, out=st.g(D + 'lstm_cell/kernel'))
            else:
                ops.gemm(dec_in.view(To * Bp, -1), dpre_all.view(To * Bp, 4 * U), transA=True,
                         out=st.g(D + 'lstm_cell/kernel'))
            ops.colsum(dpre_all.view(To * Bp, 4 * U), out=st.g(D + 'lstm_cell/bias'))
            if self.use_peephole:
                dp = ops.colsum(dpeep_all.view(To * Bp, 3 * U))
                st.g(D + 'lstm_cell/w_i_diag').copy_(dp[:U])
                st.g(D + 'lstm_cell/w_f_diag').copy_(dp[U:2 * U])
                st.g(D + 'lstm_cell/w_o_diag').copy_(dp[2 * U:])
            if at in AL.HAS_QUERY_FC:
                ops.gemm(cell_out_all.reshape(To * Bp, U), dq2d, transA=True, out=self._wq(grad=True))
            if at in AL.HAS_FILTER:
                ops.colsum(dq2d, out=st.g(AT + 'W_filter/biases'))
                if self.carry_alpha:       # per-utterance sums over the steps -> sum over the batch
                    ops.colsum(dwfil_rows.view(Bp, -1), out=st.g(AT + 'W_filter/weights').view(-1))
                    ops.colsum(dfilt_rows.view(Bp, -1), out=st.g(AT + 'filter').view(-1))
            if self.att_mode == 0:
                ops.colsum(dv_all.view(To * Bp, -1), out=st.g(AT + 'v_a'))
            if dk2d is not None:
                if dk16 is not None:
                    ops.gemm(tp['enc_att'].view(T * Bp, E2), dk16, transA=True, out=self._wk(grad=True))
                else:
                    ops.gemm(enc.view(T * Bp, E2), dk2d, transA=True, out=self._wk(grad=True))
                if (AT + 'W_keys/biases') in st.views:
                    ops.colsum(dk2d, out=st.g(AT + 'W_keys/biases'))
            # embedding
            demb_all = d_in_all[:, :, :Em].contiguous()
            if tp['emb_mask'] is not None:
                demb_all = ops.apply_mask(demb_all, tp['emb_mask'])
            ops.embedding_scatter(demb_all.view(To * Bp, Em), tp['ids'].view(-1), C2,
                                  st.g('output_embedding/W_embedding'))
            ops.gemm(tp['bi'], dinit, transA=True, out=st.g('bridge/fully_connected/weights'))
            ops.colsum(dinit, out=st.g('bridge/fully_connected/biases'))
        if tp['ctc'] is not None and not ctc_denc_done:
            self._ctc_head_backward(tp['ctc'], enc, denc, accumulate=True)
        self.encoder.backward(denc, d_final=(dcf, dhf))
        ops.join_side(dev)
        if self.weight_decay > 0:
            ops.weight_decay(st.grad, st.flat, st.plan, st.decay_mask, self.weight_decay)
        self._tape = None

    def _ctc_head_backward(self, *a):
        raise NotImplementedError

    # ------------------------------------------------------------------ inference
    def _decode_infer(self, inputs, isl, native=True):
        """GreedyEmbeddingHelper decode (attention_seq2seq.py:462-509, the reference's in-graph while_loop): at most
        max_decode_length steps, impute_finished.  Returns the predicted ids [B, <= max_decode_length].
        native=True: ONE call (ops.att_decoder_infer -> asr_att_decoder_infer) issues every step -- decoder cell,
        attention, attentional vector, output layer, argmax, embedding of the chosen id, finished flags -- without the
        host looking at the device until the end (one read-back of the per-step live counts); native=False: the
        class-surface loop below (AttentionDecoder.step under dynamic_decode: one device sync per token), kept as the
        reference-shaped form and as the oracle of the native one (ids identical, tests/test_gpu_attention.py)."""
        if not native:
            return self._decode_infer_class_surface(inputs, isl)
        st, dev = self.store, self.device
        B = inputs.shape[0]
        enc, seq_p = self._encode(inputs, isl, 1.0, False)
        T, Bp, E2 = enc.shape
        cf, hf = self.encoder._final_ch
        _, c, h = self._bridge(cf, hf, B)
        U, Em, A = self.decoder_num_units, self.embedding_dim, self.key_dim
        To, Din = int(self.max_decode_length), self.dec_in_dim
        keys = self._keys(enc)
        enc_att = self.encoder._out_op.contiguous() if self.dtype == ASR_BF16 else enc
        has_q = self.attention_type in AL.HAS_QUERY_FC
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)          # noqa: E731
        dec_in = f32(To, Bp, Din)
        emb0 = ops.embedding_gather(st['output_embedding/W_embedding'],
                                    torch.full((Bp,), self.sos_index, dtype=torch.int32, device=dev))
        dec_in[0, :, :Em].copy_(emb0)
        dec_in[0, :, Em:Em + E2].zero_()
        dec_in[0, :, Em + E2:].copy_(h)
        c_all, h_all = f32(To + 1, Bp, U), f32(To + 1, Bp, U)
        c_all[0].copy_(c)
        h_all[0].copy_(h)
        live = torch.zeros((To + 1, Bp), dtype=torch.float32, device=dev)
        live[0, :B] = 1.0                                    # rows B.. are the zero-length padding of the batch tile
        loop = dict(To=To, B=Bp, T=T, U=U, Em=Em, E2=E2, A=A, att_mode=self.att_mode, has_query_fc=int(has_q),
                    carry_alpha=int(self.carry_alpha), taps=int(st[AT + 'filter'].shape[0]) if self.carry_alpha else 0,
                    enc_dtype=ops.dtype_id(enc_att.dtype), forget_bias=1.0,
                    cell_clip=float(self.clip_activation_decoder or 0.0), sharpening=float(self.sharpening_factor),
                    W_cell=self._w_cell(), b_cell=st[D + 'lstm_cell/bias'], cell_bf16=(self.dtype == ASR_BF16),
                    peep=self._peep(),
                    W_q=self._wq() if has_q else None,
                    b_q=st[AT + 'W_filter/biases'] if (has_q and self.attention_type in AL.HAS_FILTER) else None,
                    v=st[AT + 'v_a'] if self.att_mode == 0 else None, keys=keys, enc=enc_att, seq_len=seq_p,
                    filt=st[AT + 'filter'] if self.carry_alpha else None,
                    wfil=st[AT + 'W_filter/weights'] if self.carry_alpha else None,
                    alpha_zero=torch.zeros((Bp, T), dtype=torch.float32, device=dev) if self.carry_alpha else None,
                    live=live, dmask=None, dec_in=dec_in, av_in=f32(To, Bp, U + E2), alpha_all=f32(To, Bp, T),
                    snorm_all=f32(To, Bp) if self.sigmoid_smoothing else None,
                    gates_all=f32(1, Bp, 4 * U), craw_all=f32(1, Bp, U), c_all=c_all, h_all=h_all, qz_all=f32(1, Bp, A))
        out = ops.att_decoder_infer(loop, st[D + 'attentional_vector/weights'], st[D + 'output_layer/weights'],
                                    st[D + 'output_layer/biases'], st['output_embedding/W_embedding'], self.eos_index,
                                    n_live=B)
        counts = out['live_count'].cpu().numpy()             # the ONE synchronisation of the decode
        dead = np.flatnonzero(counts[:out['steps_issued'] + 1] == 0)
        n = int(dead[0]) if len(dead) else min(To, out['steps_issued'])
        self._infer_raw = dict(out, alpha=loop['alpha_all'], steps=n, B=B)     # un-imputed fields, for visualisation
        return out['ids'][:n, :B].t().contiguous()

    def _decode_infer_class_surface(self, inputs, isl):
        """The reference-shaped inference path: AttentionDecoder.step under dynamic_decode (one device sync per token)."""
        st, dev = self.store, self.device
        B = inputs.shape[0]
        enc, seq_p = self._encode(inputs, isl, 1.0, False)
        T, Bp, E2 = enc.shape
        cf, hf = self.encoder._final_ch
        _, c, h = self._bridge(cf, hf, B)
        layer = self.attention_layer(time_major_inputs=True)
        cell = LSTMDecoderCell(st, self.decoder_num_units, self.use_peephole, self.clip_activation_decoder,
                               kernel=self._w_cell())
        decoder = AttentionDecoder(cell, self.parameter_init, self.max_decode_length, self.num_classes, enc, seq_p, layer,
                                   time_major=False, mode='infer', store=st)
        decoder.live_rows = torch.arange(Bp, device=dev) < B      # rows B.. are the zero-length padding of the batch tile
        helper = GreedyEmbeddingHelper(st['output_embedding/W_embedding'],
                                       torch.full((Bp,), self.sos_index, dtype=torch.int32, device=dev), self.eos_index)
        outputs, _ = decoder((c, h), helper)
        return outputs.predicted_ids[:B]

    def infer(self, inputs, inputs_seq_len, native=True):
        """Greedy inference ids [B, <= max_decode_length] (numpy) for a batch of features -- what running the
        reference's `decode_op_infer` with a feed_dict of inputs / inputs_seq_len / keep_prob = 1 returns
        (examples/timit/metrics/attention.py:80-86).  native: see _decode_infer."""
        self.encoder._lens_host = ops.host_ints(inputs_seq_len)
        inputs = ops.to_device(inputs, torch.float32, self.device)
        isl = ops.to_device(inputs_seq_len, torch.int32, self.device)
        return self._decode_infer(inputs, isl, native=native).cpu().numpy()

    def decode(self, decoder_outputs_train, decoder_outputs_infer):
        """attention_seq2seq.py:666-701."""
        return decoder_outputs_train.predicted_ids, decoder_outputs_infer.predicted_ids

    def compute_ler(self, labels_true, labels_pred):
        """mean normalised edit distance between two lists of label sequences
        (tf.edit_distance(labels_pred, labels_true, normalize=True), :703-725)."""
        return _ler(labels_pred, labels_true)

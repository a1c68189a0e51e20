"""Attention layer types -- mirror of models/attention/decoders/attention_layer.py:12-16,128-131,
188-189,267-268,287-288 (the type table and its error behaviour).  The arithmetic runs in
csrc/attention.hip (asr_att_energy_*, asr_att_softmax_ctx_*), driven by AttentionSeq2Seq."""

ATTENTION_TYPE = [
    'bahdanau_content', 'normed_bahdanau_content',
    'location', 'hybrid', 'dot_product',
    'luong_dot', 'scaled_luong_dot', 'luong_general', 'luong_concat',
    'baidu_attetion']

ADDITIVE = ('bahdanau_content', 'location', 'hybrid', 'luong_concat')      # energy = sum_a v_a tanh(.)
DOT = ('dot_product', 'luong_dot', 'luong_general')        # energy = keys . query
HAS_KEYS_FC = ('bahdanau_content', 'location', 'hybrid', 'dot_product', 'luong_general')
USES_KEYS = ('bahdanau_content', 'hybrid', 'dot_product', 'luong_general', 'luong_concat')
HAS_QUERY_FC = ('bahdanau_content', 'location', 'hybrid', 'dot_product', 'luong_concat')
# luong_concat (:314-345): ONE bias-free FC W_concat over [h_enc; h_dec]; its first 2H rows act as the key
# projection and its last U rows as the query projection (tanh(W [a;b]) == tanh(W_a a + W_b b))
HAS_FILTER = ('location', 'hybrid')


def check_attention_type(attention_type):
    if attention_type not in ATTENTION_TYPE:
        raise ValueError(
            "attention type should be one of [%s], you provided %s." %
            (", ".join(ATTENTION_TYPE), attention_type))
    if attention_type in ('normed_bahdanau_content', 'scaled_luong_dot', 'baidu_attetion'):
        raise NotImplementedError      # as the reference (:188-189, :267-268, :287-288)


D_SCOPE = 'attention_decoder/decoder/'
AT_SCOPE = D_SCOPE + 'attention_layer/'


class AttentionLayer(object):
    """models/attention/decoders/attention_layer.py:19-113 AttentionLayer: one scoring + normalisation + context step,
    `(attention_weights, context_vector) = layer(encoder_outputs, decoder_output, encoder_outputs_length,
    attention_weights)`, on the HIP kernels (asr_att_energy_fwd / asr_att_loc_energy_fwd, asr_att_softmax_ctx_fwd).

    Same constructor arguments as the reference (`mode` is accepted and unused: there is no graph to reuse) plus
      store        the ParamStore that holds the layer's variables under their TF names
                   (attention_decoder/decoder/attention_layer/{W_query,W_keys,W_filter}/..., filter, v_a, W_concat);
                   AttentionSeq2Seq hands out a layer bound to its own store (model.attention_layer()), a layer built
                   without one declares its variables at the first call with the reference's initialisers;
      prev_alpha   'zeros' (default) reproduces what the reference's graph computes -- the `attention_weights`
                   argument of the location / hybrid types is the zeros tensor of AttentionDecoder.initialize() at
                   every step; 'carry' uses the argument as given.
    The training loop of AttentionSeq2Seq runs the same arithmetic through the fused native loop
    (asr_att_decoder_fwd); this class is the step-at-a-time form the inference path (AttentionDecoder.step) uses.
    encoder_outputs: [B, T, 2H] as in the reference, or time-major [T, B, 2H] with time_major_inputs=True (what the
    kernels read; the batch-major form is transposed once per distinct tensor)."""

    def __init__(self, attention_type, num_units, parameter_init, sharpening_factor, sigmoid_smoothing, mode=None,
                 name='attention_layer', store=None, prev_alpha='zeros', time_major_inputs=False, decoder_num_units=None,
                 seed=0):
        check_attention_type(attention_type)
        self.attention_type = attention_type
        self.num_units = num_units
        self.parameter_init = parameter_init
        self.sharpening_factor = sharpening_factor
        self.sigmoid_smoothing = sigmoid_smoothing
        self.reuse = mode not in (None, 'train')
        self.name = name
        self.store = store
        self.carry = prev_alpha == 'carry' and attention_type in HAS_FILTER
        self.time_major_inputs = time_major_inputs
        self.decoder_num_units = decoder_num_units
        self.seed = seed
        self._enc_key = None       # (tensor identity) -> cached time-major copy and keys

    # -- variables (attention_layer.py:128-131, 150-158, 200-221): declared at the first call of a standalone layer
    def _ensure_vars(self, E2, U, device):
        if self.store is not None:
            return
        import numpy as np
        from ....utils.parameter import ParamStore
        from ...ctc.ctc import truncated_normal
        rng = np.random.RandomState(self.seed)
        st = self.store = ParamStore(device)
        at, A, init = self.attention_type, self.num_units, self.parameter_init
        tn = lambda *s: truncated_normal(rng, init, s)
        key_dim = {'luong_dot': E2, 'luong_general': U}.get(at, A)
        if at == 'luong_concat':
            st.declare(AT_SCOPE + 'W_concat/weights', (E2 + U, A), tn(E2 + U, A))
        elif at in HAS_QUERY_FC:
            st.declare(AT_SCOPE + 'W_query/weights', (U, A), tn(U, A))
        if at in HAS_KEYS_FC:
            st.declare(AT_SCOPE + 'W_keys/weights', (E2, key_dim), tn(E2, key_dim))
            if at not in ('dot_product', 'luong_general'):
                st.declare(AT_SCOPE + 'W_keys/biases', (A,), np.zeros(A))
        if at in HAS_FILTER:
            taps = 201 if at == 'location' else 200
            st.declare(AT_SCOPE + 'filter', (taps, 1, 10), truncated_normal(rng, 0.1 if at == 'location' else init, (taps, 1, 10)))
            st.declare(AT_SCOPE + 'W_filter/weights', (10, A), tn(10, A))
            st.declare(AT_SCOPE + 'W_filter/biases', (A,), np.zeros(A))
        if at in ADDITIVE:
            lim = np.sqrt(3.0 / A)
            st.declare(AT_SCOPE + 'v_a', (A,), rng.uniform(-lim, lim, size=(A,)))
        st.finalize()

    def _w(self, which, E2):
        st = self.store
        if self.attention_type == 'luong_concat':
            w = st[AT_SCOPE + 'W_concat/weights']
            return w[:E2] if which == 'keys' else w[E2:]
        return st[AT_SCOPE + ('W_keys/weights' if which == 'keys' else 'W_query/weights')]

    def keys(self, enc_tm):
        """The projected encoder outputs the scoring kernels read ([T,B,A]; None for 'location', Q6)."""
        from .... import ops
        st, at = self.store, self.attention_type
        T, B, E2 = enc_tm.shape
        if at in USES_KEYS:
            b = st[AT_SCOPE + 'W_keys/biases'] if (AT_SCOPE + 'W_keys/biases') in st.views else None
            w = self._w('keys', E2)
            return ops.gemm(enc_tm.view(T * B, E2), w, bias=b).view(T, B, w.shape[1])
        if at == 'luong_dot':
            return enc_tm
        return None

    def query(self, decoder_output):
        from .... import ops
        st, at = self.store, self.attention_type
        if at in HAS_QUERY_FC:
            b = st[AT_SCOPE + 'W_filter/biases'] if at in HAS_FILTER else None
            return ops.gemm(decoder_output, self._w('query', self._E2), bias=b)
        return decoder_output

    def bind(self, enc_tm, keys=None):
        """Fix the encoder outputs (time-major [T,B,2H]) the following calls attend over; computes the keys once."""
        self._enc_tm, self._E2 = enc_tm, enc_tm.shape[2]
        self._keys = self.keys(enc_tm) if keys is None else keys
        self._enc_key = None
        return self

    def __call__(self, encoder_outputs, decoder_output, encoder_outputs_length, attention_weights, sigmoid_norm=None):
        """attention_layer.py:45-113.  Returns (attention_weights [B,T], context_vector [B,2H])."""
        import torch
        from .... import ops
        if self._enc_key is None or self._enc_key is not encoder_outputs:
            enc_tm = encoder_outputs if self.time_major_inputs else encoder_outputs.transpose(0, 1).contiguous()
            self._ensure_vars(enc_tm.shape[2], decoder_output.shape[1], enc_tm.device)
            self.bind(enc_tm)
            self._enc_key = encoder_outputs
        enc_tm, keys = self._enc_tm, self._keys
        T = enc_tm.shape[0]
        st = self.store
        v = st[AT_SCOPE + 'v_a'] if self.attention_type in ADDITIVE else None
        qz = self.query(decoder_output)
        if self.carry:
            energy = ops.att_loc_energy_fwd(attention_weights, st[AT_SCOPE + 'filter'], st[AT_SCOPE + 'W_filter/weights'],
                                            keys, qz, v, T)
        else:
            energy = ops.att_energy_fwd(keys, qz, v, T, 0 if self.attention_type in ADDITIVE else 1)
        seq = torch.as_tensor(encoder_outputs_length, dtype=torch.int32, device=enc_tm.device)
        if self.sigmoid_smoothing and sigmoid_norm is None:
            sigmoid_norm = torch.empty((enc_tm.shape[1],), dtype=torch.float32, device=enc_tm.device)
        return ops.att_softmax_ctx_fwd(energy, seq, self.sharpening_factor, enc_tm,
                                       sigmoid_norm=sigmoid_norm if self.sigmoid_smoothing else None)

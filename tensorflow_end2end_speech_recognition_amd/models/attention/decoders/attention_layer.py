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

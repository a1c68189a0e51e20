"""Attention decoder, step at a time -- mirror of models/attention/decoders/attention_decoder.py:17-308
(AttentionDecoderOutput, class AttentionDecoder: initialize / step / finalize / __call__ over dynamic_decode) and of
the tf.contrib.seq2seq helpers it is driven by (TrainingHelper: attention_seq2seq.py:444, GreedyEmbeddingHelper: :490).

Eager, on the HIP kernels: step() = decoder cell (one GEMM + asr_lstm_cell_fwd) -> AttentionLayer -> attentional vector
tanh(W [cell_out; ctx]) (no bias) -> output layer -> helper.sample -> next input [emb(next); ctx] (input feeding).
This is the path AttentionSeq2Seq's greedy inference runs (GreedyEmbeddingHelper, at most max_decode_length steps);
the teacher-forced training pass computes the same steps through the fused native loop (asr_att_decoder_fwd / _bwd),
which also keeps what the backward pass needs.  Variables live in the model's ParamStore under their TF names."""
import torch

from .... import ops
from .attention_layer import D_SCOPE
from .dynamic_decoder import dynamic_decode


class AttentionDecoderOutput(object):
    """namedtuple stand-in of attention_decoder.py:17-26 (logits, predicted_ids, decoder_output, attention_weights,
    context_vector); predicted_ids may be lazy (the INFER decoder only runs when its output is fetched)."""

    def __init__(self, logits=None, predicted_ids=None, decoder_output=None, attention_weights=None,
                 context_vector=None, lazy=None):
        self.logits, self._ids = logits, predicted_ids
        self.decoder_output, self.attention_weights, self.context_vector = \
            decoder_output, attention_weights, context_vector
        self._lazy = lazy

    @property
    def predicted_ids(self):
        if self._ids is None and self._lazy is not None:
            self._ids = self._lazy()
        return self._ids


class LSTMDecoderCell(object):
    """tf.contrib.rnn.LSTMBlockCell of the decoder (attention_seq2seq.py:353-363): (c, h) state, peepholes, cell clip,
    forget_bias 1; variables attention_decoder/decoder/lstm_cell/{kernel,bias,w_*_diag} of `store`."""

    def __init__(self, store, num_units, use_peephole=True, cell_clip=None, forget_bias=1.0, kernel=None):
        """kernel: the tensor to multiply with instead of the stored variable (a bf16-operand model hands in the kernel
        rounded to bf16, AttentionSeq2Seq._w_cell)."""
        self.store, self.num_units = store, num_units
        self.kernel = kernel
        self.output_size = num_units
        self.cell_clip, self.forget_bias = float(cell_clip or 0.0), forget_bias
        self.peep = None
        if use_peephole:
            self.peep = torch.stack([store[D_SCOPE + 'lstm_cell/w_i_diag'], store[D_SCOPE + 'lstm_cell/w_f_diag'],
                                     store[D_SCOPE + 'lstm_cell/w_o_diag']]).contiguous()

    def __call__(self, inputs, state, live=None):
        """inputs [B, E + 2H + U]: the reference concatenates [inputs, h] inside the cell; here the caller's buffer
        already carries h in its last U columns.  Returns (cell_output [B,U], (c, h)); rows with live == 0 keep their
        state (dynamic_decode's impute_finished copy-through)."""
        st = self.store
        c, h = state
        if live is None:
            live = torch.ones((inputs.shape[0],), dtype=torch.float32, device=inputs.device)
        W = self.kernel if self.kernel is not None else st[D_SCOPE + 'lstm_cell/kernel']
        pre = ops.gemm(inputs, W, bias=st[D_SCOPE + 'lstm_cell/bias'])
        _, _, c_new, h_new, h_raw = ops.lstm_cell_fwd(pre, c, h, self.peep, live, self.forget_bias, self.cell_clip)
        return h_raw, (c_new, h_new)


class TrainingHelper(object):
    """tf.contrib.seq2seq.TrainingHelper: feeds the embedded ground-truth token of every step (inputs [B,To,E] or
    time-major), finished where time + 1 >= sequence_length, zero inputs afterwards."""

    def __init__(self, inputs, sequence_length, time_major=False):
        self.inputs = inputs if time_major else inputs.transpose(0, 1)        # [To,B,E]
        self.sequence_length = torch.as_tensor(sequence_length, device=self.inputs.device).to(torch.int32)

    def initialize(self):
        return self.sequence_length <= 0, self.inputs[0].contiguous()

    def sample(self, time, outputs, state):
        return ops.argmax_rows(outputs)

    def next_inputs(self, time, outputs, state, sample_ids):
        nt = time + 1
        finished = nt >= self.sequence_length
        if nt < self.inputs.shape[0]:
            nxt = self.inputs[nt] * (~finished).float().unsqueeze(1)
        else:
            nxt = torch.zeros_like(self.inputs[0])
        return finished, nxt.contiguous(), state


class GreedyEmbeddingHelper(object):
    """tf.contrib.seq2seq.GreedyEmbeddingHelper: argmax of the logits, embedded, is the next input; finished at
    end_token; finished rows are fed the start token (attention_seq2seq.py:487-494)."""

    def __init__(self, embedding, start_tokens, end_token):
        self.embedding = embedding
        self.start_tokens = torch.as_tensor(start_tokens, dtype=torch.int32, device=embedding.device)
        self.end_token = int(end_token)

    def initialize(self):
        finished = torch.zeros_like(self.start_tokens, dtype=torch.bool)
        return finished, ops.embedding_gather(self.embedding, self.start_tokens)

    def sample(self, time, outputs, state):
        return ops.argmax_rows(outputs)

    def next_inputs(self, time, outputs, state, sample_ids):
        finished = sample_ids == self.end_token
        return finished, ops.embedding_gather(self.embedding, sample_ids), state


class AttentionDecoder(object):
    """attention_decoder.py:29-308.  Same constructor arguments; `store` holds attentional_vector / output_layer
    variables (attention_decoder/decoder/...).  encoder_outputs in the layout `attention_layer` was built for."""

    def __init__(self, rnn_cell, parameter_init, max_decode_length, num_classes, encoder_outputs,
                 encoder_outputs_seq_len, attention_layer, time_major, mode=None, name='attention_decoder', store=None):
        self.rnn_cell = rnn_cell
        self.parameter_init = parameter_init
        self.max_decode_length = max_decode_length
        self.num_classes = num_classes
        self.encoder_outputs = encoder_outputs
        self.encoder_outputs_seq_len = encoder_outputs_seq_len
        self.attention_layer = attention_layer
        self.time_major = time_major
        self.mode = mode
        self.name = name
        self.store = store if store is not None else rnn_cell.store
        self.initial_state = None
        self.helper = None

    @property
    def batch_size(self):
        return self.initial_state[0].shape[0]

    def __call__(self, initial_state, helper):
        """Returns (outputs: AttentionDecoderOutput of stacked per-step fields, time-major if self.time_major,
        final_state)."""
        self._setup(initial_state, helper)
        maximum_iterations = None if self.mode == 'train' else self.max_decode_length
        outputs, final_state = dynamic_decode(self, output_time_major=self.time_major, impute_finished=True,
                                              maximum_iterations=maximum_iterations)
        return self.finalize(outputs, final_state, None)

    def _setup(self, initial_state, helper):
        self.initial_state = initial_state
        self.helper = helper

    def initialize(self):
        """:142-168: first input = [helper's first input ; zero context], attention weights start at zero."""
        finished, first_inputs = self.helper.initialize()
        enc = self.encoder_outputs
        tm = self.attention_layer.time_major_inputs
        B, T, E2 = (enc.shape[1], enc.shape[0], enc.shape[2]) if tm else enc.shape
        U = self.rnn_cell.num_units
        self._E, self._E2, self._U = first_inputs.shape[1], E2, U
        self._cell_in = torch.empty((B, self._E + E2 + U), dtype=torch.float32, device=enc.device)
        self._av_in = torch.empty((B, U + E2), dtype=torch.float32, device=enc.device)
        self.attention_weights = torch.zeros((B, T), dtype=torch.float32, device=enc.device)
        self._snorm = torch.empty((B,), dtype=torch.float32, device=enc.device) \
            if self.attention_layer.sigmoid_smoothing else None
        ctx = torch.zeros((B, E2), dtype=torch.float32, device=enc.device)
        return finished, (first_inputs, ctx), self.initial_state

    def _compute_output(self, decoder_output, attention_weights):
        """:170-211."""
        st = self.store
        attention_weights, context_vector = self.attention_layer(
            encoder_outputs=self.encoder_outputs, decoder_output=decoder_output,
            encoder_outputs_length=self.encoder_outputs_seq_len, attention_weights=attention_weights,
            sigmoid_norm=self._snorm)
        U = self._U
        self._av_in[:, :U].copy_(decoder_output)
        self._av_in[:, U:].copy_(context_vector)
        attentional_vector = ops.tanh_fwd(ops.gemm(self._av_in, st[D_SCOPE + 'attentional_vector/weights']))
        logits = ops.gemm(attentional_vector, st[D_SCOPE + 'output_layer/weights'],
                          bias=st[D_SCOPE + 'output_layer/biases'])
        return attentional_vector, logits, attention_weights, context_vector

    def step(self, time, inputs, state, live=None):
        """:256-295.  inputs = (helper input [B,E], previous context [B,2H]) -- the input-feeding concatenation of
        _att_next_inputs (:222-247) is formed in the cell's input buffer together with the fed-back h."""
        emb, ctx = inputs
        E, E2 = self._E, self._E2
        self._cell_in[:, :E].copy_(emb)
        self._cell_in[:, E:E + E2].copy_(ctx)
        self._cell_in[:, E + E2:].copy_(state[1])
        cell_output, cell_state = self.rnn_cell(self._cell_in, state, live=live)
        attentional_vector, logits, attention_weights, context_vector = self._compute_output(
            decoder_output=cell_output, attention_weights=self.attention_weights)
        self.attention_weights = attention_weights
        sample_ids = self.helper.sample(time=time, outputs=logits, state=cell_state)
        outputs = AttentionDecoderOutput(logits=logits, predicted_ids=sample_ids, decoder_output=attentional_vector,
                                         attention_weights=attention_weights, context_vector=context_vector)
        finished, next_inputs, next_state = self.helper.next_inputs(time=time, outputs=outputs, state=cell_state,
                                                                    sample_ids=sample_ids)
        return outputs, next_state, (next_inputs, context_vector), finished

    def finalize(self, outputs, final_state, final_seq_len):
        return outputs, final_state

"""Joint CTC-attention model -- mirror of models/attention/joint_ctc_attention.py:15-346
(class JointCTCAttention): an extra 'ctc_output' FC on the encoder outputs and
loss = (1 - lambda_weight) * sequence_loss + lambda_weight * mean(ctc_loss)   (:296-318).

ctc num_classes = num_classes + 1 (blank = last).  ignore_longer_outputs_than_inputs=False in the
reference (:315): an utterance without a valid alignment raises ValueError here too.
Quirk Q2 (the [B*T,C] -> [T,B,C] reshape of batch-major logits, :190-226) is NOT reproduced: the
encoder outputs are kept time-major, which is the intended computation.
Quirk Q15 (found by EXECUTING the reference's class, tests/golden/make_golden_tfshim.py): its constructor hands
clip_activation_decoder=50, weight_decay=0.0, time_major=True, sharpening_factor=1.0, logits_temperature=1.0 to the base
class as literals (:133-137) -- whatever the recipe passed (examples/timit/training/train_joint_ctc_attention.py:382-386
passes all of them from the config) is ignored.  REPRODUCED by default, with a warning when a passed value differs from
the literal; `honour_ctor_args=True` (extension keyword) uses the caller's values instead.
"""
import warnings

import numpy as np
import torch

from ... import ops
from ..._lib import ASR_BF16, ASR_F32
from ...utils.io.labels.sparsetensor import dense_to_flat, sparse_to_flat
from ..ctc.ctc import CTC, truncated_normal
from .attention_seq2seq import AttentionSeq2Seq


def _not_enough_time(n):
    return ValueError('Not enough time for target transition sequence (%d utterance(s))' % n)


class JointCTCAttention(AttentionSeq2Seq):

    def __init__(self, input_size, encoder_type, encoder_num_units, encoder_num_layers, encoder_num_proj,
                 attention_type, attention_dim, decoder_type, decoder_num_units, decoder_num_layers,
                 embedding_dim, lambda_weight, num_classes, sos_index, eos_index, max_decode_length,
                 lstm_impl='LSTMBlockCell', use_peephole=True, splice=1, parameter_init=0.1,
                 clip_grad_norm=5.0, clip_activation_encoder=50, clip_activation_decoder=50,
                 weight_decay=0.0, time_major=True, sharpening_factor=1.0, logits_temperature=1.0,
                 name='joint_ctc_attention', honour_ctor_args=False, **kw):
        assert 0 <= lambda_weight <= 1, 'lambda_weight must be in [0, 1]'
        if not honour_ctor_args:                      # Q15: joint_ctc_attention.py:133-137
            passed = dict(clip_activation_decoder=clip_activation_decoder, weight_decay=weight_decay,
                          time_major=time_major, sharpening_factor=sharpening_factor,
                          logits_temperature=logits_temperature)
            fixed = dict(clip_activation_decoder=50, weight_decay=0.0, time_major=True, sharpening_factor=1.0,
                         logits_temperature=1.0)
            dropped = {k: v for k, v in passed.items() if v != fixed[k]}
            if dropped:
                warnings.warn('JointCTCAttention ignores %s as the reference does (joint_ctc_attention.py:133-137 '
                              'passes literals to its base class); honour_ctor_args=True uses them'
                              % ', '.join('%s=%r' % kv for kv in sorted(dropped.items())))
            clip_activation_decoder, weight_decay, time_major = 50, 0.0, True
            sharpening_factor, logits_temperature = 1.0, 1.0
        self.lambda_weight = float(lambda_weight)
        self.ctc_num_classes = num_classes + 1
        init = parameter_init

        def extra(rng, enc_dim):
            return [('ctc_output/weights', (enc_dim, num_classes + 1), truncated_normal(rng, init, (enc_dim, num_classes + 1))),
                    ('ctc_output/biases', (num_classes + 1,), np.zeros(num_classes + 1))]
        super(JointCTCAttention, self).__init__(
            input_size=input_size, encoder_type=encoder_type, encoder_num_units=encoder_num_units,
            encoder_num_layers=encoder_num_layers, encoder_num_proj=encoder_num_proj,
            attention_type=attention_type, attention_dim=attention_dim, decoder_type=decoder_type,
            decoder_num_units=decoder_num_units, decoder_num_layers=decoder_num_layers,
            embedding_dim=embedding_dim, num_classes=num_classes, sos_index=sos_index, eos_index=eos_index,
            max_decode_length=max_decode_length, lstm_impl=lstm_impl, use_peephole=use_peephole, splice=splice,
            parameter_init=parameter_init, clip_grad_norm=clip_grad_norm,
            clip_activation_encoder=clip_activation_encoder, clip_activation_decoder=clip_activation_decoder,
            weight_decay=weight_decay, time_major=time_major, sharpening_factor=sharpening_factor,
            logits_temperature=logits_temperature, name=name, _extra_vars=extra, **kw)

    def compute_loss(self, inputs, labels, ctc_labels, inputs_seq_len, labels_seq_len, keep_prob_encoder,
                     keep_prob_decoder, keep_prob_embedding, scope=None, is_training=True):
        """:237-346.  Returns (total_loss, logits, ctc_logits [T,B,C+1], dec_out_train, dec_out_infer)."""
        return super(JointCTCAttention, self).compute_loss(
            inputs, labels, inputs_seq_len, labels_seq_len, keep_prob_encoder, keep_prob_decoder,
            keep_prob_embedding, scope=scope, is_training=is_training, lambda_weight=self.lambda_weight,
            ctc_labels=ctc_labels)

    def _ctc_head_launch(self, enc, seq_p, ctc_labels, B, lam, is_training):
        """Issue the CTC head (on whatever stream is current -- AttentionSeq2Seq.compute_loss puts it on the side lane);
        nothing here waits for the device."""
        st, dev = self.store, self.device
        T, Bp, E2 = enc.shape
        Cc = self.ctc_num_classes
        # bf16-operand models: the head multiplies the encoder's operand copy by the operand copy of its weights, as the
        # CTC model's head does (ctc.py) -- with a few thousand classes (cfg E: 3 387) the fp32 form of these three
        # [T*B x 2H x C] products is 17 ms of an 83 ms step
        if self.dtype == ASR_BF16:
            x_op, w_op = self.encoder._out_op.contiguous().view(T * Bp, E2), st.shadow(self.dtype)['ctc_output/weights']
        else:
            x_op, w_op = enc.view(T * Bp, E2), st['ctc_output/weights']
        self._ctc_x_op = x_op
        logits = torch.empty((T, Bp, Cc), dtype=torch.float32, device=dev)
        ops.gemm(x_op, w_op, bias=st['ctc_output/biases'], out=logits.view(T * Bp, Cc))
        flat, offsets, max_len = CTC._labels_to_flat(ctc_labels, B)
        if Bp > B:
            offsets = np.concatenate([offsets, np.full(Bp - B, offsets[-1], dtype=np.int32)])
        # pinned staging + asynchronous copies: a pageable upload would drain the stream (ops.to_device)
        flat_d = ops.to_device(flat if len(flat) else np.zeros(1, np.int32), torch.int32, dev)
        off_d = ops.to_device(offsets, torch.int32, dev)
        losses, grad, ninf = ops.ctc_loss(logits, flat_d, off_d, seq_p, max_len, grad_scale=lam / B,
                                          want_grad=is_training)
        return logits, losses, grad, ninf, flat_d, off_d

    def _ctc_head_finish(self, pending, B):
        logits, losses, grad, ninf = pending[:4]
        # ignore_longer_outputs_than_inputs=False: checked through the deferred counter watch while training (the host
        # must not wait for the forward pass every step; the error surfaces at most ops.DeferredCheck.DEPTH (4) arm() calls late, in steady state <= 3 steps), at once otherwise
        ops.defer_zero_check(ninf, _not_enough_time, blocking=grad is None)
        self.ctc_losses = losses[:B]
        return logits[:, :B], losses[:B].mean(), dict(dlogits=grad)

    def _ctc_head_backward(self, tape, enc, denc, accumulate=True):
        """d_enc (+)= dlogits . W^T on the current stream; the head's own weight gradients on side lane 1 (joined by
        encoder.backward)."""
        st = self.store
        T, Bp, E2 = enc.shape
        dl = tape['dlogits'].view(T * Bp, -1)
        x_op = self._ctc_x_op
        if self.dtype == ASR_BF16:
            dl_op, w_op = ops.cast_from_f32(dl, ASR_BF16), st.shadow(self.dtype)['ctc_output/weights']
        else:
            dl_op, w_op = dl, st['ctc_output/weights']
        ops.gemm(dl_op, w_op, transB=True, out=denc.view(T * Bp, E2), accumulate=accumulate)
        with ops.side_lane(enc.device, keep=(x_op, dl_op, dl), lane=1):
            ops.gemm(x_op, dl_op, transA=True, out=st.g('ctc_output/weights'))
            ops.colsum(dl, out=st.g('ctc_output/biases'))

"""Multi-task CTC model -- mirror of models/ctc/multitask_ctc.py:14-416 (class MultitaskCTC) on the HIP path.

One encoder, two CTC heads: the main head ('output_main', optionally behind the 'bottleneck' FC) on the top layer
and the sub head ('output_sub') on the outputs of layer `num_layers_sub` (:124-206); the loss is
main_task_weight * mean CTC(main) + (1 - main_task_weight) * mean CTC(sub) (+ weight decay) (:250-276).  Every
kernel is the one the single-task model uses; the new data flow is the second head and the join of its gradient
inside the encoder stack.

Differences from the reference, flagged: `ignore_longer_outputs_than_inputs=False` (:256,269) makes TF raise on
an utterance whose label does not fit its frames; here such utterances contribute loss 0 / gradient 0 as in the
single-task model and are counted in `num_infeasible` / `num_infeasible_sub`.  The reference's `_build` calls the
encoder without `is_training` (:122) and cannot run as written; `is_training` defaults to True here."""
import numpy as np
import torch

from ... import ops
from ..._lib import ASR_BF16, ASR_F32
from ..encoders.load_encoder import load
from .ctc import CTC, Placeholder, truncated_normal


class MultitaskCTC(CTC):
    head_scope = 'output_main'

    def __init__(self, encoder_type, input_size, num_units, num_layers_main, num_layers_sub, num_classes_main,
                 num_classes_sub, main_task_weight, lstm_impl='LSTMBlockCell', use_peephole=True, splice=1,
                 parameter_init=0.1, clip_grad_norm=None, clip_activation=None, num_proj=None, weight_decay=0.0,
                 bottleneck_dim=None, time_major=True, **kw):
        if float(main_task_weight) < 0 or float(main_task_weight) > 1:
            raise ValueError('Set main_task_weight between 0 to 1.')
        if encoder_type not in ('multitask_blstm', 'multitask_lstm'):
            load(encoder_type)                   # ValueError for unknown keys
            raise NotImplementedError            # multitask_ctc.py:97-98
        self.num_layers_sub = num_layers_sub
        self.num_classes_sub = num_classes_sub + 1           # + blank (:81)
        self.main_task_weight = float(main_task_weight)
        self.sub_task_weight = 1.0 - self.main_task_weight
        self.labels_sub_pl_list = []
        super(MultitaskCTC, self).__init__(
            encoder_type, input_size, num_units, num_layers_main, num_classes_main, lstm_impl=lstm_impl,
            use_peephole=use_peephole, splice=splice, parameter_init=parameter_init, clip_grad_norm=clip_grad_norm,
            clip_activation=clip_activation, num_proj=num_proj, weight_decay=weight_decay,
            bottleneck_dim=bottleneck_dim, time_major=time_major, **kw)
        self.name = encoder_type + '_ctc'

    def _create_encoder(self, encoder_type, input_size, splice, num_stack, num_units, num_layers, lstm_impl,
                        use_peephole, parameter_init, clip_activation):
        return load(encoder_type)(                                          # multitask_ctc.py:87-96
            num_units=num_units, num_proj=self.num_proj, num_layers_main=num_layers,
            num_layers_sub=self.num_layers_sub, lstm_impl=lstm_impl, use_peephole=use_peephole,
            parameter_init=parameter_init, clip_activation=clip_activation, time_major=True,
            dtype=self._requested_dtype)

    def _declare_heads(self, rng, enc_dim, parameter_init):
        # creation order of the reference graph: output_sub (:150-158), bottleneck (:173-181), output_main (:186-194)
        self.store.declare('output_sub/weights', (enc_dim, self.num_classes_sub),
                           truncated_normal(rng, parameter_init, (enc_dim, self.num_classes_sub)))
        self.store.declare('output_sub/biases', (self.num_classes_sub,), np.zeros(self.num_classes_sub))
        super(MultitaskCTC, self)._declare_heads(rng, enc_dim, parameter_init)

    # ------------------------------------------------------------------ graph pieces
    def _build(self, inputs, inputs_seq_len, keep_prob, is_training=True):
        """:100-206 -> (logits_main [T,Bp,C_main], logits_sub [T,Bp,C_sub]), time-major."""
        logits_main = super(MultitaskCTC, self)._build(inputs, inputs_seq_len, keep_prob, is_training)
        sub_op = self.encoder._out_sub_op                     # [T,Bp,E] in the MFMA operand dtype
        T, Bp, E = sub_op.shape
        sh = self.store.shadow(self.dtype)
        logits_sub = torch.empty((T, Bp, self.num_classes_sub), dtype=torch.float32, device=sub_op.device)
        ops.gemm(sub_op.view(T * Bp, E), sh['output_sub/weights'], bias=self.store['output_sub/biases'],
                 out=logits_sub.view(T * Bp, self.num_classes_sub))
        self._sub_head_in = sub_op
        return logits_main, logits_sub

    def create_placeholders(self):
        """:208-225."""
        super(MultitaskCTC, self).create_placeholders()
        self.labels_sub_pl_list.append(Placeholder('labels_sub'))

    def _upload_labels(self, labels, B, Bp):
        flat, offsets, max_len = self._labels_to_flat(labels, B)
        if Bp > B:
            offsets = np.concatenate([offsets, np.full(Bp - B, offsets[-1], dtype=np.int32)])
        flat_d, off_d = ops.upload_ints(self.device, [flat if len(flat) else np.zeros(1, np.int32), offsets])
        return flat_d, off_d, max_len

    def compute_loss(self, inputs, labels_main, labels_sub, inputs_seq_len, keep_prob, scope=None,
                     is_training=True):
        """:227-312.  Returns (total_loss, logits_main [T,B,C_main], logits_sub [T,B,C_sub])."""
        dev = self.device
        self.encoder._lens_host = ops.host_ints(inputs_seq_len)
        inputs = ops.to_device(inputs, torch.float32, dev)
        inputs_seq_len = ops.to_device(inputs_seq_len, torch.int32, dev)
        B = inputs.shape[0]
        logits_main, logits_sub = self._build(inputs, inputs_seq_len, keep_prob, is_training)
        Bp = logits_main.shape[1]
        seq_p = self.encoder.seq_len_padded
        fm, om, lm = self._upload_labels(labels_main, B, Bp)
        fs, os_, ls = self._upload_labels(labels_sub, B, Bp)
        # the task weights and the 1/B of the batch mean are folded into the CTC gradient (:258-276)
        losses_m, grad_m, ninf_m = ops.ctc_loss(logits_main, fm, om, seq_p, lm,
                                                grad_scale=self.main_task_weight / B, want_grad=is_training)
        losses_s, grad_s, ninf_s = ops.ctc_loss(logits_sub, fs, os_, seq_p, ls,
                                                grad_scale=self.sub_task_weight / B, want_grad=is_training)
        self.ctc_loss_main = losses_m[:B].mean()
        self.ctc_loss_sub = losses_s[:B].mean()
        total_loss = self.ctc_loss_main * self.main_task_weight + self.ctc_loss_sub * self.sub_task_weight
        if self.weight_decay > 0:
            l2 = torch.zeros((), dtype=torch.float32, device=dev)
            ops.weight_decay(None, self.store.flat, self.store.plan, self.store.decay_mask,
                             self.weight_decay, l2_out=l2)
            total_loss = total_loss + l2                                     # :240-247
        self.ctc_losses, self.ctc_losses_sub = losses_m[:B], losses_s[:B]
        self.num_infeasible, self.num_infeasible_sub = ninf_m, ninf_s
        from .ctc import _not_enough_time
        ops.defer_zero_check(ninf_m, _not_enough_time, blocking=not is_training)
        ops.defer_zero_check(ninf_s, _not_enough_time, blocking=not is_training)
        self._tape = dict(dlogits=grad_m, dlogits_sub=grad_s, B=B) if is_training else None
        total_loss._asr_model = self
        return total_loss, logits_main[:, :B], logits_sub[:, :B]

    # ------------------------------------------------------------------ backward
    def _encoder_backward_extra(self):
        """Sub head: its weight / bias gradients (side stream) and the gradient entering the encoder at the
        sub-task layer."""
        st = self.store
        dl = self._tape['dlogits_sub']
        T, Bp, Cs = dl.shape
        x_op = self._sub_head_in
        E = x_op.shape[2]
        sh = st.shadow(self.dtype)
        dl2d = dl.view(T * Bp, Cs)
        dl_op = ops.cast_from_f32(dl2d, ASR_BF16) if self.dtype == ASR_BF16 else dl2d
        dsub = ops.gemm(dl_op, sh['output_sub/weights'], transB=True, out_dtype=ASR_F32)
        with ops.side_lane(dl.device, keep=(x_op, dl_op, dl2d)):            # joined by encoder.backward
            ops.gemm(x_op.view(T * Bp, E), dl_op, transA=True, out=st.g('output_sub/weights'))
            ops.colsum(dl2d, out=st.g('output_sub/biases'))
        return dict(d_outputs_sub=dsub.view(T, Bp, E))

    # ------------------------------------------------------------------ decode / eval
    def decoder(self, logits_main, logits_sub, inputs_seq_len, beam_width=1, merge_repeated=True):
        """:314-347 -> (decode_op_main, decode_op_sub), each the SparseTensor triple (merge_repeated: see CTC.decoder)."""
        dec = super(MultitaskCTC, self).decoder
        return (dec(logits_main, inputs_seq_len, beam_width, merge_repeated),
                dec(logits_sub, inputs_seq_len, beam_width, merge_repeated))

    def posteriors(self, logits_main, logits_sub):
        """:349-372: softmax over classes on the batch-major flattenings."""
        lm = logits_main.transpose(0, 1).contiguous()
        ls = logits_sub.transpose(0, 1).contiguous()
        return ops.softmax_rows(lm.view(-1, self.num_classes)), ops.softmax_rows(ls.view(-1, self.num_classes_sub))

    def compute_ler(self, decode_op_main, decode_op_sub, labels_main, labels_sub):
        """:374-416 -> (ler_main, ler_sub)."""
        ler = super(MultitaskCTC, self).compute_ler
        return ler(decode_op_main, labels_main), ler(decode_op_sub, labels_sub)

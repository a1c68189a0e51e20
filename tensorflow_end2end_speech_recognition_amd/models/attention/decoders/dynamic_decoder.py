"""dynamic_decode -- mirror of models/attention/decoders/dynamic_decoder.py:68-218 (the reference's copy of
tf.contrib.seq2seq.dynamic_decode without final sequence lengths): run decoder.step until every row has finished or
`maximum_iterations` steps have been taken; with impute_finished the emitted fields of a finished row are zero and
its state is copied through (:172-190).  Eager loop; the per-step tensors are stacked at the end."""
import torch


def dynamic_decode(decoder, output_time_major=False, impute_finished=False, maximum_iterations=None, scope=None):
    from .attention_decoder import AttentionDecoderOutput
    if maximum_iterations is not None and maximum_iterations < 0:
        raise ValueError('maximum_iterations must be a non-negative scalar')
    finished, inputs, state = decoder.initialize()
    finished = finished.clone()
    B = finished.shape[0]
    live_rows = getattr(decoder, 'live_rows', None)          # rows of a padded batch that never decode
    if live_rows is not None:
        finished = finished | ~live_rows
    steps = []
    time = 0
    while True:
        if maximum_iterations is not None and time >= maximum_iterations:
            break
        if bool(finished.all()):
            break
        live = (~finished).float()
        outputs, next_state, next_inputs, step_finished = decoder.step(time, inputs, state, live=live)
        if impute_finished:      # zero the emitted fields of rows that had finished BEFORE this step
            lv = live.unsqueeze(1)
            outputs = AttentionDecoderOutput(
                logits=outputs.logits * lv, predicted_ids=outputs.predicted_ids * live.to(outputs.predicted_ids.dtype),
                decoder_output=outputs.decoder_output * lv, attention_weights=outputs.attention_weights * lv,
                context_vector=outputs.context_vector * lv)
            next_inputs = (next_inputs[0], next_inputs[1] * lv) if isinstance(next_inputs, tuple) else next_inputs
        steps.append(outputs)
        finished = finished | step_finished
        inputs, state = next_inputs, next_state
        time += 1
    if not steps:
        return AttentionDecoderOutput(), state

    def stack(field):
        t = torch.stack([getattr(s, field) for s in steps], 0)          # [T_out, B, ...]
        return t if output_time_major else t.transpose(0, 1)
    out = AttentionDecoderOutput(logits=stack('logits'), predicted_ids=stack('predicted_ids'),
                                 decoder_output=stack('decoder_output'), attention_weights=stack('attention_weights'),
                                 context_vector=stack('context_vector'))
    return out, state

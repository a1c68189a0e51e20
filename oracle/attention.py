"""Oracle (test infrastructure; PINNED to the reference's own code since round 5, see oracle/__init__.py): CPU
restatement of the attention encoder-decoder and the joint CTC-attention model.

Pin: tests/golden/tfshim_v1.npz holds what the UNCHANGED reference files listed below computed when executed on an eager
float64 TensorFlow stand-in (tests/golden/make_golden_tfshim.py); tests/test_oracle_tfshim.py holds attention_step (all
seven types, sharpening, sigmoid smoothing, ragged masks, carried weights) and attention_model_forward / _infer (logits,
ids, attention weights, loss, every gradient; joint loss) to those numbers at 1e-11 / 1e-9.

Follows
  * models/attention/attention_seq2seq.py:193-277 (_build), :413-509 (_decode_train /
    _decode_infer), :579-664 (compute_loss: logits/temperature, +1e-10, sequence_loss masked
    by labels_seq_len-1 over labels[:,1:])
  * models/attention/decoders/attention_layer.py:45-113 (mask with float32.min, sharpening,
    softmax, context) and :115-347 (energies)
  * models/attention/decoders/attention_decoder.py:142-295 (initialize: first input
    [emb(<SOS>); zeros], step: cell -> attention -> tanh(FC_nobias([cell_out; ctx])) ->
    output FC -> next input [emb(next); ctx]), dynamic_decoder.py:148-197 (impute_finished)
  * models/attention/bridge.py:128-151 (flatten final (c,h) fw+bw -> FC -> split (c0,h0))
  * models/attention/joint_ctc_attention.py:182-346 ((1-lambda)*xent + lambda*mean ctc on a
    'ctc_output' FC over the encoder outputs; the [B*T,C] -> [T,B,C] reshape bug Q2 is NOT
    reproduced: the intended transpose is used)
Reference quirk Q1 is reproduced by default because it is what the reference graph
computes: the "previous attention weights" fed to location / hybrid attention are always the
zeros tensor of initialize(), so the location features reduce to the W_filter bias.  prev_alpha='carry'
selects the recurrence the code was written to express (attention_layer.py:191-265): the previous step's
weights through conv1d([201|200, 1, 10], SAME) -> W_filter.
"""
import numpy as np
import torch

from . import lstm as olstm
from .model import ctc_loss as _ctc_loss

F32_MIN = float(np.finfo(np.float32).min)
ADDITIVE = ('bahdanau_content', 'location', 'hybrid')
DOT = ('dot_product', 'luong_dot', 'luong_general')


def location_features(p, prev_alpha):
    """attention_layer.py:200-221 / :239-257: f = tf.nn.conv1d(alpha_{i-1} [B,T,1], filter [taps,1,10], stride 1,
    'SAME') -> fully_connected(10 -> A, bias).  SAME with an even kernel (hybrid: 200 taps) pads
    (taps-1)//2 = 99 frames before and 100 after; 201 taps (location): 100 / 100.
    prev_alpha [B,T] -> [B,T,A]."""
    filt = p['filter']                                                    # [taps,1,10]
    taps = filt.shape[0]
    before = (taps - 1) // 2
    a = torch.nn.functional.pad(prev_alpha.unsqueeze(1), (before, taps - 1 - before))      # [B,1,T+taps-1]
    f = torch.nn.functional.conv1d(a, filt.permute(2, 1, 0))             # cross-correlation, like tf.nn.conv1d
    return f.transpose(1, 2) @ p['W_filter/weights'] + p['W_filter/biases']


class _MMBwdRound(torch.autograd.Function):
    """y = a @ w, exact in the forward; in the BACKWARD the products take operands rounded by `rnd` the way the device's
    bf16-operand model rounds them outside the decoder loop (models/attention/attention_seq2seq.py, ASR_ATT_BWD_BF16):
    da = R(g) @ R(w)^T (round_g_da) or g @ R(w)^T, dw = R(a)^T @ R(g) (round_a_dw) or a^T @ R(g) (the key projection: its
    input is the encoder's operand-dtype output as it lies in memory, bf16-valued already on the device)."""

    @staticmethod
    def forward(ctx, a, w, rnd, round_g_da, round_a_dw=True):
        ctx.save_for_backward(a, w)
        ctx.rnd, ctx.round_g_da, ctx.round_a_dw = rnd, round_g_da, round_a_dw
        return a @ w

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        R = ctx.rnd
        gr = R(g)
        da = (gr if ctx.round_g_da else g) @ R(w).transpose(-1, -2)
        dw = (R(a) if ctx.round_a_dw else a).reshape(-1, a.shape[-1]).t() @ gr.reshape(-1, g.shape[-1])
        return da, dw, None, None, None


class _CtxBwdRound(torch.autograd.Function):
    """ctx[b] = sum_t alpha[b, t] enc[b, t]; backward: d alpha = enc . g (inside the decoder loop: unrounded), d enc =
    R(alpha) (x) R(g) (the per-utterance alpha^T . dctx products after the loop, on bf16 operands)."""

    @staticmethod
    def forward(ctx, alpha, enc, rnd):
        ctx.save_for_backward(alpha, enc)
        ctx.rnd = rnd
        return (alpha.unsqueeze(2) * enc).sum(1)

    @staticmethod
    def backward(ctx, g):
        alpha, enc = ctx.saved_tensors
        R = ctx.rnd
        return (enc * g.unsqueeze(1)).sum(2), R(alpha).unsqueeze(2) * R(g).unsqueeze(1), None


def attention_step(p, att_type, enc_bt, keys, s, seq_len, sharpening=1.0, sigmoid_smoothing=False,
                   prev_alpha=None, bwd_round=None):
    """enc_bt [B,T,2H]; keys [B,T,A] or None; s [B,U] -> (alpha [B,T], ctx [B,2H]).
    sigmoid_smoothing: attention_layer.py:92-96, sigmoid(e) / sum_t sigmoid(e) instead of the softmax.
    prev_alpha: None = the reference's EFFECTIVE graph (quirk Q1: the location features see the zeros of
    initialize() at every step, i.e. reduce to the W_filter bias); a [B,T] tensor = the INTENDED recurrence,
    the previous step's attention weights go through conv1d -> W_filter (model switch prev_alpha='carry')."""
    B, T, _ = enc_bt.shape
    if att_type in ADDITIVE:
        z = (s @ p['W_query/weights']).unsqueeze(1)                       # [B,1,A]
        if att_type in ('bahdanau_content', 'hybrid'):
            z = z + keys
        if att_type in ('location', 'hybrid'):
            if prev_alpha is None:
                z = z + p['W_filter/biases']      # conv(zeros) @ W_filter + b  (quirk Q1)
            else:
                z = z + location_features(p, prev_alpha)
        if att_type == 'location' and z.shape[1] == 1:
            z = z.expand(B, T, z.shape[2])
        energy = (p['v_a'] * torch.tanh(z)).sum(2)
    elif att_type == 'dot_product':
        energy = (keys @ (s @ p['W_query/weights']).unsqueeze(2)).squeeze(2)
    elif att_type == 'luong_dot':
        energy = (enc_bt @ s.unsqueeze(2)).squeeze(2)
    elif att_type == 'luong_general':
        energy = (keys @ s.unsqueeze(2)).squeeze(2)
    elif att_type == 'luong_concat':      # attention_layer.py:314-345, as written there: FC over the concatenation
        cat = torch.cat([enc_bt, s.unsqueeze(1).expand(B, T, s.shape[1])], 2)
        energy = (p['v_a'] * torch.tanh(cat @ p['W_concat/weights'])).sum(2)
    else:
        raise NotImplementedError(att_type)
    mask = (torch.arange(T).unsqueeze(0) < seq_len.unsqueeze(1)).to(enc_bt.dtype)
    energy = energy * mask + (1.0 - mask) * F32_MIN
    energy = energy * sharpening
    if sigmoid_smoothing:
        sg = torch.sigmoid(energy)
        alpha = sg / sg.sum(dim=1, keepdim=True)
    else:
        alpha = torch.softmax(energy, dim=1)
    ctx = (alpha.unsqueeze(2) * enc_bt).sum(1) if bwd_round is None else _CtxBwdRound.apply(alpha, enc_bt, bwd_round)
    return alpha, ctx


def compute_keys(p, att_type, enc_bt, bwd_round=None):
    mm = (lambda a, w: a @ w) if bwd_round is None else (lambda a, w: _MMBwdRound.apply(a, w, bwd_round, True, False))
    if att_type in ('bahdanau_content', 'hybrid'):
        return mm(enc_bt, p['W_keys/weights']) + p['W_keys/biases']
    if att_type in ('dot_product', 'luong_general'):
        return mm(enc_bt, p['W_keys/weights'])
    return None


def decoder_params(sd, dtype=torch.float64, requires_grad=True):
    pre = 'attention_decoder/decoder/'
    out = {}
    for k, v in sd.items():
        if k.startswith(pre) or k.startswith('output_embedding/') or k.startswith('bridge/') \
                or k.startswith('ctc_output/'):
            t = torch.as_tensor(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v), dtype=dtype).clone()
            t.requires_grad_(requires_grad)
            out[k] = t
    return out


def attention_model_forward(sd, inputs_btd, labels, inputs_seq_len, labels_seq_len, enc_layers,
                            att_type, clip_enc=0.0, clip_dec=0.0, sharpening=1.0, temperature=1.0,
                            drop_emb=None, drop_dec=None, ctc_labels=None, lambda_weight=None,
                            dtype=torch.float64, sigmoid_smoothing=False, prev_alpha='zeros', operand_round=None,
                            bwd_round='same'):
    """Teacher-forced forward + loss + all parameter gradients.
    bwd_round: the rounding of the BACKWARD pass's batched products outside the decoder loop (round 6; the device's
    ASR_ATT_BWD_BF16 default of a bf16-operand model): the attentional vector's weight / input gradients, alpha^T . dctx
    into d enc, the key projection's two gradients, the decoder cell's weight gradient -- 'same' = operand_round (none for
    the plain oracle), None = exact products, or a rounding function.
    prev_alpha: 'zeros' (reference's effective graph, Q1) | 'carry' (previous step's weights feed location / hybrid).
    labels [B, Lmax] int (<SOS> y <EOS>, padded with eos); returns dict(loss, logits [B,To,C],
    alphas, grads, ...).
    operand_round (e.g. oracle.lstm.bf16_round_t): the rounding points of the bf16-operand device model in the forward
    -- the inputs, the encoder's LSTM kernels and every h it emits / feeds back, and the CTC head's weight matrix
    (straight-through), and since round 4 the decoder cell's kernel (its per-step products stream bf16 weights against
    fp32 activations); the attention layer, the bridge and the decoder's other products multiply in fp32 on the device
    and stay in `dtype` here."""
    from .model import params_from_state_dict
    if bwd_round == 'same':
        bwd_round = operand_round
    if operand_round is not None:
        sd = dict(sd)
        for k in list(sd):
            if ((k.startswith('encoder/') and k.endswith('/kernel')) or k == 'ctc_output/weights' or
                    k == 'attention_decoder/decoder/lstm_cell/kernel'):
                v = sd[k].detach().cpu() if torch.is_tensor(sd[k]) else torch.as_tensor(np.asarray(sd[k]))
                sd[k] = operand_round(v.to(torch.float64)).numpy()
        inputs_btd = operand_round(torch.as_tensor(np.asarray(inputs_btd), dtype=torch.float64)).numpy()
    layers = params_from_state_dict(sd, enc_layers, 2, dtype, prefix='encoder/')
    P = decoder_params(sd, dtype)
    D = 'attention_decoder/decoder/'
    A = D + 'attention_layer/'
    ap = {k[len(A):]: v for k, v in P.items() if k.startswith(A)}
    x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype)
    sl = torch.as_tensor(np.asarray(inputs_seq_len), dtype=torch.long)
    lsl = torch.as_tensor(np.asarray(labels_seq_len), dtype=torch.long)
    lab = torch.as_tensor(np.asarray(labels), dtype=torch.long)
    peep = layers[0][0]['_peep']
    enc_kw = dict(h_round=operand_round) if operand_round is not None else {}
    enc_tm, final = olstm.blstm_encoder(x, sl, layers, None, forget_bias=1.0, cell_clip=clip_enc,
                                        use_peephole=peep, **enc_kw)
    enc = enc_tm.transpose(0, 1)                                          # [B,T,2H]
    B, T, E2 = enc.shape
    # bridge: flatten (c_fw, h_fw, c_bw, h_bw) -> FC -> (c0, h0)
    (c_fw, h_fw), (c_bw, h_bw) = final
    bi = torch.cat([c_fw, h_fw, c_bw, h_bw], dim=1)
    init = bi @ P['bridge/fully_connected/weights'] + P['bridge/fully_connected/biases']
    U = init.shape[1] // 2
    c, h = init[:, :U], init[:, U:]
    emb_w = P['output_embedding/W_embedding']
    emb = emb_w[lab]                                                      # [B,Lmax,E]
    if drop_emb is not None:
        emb = emb * torch.as_tensor(drop_emb, dtype=dtype)
    To = int(lsl.max()) - 1
    keys = compute_keys(ap, att_type, enc, bwd_round)
    cell = dict(w=P[D + 'lstm_cell/kernel'], b=P[D + 'lstm_cell/bias'])
    cell_mm = None if bwd_round is None else (lambda a, w: _MMBwdRound.apply(a, w, bwd_round, False))
    av_mm = (lambda a, w: a @ w) if bwd_round is None else (lambda a, w: _MMBwdRound.apply(a, w, bwd_round, True))
    has_peep = (D + 'lstm_cell/w_i_diag') in P
    z = torch.zeros(U, dtype=dtype)
    wci, wcf, wco = (P[D + 'lstm_cell/w_i_diag'], P[D + 'lstm_cell/w_f_diag'], P[D + 'lstm_cell/w_o_diag']) \
        if has_peep else (z, z, z)
    ctx = x.new_zeros(B, E2)
    logits_steps, alphas, ids = [], [], []
    carry = prev_alpha == 'carry' and att_type in ('location', 'hybrid')
    a_prev = x.new_zeros(B, T) if carry else None             # AttentionDecoder.initialize(): zeros
    for k in range(To):
        fin_prev = (k >= (lsl - 1)).to(dtype).unsqueeze(1)               # finished BEFORE this step
        inp_emb = emb[:, k] if k == 0 else emb[:, k] * (1.0 - fin_prev_in)
        inp = torch.cat([inp_emb, ctx], dim=1)
        cn, hn = olstm.lstm_block_cell(inp, c, h, cell['w'], cell['b'], wci, wcf, wco, 1.0, clip_dec, has_peep, mm=cell_mm)
        cell_out = hn if drop_dec is None else hn * torch.as_tensor(drop_dec[k], dtype=dtype)
        alpha, ctx_k = attention_step(ap, att_type, enc, keys, cell_out, sl, sharpening, sigmoid_smoothing, a_prev,
                                      bwd_round=bwd_round)
        if carry:
            a_prev = alpha
        av = torch.tanh(av_mm(torch.cat([cell_out, ctx_k], dim=1), P[D + 'attentional_vector/weights']))
        lg = av @ P[D + 'output_layer/weights'] + P[D + 'output_layer/biases']
        live = 1.0 - fin_prev                                             # impute_finished
        logits_steps.append(lg * live)
        alphas.append(alpha * live)
        ids.append(torch.argmax(lg, dim=1) * live.squeeze(1).long())
        c = live * cn + fin_prev * c
        h = live * hn + fin_prev * h
        ctx = ctx_k * live                     # outputs (incl. context) are zeroed once finished
        # TrainingHelper.next_inputs: zeros once time+1 >= sequence_length
        fin_prev_in = ((k + 1) >= (lsl - 1)).to(dtype).unsqueeze(1)
    logits = torch.stack(logits_steps, dim=1) / temperature               # [B,To,C]
    lg = logits + 1e-10
    targets = lab[:, 1:To + 1]
    w = (torch.arange(To).unsqueeze(0) < (lsl - 1).unsqueeze(1)).to(dtype)
    xent = torch.nn.functional.cross_entropy(lg.reshape(B * To, -1), targets.reshape(-1), reduction='none')
    seq_loss = (xent.view(B, To) * w).sum() / (w.sum() + 1e-12)
    out = dict(logits=logits.detach().numpy(), alphas=torch.stack(alphas, 1).detach().numpy(),
               predicted_ids=torch.stack(ids, 1).numpy(), sequence_loss=float(seq_loss.detach()))
    total = seq_loss
    if lambda_weight is not None:
        ctc_lg = (enc_tm.reshape(T * B, E2) @ P['ctc_output/weights'] + P['ctc_output/biases']).reshape(T, B, -1)
        ctc_losses = _ctc_loss(ctc_lg, ctc_labels, np.asarray(inputs_seq_len))
        total = (1.0 - lambda_weight) * seq_loss + lambda_weight * ctc_losses.mean()
        out['ctc_logits'] = ctc_lg.detach().numpy()
        out['ctc_losses'] = ctc_losses.detach().numpy()
    total.backward()
    grads = {}
    for name, t in P.items():
        grads[name] = t.grad.detach().numpy().copy() if t.grad is not None else np.zeros(tuple(t.shape))
    for layer in layers:
        for p in layer:
            base = p['_base']
            grads[base + '/kernel'] = p['w'].grad.numpy().copy()
            grads[base + '/bias'] = p['b'].grad.numpy().copy()
            if p['_peep']:
                grads[base + '/w_i_diag'] = p['wci'].grad.numpy().copy()
                grads[base + '/w_f_diag'] = p['wcf'].grad.numpy().copy()
                grads[base + '/w_o_diag'] = p['wco'].grad.numpy().copy()
    out.update(total_loss=float(total.detach()), grads=grads, enc=enc.detach().numpy())
    return out


def attention_model_infer(sd, inputs_btd, inputs_seq_len, enc_layers, att_type, sos, eos, max_len,
                          clip_enc=0.0, clip_dec=0.0, sharpening=1.0, dtype=torch.float64,
                          sigmoid_smoothing=False, prev_alpha='zeros'):
    """GreedyEmbeddingHelper decode (attention_seq2seq.py:462-509): returns predicted ids [B, <=max_len]."""
    from .model import params_from_state_dict
    with torch.no_grad():
        layers = params_from_state_dict(sd, enc_layers, 2, dtype, requires_grad=False, prefix='encoder/')
        P = decoder_params(sd, dtype, requires_grad=False)
        D = 'attention_decoder/decoder/'
        A = D + 'attention_layer/'
        ap = {k[len(A):]: v for k, v in P.items() if k.startswith(A)}
        x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype)
        sl = torch.as_tensor(np.asarray(inputs_seq_len), dtype=torch.long)
        peep = layers[0][0]['_peep']
        enc_tm, final = olstm.blstm_encoder(x, sl, layers, None, forget_bias=1.0, cell_clip=clip_enc,
                                            use_peephole=peep)
        enc = enc_tm.transpose(0, 1)
        B = enc.shape[0]
        (c_fw, h_fw), (c_bw, h_bw) = final
        init = torch.cat([c_fw, h_fw, c_bw, h_bw], 1) @ P['bridge/fully_connected/weights'] + \
            P['bridge/fully_connected/biases']
        U = init.shape[1] // 2
        c, h = init[:, :U], init[:, U:]
        emb_w = P['output_embedding/W_embedding']
        keys = compute_keys(ap, att_type, enc)
        has_peep = (D + 'lstm_cell/w_i_diag') in P
        z = torch.zeros(U, dtype=dtype)
        wci, wcf, wco = (P[D + 'lstm_cell/w_i_diag'], P[D + 'lstm_cell/w_f_diag'], P[D + 'lstm_cell/w_o_diag']) \
            if has_peep else (z, z, z)
        ctx = x.new_zeros(B, enc.shape[2])
        tok = torch.full((B,), sos, dtype=torch.long)
        finished = torch.zeros(B, dtype=torch.bool)
        out = []
        carry = prev_alpha == 'carry' and att_type in ('location', 'hybrid')
        a_prev = x.new_zeros(B, enc.shape[1]) if carry else None
        for k in range(max_len):
            inp = torch.cat([emb_w[tok], ctx], 1)
            cn, hn = olstm.lstm_block_cell(inp, c, h, P[D + 'lstm_cell/kernel'], P[D + 'lstm_cell/bias'],
                                           wci, wcf, wco, 1.0, clip_dec, has_peep)
            alpha, ctx_k = attention_step(ap, att_type, enc, keys, hn, sl, sharpening, sigmoid_smoothing, a_prev)
            if carry:
                a_prev = alpha
            av = torch.tanh(torch.cat([hn, ctx_k], 1) @ P[D + 'attentional_vector/weights'])
            lg = av @ P[D + 'output_layer/weights'] + P[D + 'output_layer/biases']
            sample = torch.argmax(lg, 1)
            live = ~finished
            out.append(torch.where(live, sample, torch.zeros_like(sample)))
            lf = live.to(dtype).unsqueeze(1)
            c = lf * cn + (1 - lf) * c
            h = lf * hn + (1 - lf) * h
            ctx = ctx_k * lf
            finished = finished | (sample == eos)
            tok = torch.where(finished, torch.full_like(tok, sos), sample)
            if bool(finished.all()):
                break
        return torch.stack(out, 1).numpy()

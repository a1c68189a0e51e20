"""Oracle (test infrastructure; PINNED to the reference's own CTC.compute_loss as executed on the eager TensorFlow
stand-in -- blstm, lstm, bottleneck, vgg_blstm, LSTMCell + num_proj, gru, bgru: encoder outputs, logits, loss, every
gradient, per-variable clip; tests/test_oracle_tfshim.py::test_tfshim_ctc_models -- see oracle/__init__.py): the CTC
model of the reference end to end on the CPU.

Follows models/ctc/ctc.py:175-323: encoder (oracle.lstm) -> reshape [T*B, 2H] ->
output FC (weights [2H,C], biases) -> logits [T,B,C] -> tf.nn.ctc_loss -> batch mean
(+ weight_decay * sum l2_loss(non-bias vars), ctc.py:280-286).  Gradients of every
variable come from torch autograd through the restated forward; the CTC gradient is
the explicit alpha-beta formula of oracle.ctc plugged in as a custom Function.
"""
import numpy as np
import torch

from . import ctc as octc
from . import lstm as olstm


class _CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels_list, seq_len):
        loss, grad = octc.ctc_loss_batch(logits.detach().cpu().numpy().astype(np.float64),
                                         labels_list, seq_len)
        ctx.save_for_backward(torch.from_numpy(grad).to(logits.dtype))
        return torch.from_numpy(loss).to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.view(1, -1, 1), None, None


def ctc_loss(logits_tbc, labels_list, seq_len):
    return _CTCLossFn.apply(logits_tbc, labels_list, np.asarray(seq_len))


def params_from_state_dict(sd, num_layers, ndir=2, dtype=torch.float64, requires_grad=True, prefix=''):
    """{TF variable name -> array} -> oracle layer dicts.  Names: SURVEY.md Appendix C."""
    def t(name):
        v = torch.as_tensor(np.asarray(sd[name].detach().cpu() if torch.is_tensor(sd[name]) else sd[name]),
                            dtype=dtype).clone()
        v.requires_grad_(requires_grad)
        return v
    layers = []
    for i in range(1, num_layers + 1):
        dirs = []
        for d in (['fw', 'bw'] if ndir == 2 else ['fw']):
            if ndir == 2:
                base = '%sblstm_hidden%d/%s/lstm_cell' % (prefix, i, d)
            else:
                base = '%smulti_lstm/multi_rnn_cell/cell_%d/lstm_cell' % (prefix, i - 1)
            p = dict(w=t(base + '/kernel'), b=t(base + '/bias'), _base=base)
            if base + '/w_i_diag' in sd:
                p.update(wci=t(base + '/w_i_diag'), wcf=t(base + '/w_f_diag'), wco=t(base + '/w_o_diag'))
                p['_peep'] = True
            else:
                H = p['b'].shape[0] // 4
                z = torch.zeros(H, dtype=dtype)
                p.update(wci=z, wcf=z, wco=z)
                p['_peep'] = False
            dirs.append(p)
        layers.append(tuple(dirs) if ndir == 2 else dirs[0])
    return layers


def ctc_model_forward(sd, inputs_btd, labels_list, seq_len, num_layers, ndir=2, cell_clip=0.0,
                      weight_decay=0.0, drop_masks=None, dtype=torch.float64, temperature=1.0, vgg=None,
                      bottleneck=False, operand_round=None, want_grads=True):
    """Returns dict(total_loss, ctc_losses [B], logits [T,B,C], grads {name: array}).
    operand_round (e.g. oracle.lstm.bf16_round_t): reproduces the rounding points of the bf16-operand device path in
    the forward -- inputs, LSTM kernels, output weights, VGG filters / activations / bridge and every emitted / fed-back
    h are rounded (straight-through),
    state, biases, peepholes and all accumulation stay in `dtype`."""
    if operand_round is not None:
        sd = dict(sd)
        for k in list(sd):
            if k.endswith('/kernel') or k == 'output/weights' or k.endswith('/weight') or k == 'bridge/weights':
                v = sd[k].detach().cpu() if torch.is_tensor(sd[k]) else torch.as_tensor(np.asarray(sd[k]))
                sd[k] = operand_round(v.to(torch.float64)).numpy()
        inputs_btd = operand_round(torch.as_tensor(np.asarray(inputs_btd), dtype=torch.float64)).numpy()
    layers = params_from_state_dict(sd, num_layers, ndir, dtype)
    w_out = torch.as_tensor(np.asarray(sd['output/weights'].detach().cpu() if torch.is_tensor(sd['output/weights']) else sd['output/weights']), dtype=dtype).clone().requires_grad_(True)
    b_out = torch.as_tensor(np.asarray(sd['output/biases'].detach().cpu() if torch.is_tensor(sd['output/biases']) else sd['output/biases']), dtype=dtype).clone().requires_grad_(True)
    x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype)
    sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    vgg_params = {}
    if vgg is not None:      # VGG front-end (models/encoders/core/vgg_blstm.py:107-177): vgg = (F, W)
        from . import vgg as ovgg
        for k in sd:
            if k.startswith('VGG') or k.startswith('bridge/'):
                vgg_params[k] = torch.as_tensor(np.asarray(sd[k].detach().cpu() if torch.is_tensor(sd[k]) else sd[k]), dtype=dtype).clone().requires_grad_(True)
        x = ovgg.vgg_frontend(x, vgg_params, vgg[0], vgg[1], act_round=operand_round)
        # frames past seq_len feed the LSTM but are masked there, exactly as in the reference
    peep = layers[0][0]['_peep'] if ndir == 2 else layers[0]['_peep']
    kw = dict(forget_bias=1.0, cell_clip=cell_clip, use_peephole=peep)
    if operand_round is not None:
        kw['h_round'] = operand_round
    if ndir == 2:
        enc, final = olstm.blstm_encoder(x, sl, layers, drop_masks, **kw)
    else:
        enc, final = olstm.lstm_encoder(x, sl, layers, drop_masks, **kw)
    T, B, E = enc.shape
    head_in = enc.reshape(T * B, E)
    bn_params = {}
    if bottleneck:   # models/ctc/ctc.py:201-216: fully_connected(relu) under scope 'bottleneck'
        for k in ('bottleneck/weights', 'bottleneck/biases'):
            bn_params[k] = torch.as_tensor(np.asarray(sd[k].detach().cpu() if torch.is_tensor(sd[k]) else sd[k]),
                                           dtype=dtype).clone().requires_grad_(True)
        head_in = torch.relu(head_in @ bn_params['bottleneck/weights'] + bn_params['bottleneck/biases'])
    logits = (head_in @ w_out + b_out).reshape(T, B, -1)
    losses = ctc_loss(logits / temperature, labels_list, seq_len)
    total = losses.mean()
    named = {}
    for li, layer in enumerate(layers):
        for p in (layer if ndir == 2 else (layer,)):
            base = p['_base']
            named[base + '/kernel'] = p['w']
            named[base + '/bias'] = p['b']
            if p['_peep']:
                named[base + '/w_i_diag'] = p['wci']
                named[base + '/w_f_diag'] = p['wcf']
                named[base + '/w_o_diag'] = p['wco']
    named['output/weights'] = w_out
    named['output/biases'] = b_out
    named.update(vgg_params)
    named.update(bn_params)
    if weight_decay > 0:
        l2 = sum(0.5 * (v ** 2).sum() for n, v in named.items() if 'bias' not in n.lower())
        total = total + weight_decay * l2
    grads = None
    if want_grads:
        total.backward()
        grads = {n: v.grad.detach().numpy().copy() for n, v in named.items()}
    return dict(total_loss=float(total.detach()), ctc_losses=losses.detach().numpy(),
                logits=logits.detach().numpy(), grads=grads, enc=enc.detach().numpy(),
                final=final)


def multitask_ctc_model_forward(sd, inputs_btd, labels_main, labels_sub, seq_len, num_layers_main, num_layers_sub,
                                main_task_weight, ndir=2, cell_clip=0.0, weight_decay=0.0, bottleneck=False,
                                dtype=torch.float64, proj=False):
    """models/ctc/multitask_ctc.py:100-312: one encoder, the sub head ('output_sub') on the outputs of layer
    num_layers_sub (models/encoders/core/blstm.py:326-328), the main head ('output_main', behind 'bottleneck' if
    present) on the top layer; total = w * mean CTC(main) + (1 - w) * mean CTC(sub) (+ weight decay).
    For ndir == 1 the caller passes num_layers_sub = num_layers_main (the list alias of lstm.py:271-272).
    Returns dict(total_loss, ctc_losses_main, ctc_losses_sub, logits_main, logits_sub, grads)."""
    def t(name):
        v = sd[name]
        return torch.as_tensor(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v), dtype=dtype).clone() \
            .requires_grad_(True)
    layers = params_from_state_dict(sd, num_layers_main, ndir, dtype)
    x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype)
    sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    peep = layers[0][0]['_peep'] if ndir == 2 else layers[0]['_peep']
    kw = dict(forget_bias=1.0, cell_clip=cell_clip, use_peephole=peep)
    run = olstm.blstm_encoder if ndir == 2 else olstm.lstm_encoder
    if proj:   # lstm_impl='LSTMCell' + num_proj (multitask_blstm.py:95 hands it to the cell builder; bidirectional here)
        assert ndir == 2
        for layer in layers:
            for p in layer:
                p['w_proj'] = t(p['_base'] + '/projection/kernel')
        run = olstm.blstmp_encoder
    enc, _ = run(x, sl, layers, None, **kw)
    # the lower layers evaluated again with the SAME parameter tensors: identical values, and autograd adds the
    # two paths' gradients exactly as the shared graph of the reference does
    enc_sub, _ = run(x, sl, layers[:num_layers_sub], None, **kw)
    T, B, E = enc.shape
    heads = {k: t(k) for k in ('output_main/weights', 'output_main/biases', 'output_sub/weights',
                               'output_sub/biases')}
    head_in = enc.reshape(T * B, E)
    if bottleneck:
        heads['bottleneck/weights'], heads['bottleneck/biases'] = t('bottleneck/weights'), t('bottleneck/biases')
        head_in = torch.relu(head_in @ heads['bottleneck/weights'] + heads['bottleneck/biases'])
    logits_main = (head_in @ heads['output_main/weights'] + heads['output_main/biases']).reshape(T, B, -1)
    logits_sub = (enc_sub.reshape(T * B, E) @ heads['output_sub/weights'] + heads['output_sub/biases']) \
        .reshape(T, B, -1)
    lm = ctc_loss(logits_main, labels_main, seq_len)
    ls = ctc_loss(logits_sub, labels_sub, seq_len)
    total = main_task_weight * lm.mean() + (1.0 - main_task_weight) * ls.mean()
    named = {}
    for layer in layers:
        for p in (layer if ndir == 2 else (layer,)):
            base = p['_base']
            named[base + '/kernel'], named[base + '/bias'] = p['w'], p['b']
            if p['_peep']:
                named[base + '/w_i_diag'], named[base + '/w_f_diag'], named[base + '/w_o_diag'] = \
                    p['wci'], p['wcf'], p['wco']
            if proj:
                named[base + '/projection/kernel'] = p['w_proj']
    named.update(heads)
    if weight_decay > 0:
        total = total + weight_decay * sum(0.5 * (v ** 2).sum() for n, v in named.items() if 'bias' not in n.lower())
    total.backward()
    grads = {n: v.grad.detach().numpy().copy() for n, v in named.items()}
    return dict(total_loss=float(total.detach()), ctc_losses_main=lm.detach().numpy(),
                ctc_losses_sub=ls.detach().numpy(), logits_main=logits_main.detach().numpy(),
                logits_sub=logits_sub.detach().numpy(), grads=grads)


def gru_ctc_model_forward(sd, inputs_btd, labels_list, seq_len, num_layers, ndir=2, drop_masks=None, want_grads=True,
                          dtype=torch.float64):
    """GRU / BGRU encoder (models/encoders/core/gru.py) + output FC + CTC, as CTC(encoder_type='gru' | 'bgru') builds
    it (models/ctc/ctc.py:150-155, :198-233).  sd: {variable name -> array} with the TF 1.3 names.  Returns
    dict(total_loss, ctc_losses, logits [T,B,C], grads {name: array}, enc, final)."""
    from . import gru as ogru
    named = {}

    def t(name):
        v = sd[name]
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        named[name] = torch.as_tensor(v, dtype=dtype).clone().requires_grad_(True)
        return named[name]

    layers = []
    for i in range(1, num_layers + 1):
        bases = ['bgru_hidden%d/%s/gru_cell' % (i, d) for d in ('fw', 'bw')] if ndir == 2 else \
            ['multi_gru/rnn/multi_rnn_cell/cell_%d/gru_cell' % (i - 1)]
        ps = [dict(wg=t(b + '/gates/kernel'), bg=t(b + '/gates/bias'), wc=t(b + '/candidate/kernel'),
                   bc=t(b + '/candidate/bias')) for b in bases]
        layers.append(tuple(ps) if ndir == 2 else ps[0])
    w_out, b_out = t('output/weights'), t('output/biases')
    x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype).transpose(0, 1)
    sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    enc, final = ogru.gru_encoder(x, sl, layers, ndir, drop_masks)
    T, B, E = enc.shape
    logits = (enc.reshape(T * B, E) @ w_out + b_out).reshape(T, B, -1)
    losses = ctc_loss(logits, labels_list, seq_len)
    total = losses.mean()
    grads = None
    if want_grads:
        total.backward()
        grads = {n: v.grad.detach().numpy().copy() for n, v in named.items()}
    return dict(total_loss=float(total.detach()), ctc_losses=losses.detach().numpy(), logits=logits.detach().numpy(),
                grads=grads, enc=enc.detach().numpy(), final=final)


def cldnn_ctc_model_forward(sd, inputs_btd, labels_list, seq_len, num_layers, F, W, cell_clip=0.0, want_grads=True,
                            dtype=torch.float64, operand_round=None, proj=False):
    """CTC(encoder_type='cldnn_wang'): conv stack -> BLSTM stack -> fc1 relu -> fc2 relu -> output FC -> CTC
    (models/encoders/core/cldnn_wang.py:134-249, models/ctc/ctc.py:135-147).  No dropout.
    operand_round: the rounding points of the bf16-operand device path (inputs, every matrix operand, every stored
    activation and emitted h; straight-through)."""
    from . import cldnn as ocl
    named = {}
    rnd = (lambda v: v) if operand_round is None else (lambda v: olstm.ste_round(v, operand_round))
    if operand_round is not None:
        sd = dict(sd)
        for k in list(sd):
            if k.endswith('/kernel') or k.endswith('/weights') or k.endswith('/weight'):
                v = sd[k].detach().cpu() if torch.is_tensor(sd[k]) else torch.as_tensor(np.asarray(sd[k]))
                sd[k] = operand_round(v.to(torch.float64)).numpy()
        inputs_btd = operand_round(torch.as_tensor(np.asarray(inputs_btd), dtype=torch.float64)).numpy()

    def t(name):
        v = sd[name]
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        named[name] = torch.as_tensor(v, dtype=dtype).clone().requires_grad_(True)
        return named[name]

    P = {n: t(n) for n in sd if n.startswith('CNN')}
    layers = params_from_state_dict(sd, num_layers, 2, dtype)
    x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype)
    sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    feat = ocl.conv_stack(x, P, F, W, act_round=operand_round)              # [B,T,h*w*96] (the LSTM oracle takes batch-major)
    kw = dict(forget_bias=1.0, cell_clip=cell_clip, use_peephole=layers[0][0]['_peep'])
    if operand_round is not None:
        kw['h_round'] = operand_round
    if proj:   # lstm_impl='LSTMCell' + num_proj (cldnn_wang.py:202 hands it to the cell builder of blstm.py:187-230)
        assert operand_round is None
        for layer in layers:
            for p in layer:
                p['w_proj'] = t(p['_base'] + '/projection/kernel')
        enc, final = olstm.blstmp_encoder(feat, sl, layers, None, **kw)
    else:
        enc, final = olstm.blstm_encoder(feat, sl, layers, None, **kw)
    T, B, E = enc.shape
    a1 = rnd(torch.relu(enc.reshape(T * B, E) @ t('fc1/weights') + t('fc1/biases')))
    a2 = rnd(torch.relu(a1 @ t('fc2/weights') + t('fc2/biases')))
    logits = (a2 @ t('output/weights') + t('output/biases')).reshape(T, B, -1)
    losses = ctc_loss(logits, labels_list, seq_len)
    total = losses.mean()
    for layer in layers:
        for p in layer:
            base = p['_base']
            named[base + '/kernel'], named[base + '/bias'] = p['w'], p['b']
            if p['_peep']:
                named[base + '/w_i_diag'], named[base + '/w_f_diag'], named[base + '/w_o_diag'] = p['wci'], p['wcf'], p['wco']
    grads = None
    if want_grads:
        total.backward()
        grads = {n: v.grad.detach().numpy().copy() for n, v in named.items()}
    return dict(total_loss=float(total.detach()), ctc_losses=losses.detach().numpy(), logits=logits.detach().numpy(),
                grads=grads, enc=a2.detach().numpy().reshape(T, B, -1))



def lstmp_ctc_model_forward(sd, inputs_btd, labels_list, seq_len, num_layers, cell_clip=0.0, want_grads=True,
                            dtype=torch.float64, vgg=None, weight_decay=0.0):
    """CTC(encoder_type='blstm', lstm_impl='LSTMCell', num_proj=P): stacked bidirectional projected LSTM cells
    (models/encoders/core/blstm.py:187-230, tf.contrib.rnn.LSTMCell(num_proj)) -> output FC on the [T,B,2P] outputs ->
    CTC.  Variables: blstm_hidden<i>/{fw,bw}/lstm_cell/{kernel [(Din+P),4H], bias, w_{i,f,o}_diag, projection/kernel
    [H,P]}.  Returns dict(total_loss, ctc_losses, logits, grads, enc, final).
    vgg = (F, W): the VGG front-end of ctc_model_forward in front of the projected stack (CTC(encoder_type='vgg_blstm',
    lstm_impl='LSTMCell', num_proj=P): models/encoders/core/vgg_blstm.py:107-190 hands num_proj to the same cell builder)."""
    named = {}

    def t(name):
        v = sd[name]
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        named[name] = torch.as_tensor(v, dtype=dtype).clone().requires_grad_(True)
        return named[name]

    layers = []
    peep = 'blstm_hidden1/fw/lstm_cell/w_i_diag' in sd
    for i in range(1, num_layers + 1):
        dirs = []
        for d in ('fw', 'bw'):
            b = 'blstm_hidden%d/%s/lstm_cell' % (i, d)
            p = dict(w=t(b + '/kernel'), b=t(b + '/bias'), w_proj=t(b + '/projection/kernel'))
            if peep:
                p.update(wci=t(b + '/w_i_diag'), wcf=t(b + '/w_f_diag'), wco=t(b + '/w_o_diag'))
            else:
                z = torch.zeros(p['b'].shape[0] // 4, dtype=dtype)
                p.update(wci=z, wcf=z, wco=z)
            dirs.append(p)
        layers.append(tuple(dirs))
    w_out, b_out = t('output/weights'), t('output/biases')
    x = torch.as_tensor(np.asarray(inputs_btd), dtype=dtype)
    sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    if vgg is not None:
        from . import vgg as ovgg
        vgg_params = {k: t(k) for k in sd if k.startswith('VGG') or k.startswith('bridge/')}
        x = ovgg.vgg_frontend(x, vgg_params, vgg[0], vgg[1])
    enc, final = olstm.blstmp_encoder(x, sl, layers, None, forget_bias=1.0, cell_clip=cell_clip, use_peephole=peep)
    T, B, E = enc.shape
    logits = (enc.reshape(T * B, E) @ w_out + b_out).reshape(T, B, -1)
    losses = ctc_loss(logits, labels_list, seq_len)
    total = losses.mean()
    if weight_decay > 0:   # models/ctc/ctc.py:280-286: every variable without 'bias' in its name
        total = total + weight_decay * sum(0.5 * (v ** 2).sum() for n, v in named.items() if 'bias' not in n.lower())
    grads = None
    if want_grads:
        total.backward()
        grads = {n: v.grad.detach().numpy().copy() for n, v in named.items()}
    return dict(total_loss=float(total.detach()), ctc_losses=losses.detach().numpy(), logits=logits.detach().numpy(),
                grads=grads, enc=enc.detach().numpy(), final=final)

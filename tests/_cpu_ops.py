"""TEST INFRASTRUCTURE ONLY: a torch-CPU stand-in for the C-ABI front end (package `ops`), so that the HOST logic
above the kernels -- layer wiring, gradient plumbing, head / loss composition, clip + optimizer sequencing -- runs
in the `-m "not gpu"` suite.  install(monkeypatch) swaps the functions of the `ops` module for the duration of one
test; the product never imports this file and `ops` itself still refuses non-CUDA tensors.

The stand-ins are built from the oracle (oracle/lstm.py, oracle/ctc.py, oracle/optim.py) and plain torch; they are
self-consistent rather than layout-identical to the kernels: the "packed" recurrent weights are W_h as stored, the
"interleaved" gate layout is the plain i, ci, f, o column order, and the saved-activation tensor of lstm_fwd carries
the autograd graph that lstm_bwd differentiates.  Kernel numerics are NOT tested here -- that is what the GPU
parity tests are for."""
import numpy as np
import torch

from oracle import ctc as octc
from oracle import decoders as odec
from oracle import lstm as olstm
from oracle import optim as oopt

F64 = torch.float64


class _NullLane(object):
    def __init__(self, device, keep=(), lane=1, after=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _bt_to_tb(x_btd, dtype=0, ld=None):
    y = x_btd.transpose(0, 1).contiguous()
    if ld is not None and ld > y.shape[2]:
        y = torch.cat([y, y.new_zeros(y.shape[0], y.shape[1], ld - y.shape[2])], 2)
    return y


def _transpose2d(x, out=None):
    y = x.t().contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


def _cast_from_f32(x, dtype, out=None):
    if out is not None:
        out.copy_(x)
        return out
    return x.clone()


def _cast_to_f32(x, out=None):
    y = x.float()
    if out is not None:
        out.copy_(y)
        return out
    return y


def _apply_mask(x, mask, out=None):
    y = x * mask.to(x.dtype).view(x.shape) if mask.numel() == x.numel() else x * mask
    if out is not None:
        out.copy_(y)
        return out
    return y


def _dropout_mask(shape, keep_prob, seed, offset, device=None):
    """Counter-based like the device generator (element e of a tensor = word e % 4 of block offset + e / 4), so that a
    mask formed chunk by chunk with shifted offsets equals the slice of the whole one; the bits are splitmix64's, not
    Philox's -- the stand-ins only have to be self-consistent."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        z = np.uint64(int(offset) % (1 << 62)) * np.uint64(4) + np.arange(n, dtype=np.uint64)
        z = z + np.uint64(int(seed) % (1 << 62)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy(((u < keep_prob) / float(keep_prob)).astype(np.float32)).view(*shape)


def _dropout_apply(x, keep_prob, seed, offset):
    return _apply_mask(x, _dropout_mask(tuple(x.shape), keep_prob, seed, offset))


def _colsum(a, out=None):
    y = a.double().sum(0).float()
    if out is not None:
        out.copy_(y)
        return out
    return y


def _gemm(A, B, transA=False, transB=False, bias=None, out=None, out_dtype=None, accumulate=False, relu=False,
          mul=None, drop=None):
    if A.dim() != 2 or B.dim() != 2 or A.stride(1) != 1 or B.stride(1) != 1:
        raise ValueError('gemm: operands must be 2-D with unit inner stride')
    a = A.double().t() if transA else A.double()
    b = B.double().t() if transB else B.double()
    if a.shape[1] != b.shape[0]:
        raise ValueError('gemm: inner dimensions differ (%d vs %d)' % (a.shape[1], b.shape[0]))
    y = a @ b
    if bias is not None:
        y = y + bias.double()
    if accumulate:
        y = y + out.double()
    if relu:
        y = torch.relu(y)
    if mul is not None:
        y = y * mul.double()
    if drop is not None:
        y = y * _dropout_mask(tuple(y.shape), *drop).double()
    if out is not None:
        if tuple(out.shape) != tuple(y.shape) or out.stride(1) != 1:
            raise ValueError('gemm: bad out shape %s' % (tuple(out.shape),))
        out.copy_(y)
        return out
    return y.float()


def _relu_bwd(dout, out, mask=None, drop=None):
    y = dout * (out > 0).to(dout.dtype)
    if drop is not None:
        mask = _dropout_mask(tuple(y.shape), *drop)
    if mask is not None:
        y = y * mask.view(y.shape)
    return y


def _lstm_prep_weights(kernel, bias, din, H, dtype, out=None):
    if kernel.shape != (din + H, 4 * H):
        raise ValueError('kernel must be [Din+H,4H]')
    if out is None:
        out = dict(wx_il=torch.empty((din, 4 * H)), bias_il=torch.empty((4 * H,)),
                   pf=torch.empty((H * 4 * H,)), pb=torch.empty((H * 4 * H,)))
    out['wx_il'].copy_(kernel[:din])
    out['bias_il'].copy_(bias)
    out['pf'].copy_(kernel[din:].reshape(-1))
    out['pb'].copy_(kernel[din:].reshape(-1))
    return out


def _lstm_prep_layer(variables, din, H, dtype, ldk=None):
    ldk = din if ldk is None else ldk
    for v in variables:
        if v[0].shape != (din + H, 4 * H):
            raise ValueError('kernel must be [Din+H,4H]')
    pad = lambda w: torch.cat([w, w.new_zeros(w.shape[0], ldk - din)], 1) if ldk > din else w
    has_peep = len(variables[0]) >= 5 and variables[0][2] is not None
    return dict(wxT=torch.cat([pad(v[0][:din].t()) for v in variables], 0).contiguous(),
                wx_cat=torch.cat([v[0][:din] for v in variables], 1).contiguous(),
                bias=torch.cat([v[1] for v in variables]).clone(),
                whf=torch.stack([v[0][din:].reshape(-1) for v in variables]).clone(),
                whb=torch.stack([v[0][din:].reshape(-1) for v in variables]).clone(),
                peep=torch.stack([torch.stack([v[2], v[3], v[4]]) for v in variables]) if has_peep else None)


def _lstm_grad_finish(grads, dw_il, dpeep, H):
    for d, g in enumerate(grads):
        g[0].copy_(dw_il[d])
        g[1].copy_(dpeep[d, 3:7].reshape(-1))
        if len(g) >= 5 and g[2] is not None:
            for k in range(3):
                g[2 + k].copy_(dpeep[d, k])


def _gate_deinterleave(src, dst, H):
    dst.copy_(src)
    return dst


def _lstm_graph(xproj, wh_packed, peep, sl, H, ndir, forget_bias, cell_clip, clip_blocks_gradient=False):
    eye = torch.eye(4 * H, dtype=F64)
    runs = []
    for d in range(ndir):
        xp = xproj[:, :, d * 4 * H:(d + 1) * 4 * H].double().clone().requires_grad_(True)
        wh = wh_packed[d].view(H, 4 * H).double()
        if peep is not None:
            pp = peep[d].double().clone().requires_grad_(True)
            wci, wcf, wco = pp[0], pp[1], pp[2]
        else:
            pp = None
            wci = wcf = wco = torch.zeros(H, dtype=F64)
        p = dict(w=torch.cat([eye, wh], 0), b=torch.zeros(4 * H, dtype=F64), wci=wci, wcf=wcf, wco=wco)
        out, (cf, hf) = olstm.dynamic_rnn(xp, sl, p, reverse=(d == 1), forget_bias=forget_bias,
                                          cell_clip=cell_clip, use_peephole=peep is not None,
                                          clip_blocks_gradient=clip_blocks_gradient)
        runs.append(dict(xp=xp, pp=pp, out=out, cf=cf, hf=hf))
    return runs


def _lstm_fwd(xproj, wh_packed, peep, seq_len, H, ndir, dtype, forget_bias=1.0, cell_clip=0.0, want_final=True):
    T, B, G = xproj.shape
    if G != ndir * 4 * H:
        raise ValueError('xproj last dim %d != ndir*4H' % G)
    sl = seq_len.long()
    runs = _lstm_graph(xproj, wh_packed, peep, sl, H, ndir, forget_bias, cell_clip)
    outs, cfs, hfs = ([r[k].detach() for r in runs] for k in ('out', 'cf', 'hf'))
    gates = torch.zeros((T, B, ndir * 4 * H))
    gates._runs = runs                       # the graph lstm_bwd differentiates (stand-in for the saved gates)
    # what a gradient-blocking clip needs to build its own graph (asr_lstm_bwd_ex)
    gates._again = (xproj, wh_packed, peep, sl, H, ndir, forget_bias, cell_clip)
    hout = torch.cat(outs, 2).float().contiguous()
    cs = torch.zeros((T, B, ndir * H))
    return gates, hout, cs, torch.stack(cfs).float(), torch.stack(hfs).float()


def _lstm_bwd(dhout, gates, cs, wh_packed_bwd, peep, seq_len, H, ndir, dtype, d_c_final=None, d_h_final=None,
              want_dpeep=True, clip_no_grad=0.0):
    T, B, _ = dhout.shape
    dgates = torch.zeros((T, B, ndir * 4 * H))
    dpeep = torch.zeros((ndir, 7, H)) if want_dpeep else None
    runs = gates._runs
    if clip_no_grad and clip_no_grad > 0:    # LSTMCell's tf.clip_by_value: the same forward, a clamp that passes nothing back
        assert abs(gates._again[-1] - clip_no_grad) < 1e-12, 'clip_no_grad must be the clip the forward applied'
        runs = _lstm_graph(*gates._again, clip_blocks_gradient=True)
    for d, r in enumerate(runs):
        outs = [r['out'], r['cf'], r['hf']]
        gos = [dhout[:, :, d * H:(d + 1) * H].double(),
               d_c_final[d].double() if d_c_final is not None else torch.zeros_like(r['cf']),
               d_h_final[d].double() if d_h_final is not None else torch.zeros_like(r['hf'])]
        ins = [r['xp']] + ([r['pp']] if r['pp'] is not None else [])
        gr = torch.autograd.grad(outs, ins, grad_outputs=gos, allow_unused=True)
        dxp = gr[0]
        dgates[:, :, d * 4 * H:(d + 1) * 4 * H] = dxp.float()
        if want_dpeep:
            if r['pp'] is not None and gr[1] is not None:
                dpeep[d, 0:3] = gr[1].float()
            dpeep[d, 3:7] = dxp.sum((0, 1)).view(4, H).float()
    return dgates, dpeep


def _gru_dir(xg, xc, wgh, wch, sl, reverse):
    T, B, _ = xg.shape
    H = wch.shape[0]
    if reverse:
        xg, xc = olstm.reverse_sequence(xg, sl), olstm.reverse_sequence(xc, sl)
    h = xg.new_zeros(B, H)
    outs, rhs = [], []
    for t in range(T):
        act = (t < sl).to(xg.dtype).unsqueeze(1)
        g = torch.sigmoid(xg[t] + h @ wgh)
        r, u = g[:, :H], g[:, H:]
        rh = r * h
        c = torch.tanh(xc[t] + rh @ wch)
        hn = u * h + (1 - u) * c
        outs.append(act * hn)
        rhs.append(act * rh)
        h = act * hn + (1 - act) * h
    out, rhv = torch.stack(outs), torch.stack(rhs)
    if reverse:
        out, rhv = olstm.reverse_sequence(out, sl), olstm.reverse_sequence(rhv, sl)
    return out, rhv, h


def _gru_fwd(xg, xc, wgh, wch, seq_len, tmax, H, ndir):
    T, B, G = xg.shape
    if G != ndir * 2 * H or tuple(xc.shape) != (T, B, ndir * H):
        raise ValueError('gru_fwd: xg / xc shapes do not match ndir, H')
    sl = seq_len.long()
    runs, outs, rhs, hfs = [], [], [], []
    for d in range(ndir):
        a = xg[:, :, d * 2 * H:(d + 1) * 2 * H].double().clone().requires_grad_(True)
        b = xc[:, :, d * H:(d + 1) * H].double().clone().requires_grad_(True)
        out, rhv, hf = _gru_dir(a, b, wgh[d].double(), wch[d].double(), sl, d == 1)
        runs.append(dict(xg=a, xc=b, out=out, hf=hf))
        outs.append(out.detach())
        rhs.append(rhv.detach())
        hfs.append(hf.detach())
    z = torch.zeros((T, B, ndir * H))
    hout = torch.cat(outs, 2).float().contiguous()
    return dict(r=z, u=z.clone(), c=z.clone(), rh=torch.cat(rhs, 2).float().contiguous(), hout=hout,
                h_final=torch.stack(hfs).float(), _runs=runs)


def _gru_bwd(dout, d_h_final, saved, wghT, wchT, seq_len, tmax, H, ndir):
    T, B, _ = dout.shape
    dgate = torch.zeros((T, B, ndir * 2 * H))
    dcand = torch.zeros((T, B, ndir * H))
    for d, r in enumerate(saved['_runs']):
        gos = [dout[:, :, d * H:(d + 1) * H].double(),
               d_h_final[d].double() if d_h_final is not None else torch.zeros_like(r['hf'])]
        ga, gb = torch.autograd.grad([r['out'], r['hf']], [r['xg'], r['xc']], grad_outputs=gos)
        dgate[:, :, d * 2 * H:(d + 1) * 2 * H] = ga.float()
        dcand[:, :, d * H:(d + 1) * H] = gb.float()
    return dgate, dcand


def _labels_list(labels_flat, label_offsets, B):
    flat = labels_flat.cpu().numpy()
    off = label_offsets.cpu().numpy()
    return [[int(v) for v in flat[off[b]:off[b + 1]]] for b in range(B)]


def _ctc_loss(logits, labels_flat, label_offsets, seq_len, max_label_len, grad_scale=1.0, want_grad=True):
    T, B, Cc = logits.shape
    labs = _labels_list(labels_flat, label_offsets, B)
    sl = seq_len.cpu().numpy()
    loss = np.zeros(B)
    grad = np.zeros((T, B, Cc))
    ninf = 0
    x = logits.double().numpy()
    for b in range(B):
        n = int(sl[b])
        if n == 0 and len(labs[b]) == 0:
            continue                                  # batch-padding row
        l, g, ok = octc.ctc_loss_single(x[:n, b], labs[b])
        if not ok:
            ninf += 1
            continue                                  # ignore_longer_outputs_than_inputs=True: loss 0, grad 0
        loss[b] = l
        grad[:n, b] = g
    g = torch.from_numpy(grad * grad_scale).float() if want_grad else None
    return torch.from_numpy(loss).float(), g, torch.tensor([ninf], dtype=torch.int32)


def _ctc_greedy_decode(logits, seq_len, blank=None):
    T, B, Cc = logits.shape
    blank = Cc - 1 if blank is None else blank
    lp = torch.log_softmax(logits.double(), 2).transpose(0, 1).numpy()
    out = torch.full((B, T), -1, dtype=torch.int32)
    n = torch.zeros((B,), dtype=torch.int32)
    for b in range(B):
        hyp = odec.greedy_decode(lp[b:b + 1], [int(seq_len[b])], blank)[0]
        n[b] = len(hyp)
        out[b, :len(hyp)] = torch.tensor([int(v) for v in hyp], dtype=torch.int32)
    return out, n


def _softmax_rows(x2d):
    return torch.softmax(x2d.double(), 1).float()


def _clip_by_norm_multi(flat_grads, plan, clip_norm):
    off = plan.offsets.cpu().numpy()
    g = flat_grads.numpy()
    for i in range(plan.num_tensors):
        g[off[i]:off[i + 1]] = oopt.clip_by_norm(g[off[i]:off[i + 1]], clip_norm)


def _weight_decay(flat_grads, flat_params, plan, decay_mask, wd, l2_out=None):
    off = plan.offsets.cpu().numpy()
    m = decay_mask.cpu().numpy()
    tot = 0.0
    for i in range(plan.num_tensors):
        if not m[i]:
            continue
        p = flat_params[off[i]:off[i + 1]]
        tot += 0.5 * float((p.double() ** 2).sum())
        if flat_grads is not None:
            flat_grads[off[i]:off[i + 1]] += wd * p
    if l2_out is not None:
        l2_out.fill_(wd * tot)


_OPT_NAMES = None


def _optimizer_step(opt_id, params, grads, slot0, slot1, lr, step):
    from tensorflow_end2end_speech_recognition_amd._lib import OPTIMIZER_IDS
    name = [k for k, v in OPTIMIZER_IDS.items() if v == opt_id][0]
    z = np.zeros(params.numel(), dtype=np.float64)
    s0 = slot0.double().numpy() if slot0 is not None else z
    s1 = slot1.double().numpy() if slot1 is not None else z
    p, s0, s1 = oopt.step(name, params.double().numpy(), grads.double().numpy(), s0, s1, lr, step)
    params.copy_(torch.from_numpy(np.asarray(p)).float())
    if slot0 is not None:
        slot0.copy_(torch.from_numpy(np.asarray(s0)).float())
    if slot1 is not None:
        slot1.copy_(torch.from_numpy(np.asarray(s1)).float())


def _scale_(x, s):
    x.mul_(s)
    return x


# ---------------------------------------------------------------- attention decoder stand-ins
def _lstm_cell_fwd(pre, c_prev, h_prev, peep, live, forget_bias=1.0, cell_clip=0.0, out_mask=None, want_cell_out=False,
                   h_also=None, cell_out_also=None):
    """csrc/attention.hip cell_fwd_kernel: gate-major columns i, ci, f, o; finished rows keep their state."""
    B, U4 = pre.shape
    U = U4 // 4
    p, cp = pre.double(), c_prev.double()
    z = torch.zeros(U, dtype=F64)
    wci, wcf, wco = (peep.double().view(3, U) if peep is not None else (z, z, z))
    i = torch.sigmoid(p[:, :U] + wci * cp)
    g = torch.tanh(p[:, U:2 * U])
    f = torch.sigmoid(p[:, 2 * U:3 * U] + forget_bias + wcf * cp)
    cn = g * i + cp * f
    if cell_clip and cell_clip > 0:
        cn = cn.clamp(-cell_clip, cell_clip)
    o = torch.sigmoid(p[:, 3 * U:] + wco * cn)
    hn = torch.tanh(cn) * o
    lv = (live.double() > 0).view(B, 1)
    gates = torch.cat([i, g, f, o], 1).float()
    h_out = torch.where(lv, hn, h_prev.double()).float()
    cell_out = (hn * out_mask.double()).float() if out_mask is not None else hn.float()
    if h_also is not None:
        h_also.copy_(h_out)
    if cell_out_also is not None:
        cell_out_also.copy_(cell_out)
    res = (gates, cn.float(), torch.where(lv, cn, cp).float(), h_out, hn.float())
    return res + (cell_out,) if want_cell_out else res


def _into(out, val):
    """Write a stand-in's result into the caller's buffer when one is given (the kernels' out-parameter form)."""
    if out is None or val is None:
        return val
    out.copy_(val.reshape(out.shape))
    return out


def _lstm_cell_bwd(dh_use, dc_next, dh_next, gates, c_raw, c_prev, peep, live, want_dpeep=True, dpre_out=None,
                   dpeep_out=None, cell_clip=0.0):
    B, U = dh_use.shape
    gt = gates.double()
    i, g, f, o = gt[:, :U], gt[:, U:2 * U], gt[:, 2 * U:3 * U], gt[:, 3 * U:]
    z = torch.zeros(U, dtype=F64)
    wci, wcf, wco = (peep.double().view(3, U) if peep is not None else (z, z, z))
    c, cp = c_raw.double(), c_prev.double()
    lv = (live.double() > 0).view(B, 1)
    dh = dh_use.double() + dh_next.double()
    tc = torch.tanh(c)
    d_o = dh * tc * o * (1 - o)
    dc = dc_next.double() + dh * o * (1 - tc * tc) + d_o * wco          # the clip is straight-through (LSTMBlockCell) ...
    if cell_clip and cell_clip > 0:                                       # ... or blocks a clamped state (LSTMCell)
        dc = torch.where(c.abs() >= cell_clip, torch.zeros_like(dc), dc)
    d_g = dc * i * (1 - g * g)
    d_i = dc * g * i * (1 - i)
    d_f = dc * cp * f * (1 - f)
    zero = torch.zeros_like(dc)
    dpre = torch.where(lv, torch.cat([d_i, d_g, d_f, d_o], 1), torch.zeros(B, 4 * U, dtype=F64))
    dc_prev = torch.where(lv, dc * f + d_i * wci + d_f * wcf, dc_next.double())
    dh_carry = torch.where(lv, zero, dh_next.double())
    dpeep = None
    if want_dpeep:
        dpeep = torch.stack([torch.where(lv, d_i * cp, zero), torch.where(lv, d_f * cp, zero),
                             torch.where(lv, d_o * c, zero)], 1).float()
    return _into(dpre_out, dpre.float()), dc_prev.float(), dh_carry.float(), _into(dpeep_out, dpeep)


def _att_energy_fwd(keys, qz, v, T, mode):
    """keys [T,B,A] or None, qz [B,A] -> energy [B,T]."""
    q = qz.double().unsqueeze(0)                                           # [1,B,A]
    if mode == 0:
        zz = (keys.double() if keys is not None else 0.0) + q
        if keys is None:
            zz = zz.expand(T, *zz.shape[1:])
        e = (v.double() * torch.tanh(zz)).sum(2)
    else:
        e = (keys.double() * q).sum(2)
    return e.t().contiguous().float()


def _att_energy_bwd(denergy, keys, qz, v, mode, dkeys=None, want_dv=True, dqz_out=None, dv_out=None):
    B, A = qz.shape
    T = denergy.shape[1]
    de = denergy.double().t().unsqueeze(2)                                 # [T,B,1]
    q = qz.double().unsqueeze(0)
    dv = None
    if mode == 0:
        zz = (keys.double() if keys is not None else 0.0) + q
        if keys is None:
            zz = zz.expand(T, B, A)
        th = torch.tanh(zz)
        dz = de * v.double() * (1 - th * th)
        if dkeys is not None:
            dkeys += dz.float()
        dqz = dz.sum(0)
        if want_dv:
            dv = (de * th).sum(0).float()
    else:
        if dkeys is not None:
            dkeys += (de * q).float()
        dqz = (de * keys.double()).sum(0)
        if want_dv:
            dv = torch.zeros(B, A)
    return _into(dqz_out, dqz.float()), _into(dv_out, dv)


_F32_LOWEST = float(np.finfo(np.float32).min)


def _att_softmax_ctx_fwd(energy, seq_len, sharpening, enc, alpha_out=None, sigmoid_norm=None, ctx_also=()):
    B, T = energy.shape
    mask = torch.arange(T).unsqueeze(0) < seq_len.long().clamp(0, T).unsqueeze(1)
    e = torch.where(mask, energy.double(), torch.full_like(energy, _F32_LOWEST, dtype=F64)) * sharpening
    e = e.clamp(min=_F32_LOWEST)
    if sigmoid_norm is not None:
        sg = torch.sigmoid(e) * mask
        tot = sg.sum(1, keepdim=True)
        sigmoid_norm.copy_(tot.view(-1).float())
        alpha = torch.where(tot > 0, sg / tot.clamp(min=1e-300), torch.full_like(sg, 1.0 / T))
    else:
        alpha = torch.softmax(e, 1)
    ctx = torch.einsum('bt,tbe->be', alpha * (mask | (seq_len.long() == 0).unsqueeze(1)), enc.double())
    if alpha_out is not None:
        alpha_out.copy_(alpha.float())
        alpha = alpha_out
    else:
        alpha = alpha.float()
    for t in ctx_also:
        if t is not None:
            t.copy_(ctx.float())
    return alpha, ctx.float()


def _loc_energy(alpha_prev, filt, wfil, keys, qz, v, T):
    """torch statement of asr_att_loc_energy_fwd (all float64, differentiable)."""
    taps = filt.shape[0]
    before = (taps - 1) // 2
    a = torch.nn.functional.pad(alpha_prev.unsqueeze(1), (before, taps - 1 - before))
    f = torch.nn.functional.conv1d(a, filt.reshape(taps, 1, -1).permute(2, 1, 0)).transpose(1, 2)    # [B,T,10]
    z = qz.unsqueeze(1) + f @ wfil
    if keys is not None:
        z = z + keys.transpose(0, 1)
    return (v * torch.tanh(z)).sum(2)


def _att_loc_energy_fwd(alpha_prev, filt, wfil, keys, qz, v, T):
    return _loc_energy(alpha_prev.double(), filt.double(), wfil.double(), None if keys is None else keys.double(),
                       qz.double(), v.double(), T).float()


def _att_loc_energy_bwd(denergy, alpha_prev, filt, wfil, keys, qz, v, dwfil_rows, dfilt_rows, accumulate, dkeys=None,
                        dqz_out=None, dv_out=None):
    B, A = qz.shape
    T = denergy.shape[1]
    dqz, dv, dap = torch.zeros(B, A), torch.zeros(B, A), torch.zeros(B, T)
    for b in range(B):     # per utterance, as the device op reports its row gradients
        leaves = [t.double().clone().requires_grad_(True) for t in
                  (alpha_prev[b:b + 1], filt, wfil, qz[b:b + 1], v)]
        k = None if keys is None else keys[:, b:b + 1].double().clone().requires_grad_(True)
        e = _loc_energy(leaves[0], leaves[1], leaves[2], k, leaves[3], leaves[4], T)
        (e * denergy[b:b + 1].double()).sum().backward()
        dap[b] = leaves[0].grad[0].float()
        dfr, dwr = leaves[1].grad.reshape(dfilt_rows[b].shape).float(), leaves[2].grad.float()
        dfilt_rows[b] = dfilt_rows[b] + dfr if accumulate else dfr
        dwfil_rows[b] = dwfil_rows[b] + dwr if accumulate else dwr
        dqz[b] = leaves[3].grad[0].float()
        dv[b] = leaves[4].grad.float()
        if dkeys is not None and k is not None:
            dkeys[:, b:b + 1] += k.grad.float()
    return _into(dqz_out, dqz), _into(dv_out, dv), dap


def _att_softmax_ctx_bwd(dctx, alpha, seq_len, sharpening, enc, denc=None, sigmoid_norm=None, dalpha_extra=None):
    B, T = alpha.shape
    mask = (torch.arange(T).unsqueeze(0) < seq_len.long().clamp(0, T).unsqueeze(1)).double()
    a = alpha.double()
    da = torch.einsum('be,tbe->bt', dctx.double(), enc.double()) * mask
    if dalpha_extra is not None:
        da = da + dalpha_extra.double() * mask
    dot = (a * da * mask).sum(1, keepdim=True)
    de = sharpening * a * (da - dot) * mask
    if sigmoid_norm is not None:
        de = de * (1 - a * sigmoid_norm.double().view(B, 1))
    if denc is not None:
        denc += torch.einsum('bt,be->tbe', a * mask, dctx.double()).float()
    return de.float()



def _att_decoder_fwd(a):
    """asr_att_decoder_fwd: the To steps, each the sequence of stand-ins the native loop issues as kernels."""
    To, U, Em, E2, T = a['To'], a['U'], a['Em'], a['E2'], a['T']
    dec_in, av_in, c_all, h_all = a['dec_in'], a['av_in'], a['c_all'], a['h_all']
    for k in range(To):
        pre = _gemm(dec_in[k], a['W_cell'], bias=a['b_cell'])
        nxt = dec_in[k + 1] if k + 1 < To else None
        dmask = a['dmask'][k] if a.get('dmask') is not None else None
        gates, c_raw, c_new, h_new, _, cell_out = _lstm_cell_fwd(
            pre, c_all[k], h_all[k], a.get('peep'), a['live'][k], a['forget_bias'], a['cell_clip'], out_mask=dmask,
            want_cell_out=True, h_also=nxt[:, Em + E2:] if nxt is not None else None, cell_out_also=av_in[k, :, :U])
        a['gates_all'][k].copy_(gates)
        a['craw_all'][k].copy_(c_raw)
        c_all[k + 1].copy_(c_new)
        h_all[k + 1].copy_(h_new)
        qz = _gemm(cell_out, a['W_q'], bias=a.get('b_q')) if a['has_query_fc'] else cell_out
        a['qz_all'][k].copy_(qz)
        if a['carry_alpha']:
            energy = _att_loc_energy_fwd(a['alpha_all'][k - 1] if k > 0 else a['alpha_zero'], a['filt'], a['wfil'],
                                         a.get('keys'), qz, a['v'], T)
        else:
            energy = _att_energy_fwd(a.get('keys'), qz, a.get('v'), T, a['att_mode'])
        sn = a.get('snorm_all')
        _att_softmax_ctx_fwd(energy, a['seq_len'], a['sharpening'], a['enc'], alpha_out=a['alpha_all'][k],
                             sigmoid_norm=sn[k] if sn is not None else None,
                             ctx_also=(av_in[k, :, U:], nxt[:, Em:Em + E2] if nxt is not None else None))


def _att_decoder_infer(a, W_av, W_out, b_out, embedding, eos, n_live, check_every=8):
    """asr_att_decoder_infer: every step = the forward step above (saved activations reused: row 0), attentional vector,
    output layer, the selection (argmax, emitted id, finished flag, next input row)."""
    To, B, U, Em, E2, T = a['To'], a['B'], a['U'], a['Em'], a['E2'], a['T']
    dec_in, av_in, c_all, h_all, live = a['dec_in'], a['av_in'], a['c_all'], a['h_all'], a['live']
    C2 = W_out.shape[1]
    out = dict(ids=torch.zeros((To, B), dtype=torch.int32), logits=torch.zeros((To, B, C2)), av=torch.zeros((To, B, U)),
               live=live, live_count=torch.zeros((To + 1,), dtype=torch.int32))
    out['live_count'][0] = int(n_live)
    issued = To
    for k in range(To):
        if check_every and k > 0 and k % check_every == 0 and int(out['live_count'][k]) == 0:
            issued = k           # (the device form may issue a few steps more; the result is trimmed the same way)
            break
        pre = _gemm(dec_in[k], a['W_cell'], bias=a['b_cell'])
        nxt = dec_in[k + 1] if k + 1 < To else None
        gates, c_raw, c_new, h_new, _, cell_out = _lstm_cell_fwd(
            pre, c_all[k], h_all[k], a.get('peep'), live[k], a['forget_bias'], a['cell_clip'], out_mask=None,
            want_cell_out=True, h_also=nxt[:, Em + E2:] if nxt is not None else None, cell_out_also=av_in[k, :, :U])
        c_all[k + 1].copy_(c_new)
        h_all[k + 1].copy_(h_new)
        qz = _gemm(cell_out, a['W_q'], bias=a.get('b_q')) if a['has_query_fc'] else cell_out
        if a['carry_alpha']:
            energy = _att_loc_energy_fwd(a['alpha_all'][k - 1] if k > 0 else a['alpha_zero'], a['filt'], a['wfil'],
                                         a.get('keys'), qz, a['v'], T)
        else:
            energy = _att_energy_fwd(a.get('keys'), qz, a.get('v'), T, a['att_mode'])
        sn = a.get('snorm_all')
        _att_softmax_ctx_fwd(energy, a['seq_len'], a['sharpening'], a['enc'], alpha_out=a['alpha_all'][k],
                             sigmoid_norm=sn[k] if sn is not None else None,
                             ctx_also=(av_in[k, :, U:], nxt[:, Em:Em + E2] if nxt is not None else None))
        av = _tanh_fwd(_gemm(av_in[k], W_av))
        lg = _gemm(av, W_out, bias=b_out)
        out['av'][k].copy_(av)
        out['logits'][k].copy_(lg)
        ids = _argmax_rows(lg)
        lv = live[k]
        out['ids'][k].copy_((ids.float() * lv).to(torch.int32))
        live[k + 1].copy_(lv * (ids != int(eos)).float())
        out['live_count'][k + 1] = int(live[k + 1].sum())
        if nxt is not None:
            nxt[:, :Em].copy_(_embedding_gather(embedding, ids))
            nxt[:, Em:Em + E2].mul_(lv.unsqueeze(1))
    out['steps_issued'] = issued
    return out


def _att_decoder_bwd(a):
    """asr_att_decoder_bwd, step for step."""
    To, B, U, Em, E2 = a['To'], a['B'], a['U'], a['Em'], a['E2']
    dc_next, dh_next = torch.zeros(B, U), torch.zeros(B, U)
    dctx_in = torch.zeros(B, E2)
    dalpha_next = None
    sn = a.get('snorm_all')
    for k in range(To - 1, -1, -1):
        dctx = a['dctx_all'][k]
        torch.add(a['dav_ctx'][k], dctx_in, out=dctx)
        denergy = _att_softmax_ctx_bwd(dctx, a['alpha_all'][k], a['seq_len'], a['sharpening'], a['enc'], None,
                                       sigmoid_norm=sn[k] if sn is not None else None, dalpha_extra=dalpha_next)
        dv_out = a['dv_all'][k] if a.get('dv_all') is not None else None
        if a['carry_alpha']:
            dqz, _, dalpha_next = _att_loc_energy_bwd(
                denergy, a['alpha_all'][k - 1] if k > 0 else a['alpha_zero'], a['filt'], a['wfil'], a.get('keys'),
                a['qz_all'][k], a['v'], a['dwfil_rows'], a['dfilt_rows'], accumulate=(k != To - 1),
                dkeys=a.get('dkeys'), dqz_out=a['dqz_all'][k], dv_out=dv_out)
        else:
            dqz, _ = _att_energy_bwd(denergy, a.get('keys'), a['qz_all'][k], a.get('v'), a['att_mode'],
                                     dkeys=a.get('dkeys'), want_dv=a['att_mode'] == 0, dqz_out=a['dqz_all'][k],
                                     dv_out=dv_out)
        dcell = a['dav_cell'][k]
        if a['has_query_fc']:
            _gemm(dqz, a['W_q'], transB=True, out=dcell, accumulate=True)
        else:
            dcell += dqz
        if a.get('dmask') is not None:
            dcell.copy_(_apply_mask(dcell, a['dmask'][k]))
        dp = a.get('dpeep_all')
        dpre, dc_prev, dh_carry, _ = _lstm_cell_bwd(
            dcell, dc_next, dh_next, a['gates_all'][k], a['craw_all'][k], a['c_all'][k], a.get('peep'), a['live'][k],
            want_dpeep=dp is not None, dpre_out=a['dpre_all'][k], dpeep_out=dp[k] if dp is not None else None)
        d_in = _gemm(dpre, a['W_cell'], transB=True, out=a['d_in_all'][k])
        dctx_in = d_in[:, Em:Em + E2]
        dh_next = dh_carry + d_in[:, Em + E2:]
        dc_next = dc_prev
    a['dc0'].copy_(dc_next)
    a['dh0'].copy_(dh_next)


def _tanh_fwd(x):
    return torch.tanh(x.double()).float()


def _tanh_bwd(dy, y):
    return (dy.double() * (1 - y.double() ** 2)).float()


def _embedding_gather(W, ids):
    return W[ids.long()].contiguous()


def _embedding_scatter(dout, ids, vocab, out):
    out.zero_()
    out.index_add_(0, ids.long().view(-1), dout.reshape(-1, dout.shape[-1]))
    return out


def _seq_xent(logits2d, targets, weights, eps, dscale, want_grad=True):
    x = logits2d.double() + eps
    lse = torch.logsumexp(x, 1)
    w = weights.double()
    tg = targets.long()
    row_loss = w * (lse - x.gather(1, tg.view(-1, 1)).view(-1))
    dl = None
    if want_grad:
        dl = torch.exp(x - lse.view(-1, 1))
        dl[torch.arange(x.shape[0]), tg] -= 1.0
        dl = (dl * (w * dscale).view(-1, 1)).float()
    return row_loss.float(), dl


def _argmax_rows(x2d):
    return torch.argmax(x2d, 1).to(torch.int32)


# ---------------------------------------------------------------- VGG front-end stand-ins (im2col form)
def _im2col3x3(x_nhwc, ldp=None, out=None):
    """patches[(n*H+h)*W+w, tap*Cin+ci], tap = kh*3+kw, zero padding (SAME)."""
    N, H, W, Cin = x_nhwc.shape
    ldp = ldp or 9 * Cin
    xp = torch.nn.functional.pad(x_nhwc.permute(0, 3, 1, 2), (1, 1, 1, 1))          # [N,Cin,H+2,W+2]
    cols = [xp[:, :, kh:kh + H, kw:kw + W].permute(0, 2, 3, 1).reshape(N * H * W, Cin)
            for kh in range(3) for kw in range(3)]
    pat = torch.zeros((N * H * W, ldp), dtype=x_nhwc.dtype)
    pat[:, :9 * Cin] = torch.cat(cols, 1)
    return pat


def _col2im3x3(dpatches, N, H, W, Cin):
    dp = dpatches[:, :9 * Cin].reshape(N, H, W, 9, Cin).double()
    din = torch.zeros((N, H + 2, W + 2, Cin), dtype=F64)
    for kh in range(3):
        for kw in range(3):
            din[:, kh:kh + H, kw:kw + W] += dp[:, :, :, kh * 3 + kw]
    return din[:, 1:H + 1, 1:W + 1].float().contiguous()


def _conv_geo(H, W, kh, kw, sh, sw):
    Ho, Wo = (H + sh - 1) // sh, (W + sw - 1) // sw
    ph, pw = max((Ho - 1) * sh + kh - H, 0), max((Wo - 1) * sw + kw - W, 0)
    return Ho, Wo, ph // 2, ph - ph // 2, pw // 2, pw - pw // 2


def _im2col(x_nhwc, kh, kw, sh, sw, ldp=None):
    N, H, W, Cin = x_nhwc.shape
    Ho, Wo, pt, pb, pl, pr = _conv_geo(H, W, kh, kw, sh, sw)
    xp = torch.nn.functional.pad(x_nhwc.permute(0, 3, 1, 2).double(), (pl, pr, pt, pb))
    cols = torch.nn.functional.unfold(xp, (kh, kw), stride=(sh, sw))         # [N, Cin*kh*kw, Ho*Wo], (ci, ky, kx) major
    cols = cols.view(N, Cin, kh * kw, Ho * Wo).permute(0, 3, 2, 1).reshape(N * Ho * Wo, kh * kw * Cin)
    K = kh * kw * Cin
    out = torch.zeros((N * Ho * Wo, ldp or K))
    out[:, :K] = cols.float()
    return out


def _col2im(dpatches, N, H, W, Cin, kh, kw, sh, sw):
    Ho, Wo, pt, pb, pl, pr = _conv_geo(H, W, kh, kw, sh, sw)
    K = kh * kw * Cin
    cols = dpatches[:, :K].double().reshape(N, Ho * Wo, kh * kw, Cin).permute(0, 3, 2, 1).reshape(N, Cin * kh * kw, Ho * Wo)
    xp = torch.nn.functional.fold(cols, (H + pt + pb, W + pl + pr), (kh, kw), stride=(sh, sw))
    return xp[:, :, pt:pt + H, pl:pl + W].permute(0, 2, 3, 1).float().contiguous()


def _maxpool2x2_fwd(x_nhwc):
    """2x2 stride 2 SAME: odd edges are padded at the bottom / right (never selected)."""
    N, H, W, Cc = x_nhwc.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    xp = torch.full((N, 2 * Ho, 2 * Wo, Cc), float('-inf'), dtype=x_nhwc.dtype)
    xp[:, :H, :W] = x_nhwc
    win = xp.view(N, Ho, 2, Wo, 2, Cc).permute(0, 1, 3, 5, 2, 4).reshape(N, Ho, Wo, Cc, 4)
    out, arg = win.max(dim=4)
    return out.contiguous(), arg.to(torch.uint8)


def _maxpool2x2_bwd(dout, arg, H, W):
    N, Ho, Wo, Cc = dout.shape
    g = torch.zeros((N, Ho, Wo, Cc, 4))
    g.scatter_(4, arg.long().unsqueeze(4), dout.unsqueeze(4))
    g = g.view(N, Ho, Wo, Cc, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2 * Ho, 2 * Wo, Cc)
    return g[:, :H, :W].contiguous()


def _maxpool2x2_relu_bwd(dout, arg, act, drop=None, pooled=None, hw=None):
    if pooled is not None:      # the pooled activation after its dropout is the mask source: > 0 where active and kept
        H, W = hw
        d = dout * (pooled > 0).to(dout.dtype) * (1.0 / drop[0] if drop is not None else 1.0)
        return _maxpool2x2_bwd(d, arg, H, W).to(pooled.dtype)
    d = dout if drop is None else _dropout_apply(dout, *drop)
    N, H, W, _ = act.shape
    return _relu_bwd(_maxpool2x2_bwd(d, arg, H, W), act).to(act.dtype)


# ---- the bf16 MFMA path of the VGG front-end (models/encoders/core/vgg_blstm.py: implicit-GEMM 3x3 convolutions, the
# few-channel first layer, dropout in the producing kernels' epilogues).  Values are rounded where the device rounds
# (stored activations / pre-activation gradients are bf16, data gradients fp32); products accumulate in fp64.
def _bf(t):
    return t.to(torch.bfloat16)


def _conv3x3_prep_weights(w_hwio):
    """(wf [Cout, 9 Cin], wb [Cin, 9 Cout]) bf16: forward image and flipped-tap image (asr_conv3x3_prep_weights)."""
    _, _, Cin, Cout = w_hwio.shape
    wq = _bf(w_hwio)
    wf = wq.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    wb = wq.flip(0, 1).permute(2, 0, 1, 3).reshape(Cin, 9 * Cout).contiguous()
    return wf, wb


def _conv_nhwc(x, w_oihw, bias=None):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w_oihw.double(), None if bias is None else bias.double(),
                                   padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


def _conv3x3_fwd(x, wt_fwd, bias, relu=True):
    Cout, Cin = wt_fwd.shape[0], x.shape[3]
    y = _conv_nhwc(x, wt_fwd.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bias)
    return _bf(torch.relu(y) if relu else y)


def _conv3x3_fwd_drop(x, wt_fwd, bias, drop):
    return _bf(_dropout_apply(_conv3x3_fwd(x, wt_fwd, bias, True).float(), *drop))


def _conv3x3_bwd_data(dy, wt_bwd):
    Cin, Cout = wt_bwd.shape[0], dy.shape[3]
    return _conv_nhwc(dy, wt_bwd.view(Cin, 3, 3, Cout).permute(0, 3, 1, 2)).float()      # the image holds the flipped taps


def _conv3x3_bwd_data_relu(dy, wt_bwd, act_below, drop=None, dropped=False):
    dx = _conv3x3_bwd_data(dy, wt_bwd)
    if drop is None:
        return _bf(dx * (act_below > 0))
    if dropped:                 # act_below is the DROPPED activation: > 0 where active and kept
        return _bf(dx * (act_below > 0) * (1.0 / drop[0]))
    return _bf(_relu_bwd(dx, act_below, drop=drop))


def _conv_wgrad(x, dy):
    """[9 Cin, Cout] fp64: dW[tap Cin + ci, co] = sum_p x[p + s_tap, ci] dy[p, co]."""
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    rows = []
    for ty in range(3):
        for tx in range(3):
            rows.append(torch.einsum('nhwc,nhwo->co', xp[:, ty:ty + H, tx:tx + W], dy.double()))
    return torch.cat(rows, 0)


def _conv3x3_bwd_weight(x, dy, dw, accumulate=False):
    g = _conv_wgrad(x, dy).float()
    dw.copy_(dw + g if accumulate else g)
    return dw


def _conv3x3_smallc_fwd(x, w2d, bias, relu=True):
    Cin, Cout = x.shape[3], w2d.shape[1]
    y = _conv_nhwc(x, w2d.view(3, 3, Cin, Cout).permute(3, 2, 0, 1), bias)
    return _bf(torch.relu(y) if relu else y)


def _conv3x3_smallc_fwd_drop(x, w2d, bias, drop):
    return _bf(_dropout_apply(_conv3x3_smallc_fwd(x, w2d, bias, True).float(), *drop))


def _conv3x3_smallc_bwd_weight(x, dpre, dw):
    dw.copy_(_conv_wgrad(x, dpre).float())
    return dw


def _conv3x3_bwd_weight_bias(x, dy, dw, dbias):
    _conv3x3_bwd_weight(x, dy, dw)
    dbias.copy_(dy.double().sum(dim=(0, 1, 2)).float())
    return dw, dbias


def _conv3x3_smallc_bwd_weight_bias(x, dpre, dw, dbias):
    _conv3x3_smallc_bwd_weight(x, dpre, dw)
    dbias.copy_(dpre.double().sum(dim=(0, 1, 2)).float())
    return dw, dbias


def _maxpool2x2_fwd_drop(x, drop):
    out, arg = _maxpool2x2_fwd(x)
    return _dropout_apply(out.float(), *drop).to(x.dtype), arg


def _relu_bwd_scaled(dout, out_dropped, keep):
    return (dout * (out_dropped > 0) * (1.0 / keep)).to(out_dropped.dtype)


STAND_INS = dict(
    side_lane=_NullLane, join_side=lambda device: None, keep_on_lane=lambda device, lane, tensors: None, stream_event=lambda: None, set_side_gemm_workgroups=lambda device, n: None, wait_event=lambda ev: None,
    gru_fwd=_gru_fwd, gru_bwd=_gru_bwd, lstm_prep_layer=_lstm_prep_layer, lstm_grad_finish=_lstm_grad_finish, bt_to_tb=_bt_to_tb, transpose2d=_transpose2d,
    cast_from_f32=_cast_from_f32, cast_to_f32=_cast_to_f32, apply_mask=_apply_mask, dropout_mask=_dropout_mask,
    dropout_apply=_dropout_apply, maxpool2x2_relu_bwd=_maxpool2x2_relu_bwd, touch=lambda t: None,
    colsum=_colsum, gemm=_gemm, relu_bwd=_relu_bwd, lstm_prep_weights=_lstm_prep_weights,
    gate_deinterleave=_gate_deinterleave, lstm_fwd=_lstm_fwd, lstm_bwd=_lstm_bwd, lstm_units_supported=lambda H: True, ctc_loss=_ctc_loss,
    ctc_greedy_decode=_ctc_greedy_decode, softmax_rows=_softmax_rows, clip_by_norm_multi=_clip_by_norm_multi,
    weight_decay=_weight_decay, optimizer_step=_optimizer_step, scale_=_scale_,
    lstm_cell_fwd=_lstm_cell_fwd, lstm_cell_bwd=_lstm_cell_bwd, att_energy_fwd=_att_energy_fwd,
    att_energy_bwd=_att_energy_bwd, att_softmax_ctx_fwd=_att_softmax_ctx_fwd,
    att_softmax_ctx_bwd=_att_softmax_ctx_bwd, tanh_fwd=_tanh_fwd, tanh_bwd=_tanh_bwd,
    att_loc_energy_fwd=_att_loc_energy_fwd, att_loc_energy_bwd=_att_loc_energy_bwd,
    embedding_gather=_embedding_gather, embedding_scatter=_embedding_scatter, seq_xent=_seq_xent,
    argmax_rows=_argmax_rows, im2col3x3=_im2col3x3, col2im3x3=_col2im3x3, maxpool2x2_fwd=_maxpool2x2_fwd,
    maxpool2x2_bwd=_maxpool2x2_bwd, im2col=_im2col, col2im=_col2im, att_decoder_fwd=_att_decoder_fwd,
    att_decoder_bwd=_att_decoder_bwd, att_decoder_infer=_att_decoder_infer,
    conv3x3_prep_weights=_conv3x3_prep_weights, conv3x3_fwd=_conv3x3_fwd, conv3x3_fwd_drop=_conv3x3_fwd_drop,
    conv3x3_bwd_data=_conv3x3_bwd_data, conv3x3_bwd_data_relu=_conv3x3_bwd_data_relu, conv3x3_bwd_weight=_conv3x3_bwd_weight,
    conv3x3_smallc_fwd=_conv3x3_smallc_fwd, conv3x3_smallc_fwd_drop=_conv3x3_smallc_fwd_drop,
    conv3x3_smallc_bwd_weight=_conv3x3_smallc_bwd_weight, maxpool2x2_fwd_drop=_maxpool2x2_fwd_drop,
    conv3x3_bwd_weight_bias=_conv3x3_bwd_weight_bias, conv3x3_smallc_bwd_weight_bias=_conv3x3_smallc_bwd_weight_bias,
    relu_bwd_scaled=_relu_bwd_scaled,
)


def install(monkeypatch=None):
    """Swap the kernel front end for the CPU stand-ins for one test (monkeypatch fixture), or for the life of a
    spawned worker process (monkeypatch=None); pinned staging buffers become plain ones."""
    from tensorflow_end2end_speech_recognition_amd import ops
    for name, fn in STAND_INS.items():
        assert hasattr(ops, name), name
        if monkeypatch is not None:
            monkeypatch.setattr(ops, name, fn)
        else:
            setattr(ops, name, fn)
    pin = lambda self, *a, **k: self
    if monkeypatch is not None:
        monkeypatch.setattr(torch.Tensor, 'pin_memory', pin)
    else:
        torch.Tensor.pin_memory = pin
    return ops

"""VGG + (bidirectional) LSTM encoder -- mirror of models/encoders/core/vgg_blstm.py:17-220
(class VGGBLSTMEncoder; vgg_lstm.py is the same front-end over the unidirectional stack).

__call__(inputs [B,T,num_channels*(splice*num_stack)*3], inputs_seq_len, keep_prob, is_training):
reshape to [B*T, num_channels, splice*num_stack, 3] (:108-110), VGG1: conv3x3(3->64)+relu+dropout,
conv3x3(64->64)+relu, max_pool 2x2 SAME, dropout (:113-134); VGG2: 64->128, 128->128, pool,
dropouts (:136-157); flatten; bridge FC -> 256 relu + dropout (:165-174); then the same LSTM stack
as BLSTMEncoder (:182-218).  Variables: VGG{1,2}/conv{1,2}/{weight,bias} (tf.Variable in
cnn_util.py:66-69), bridge/{weights,biases}.

Execution: bf16 layers with Cin % 64 == 0 (3 of the 4 convolutions, 98 % of the FLOPs) run as
implicit GEMMs (asr_conv3x3_fwd / _bwd_data / _bwd_weight: no patch matrix); the 3-channel first
layer and the fp32 parity path use asr_im2col3x3 + MFMA GEMM with fused bias+ReLU, chunked over
frames (CHUNK_FRAMES) so the patch matrix is a bounded scratch; backward recomputes the patches.
"""
import os

import os as _os

import numpy as np
import torch

from .... import ops
from ...._lib import ASR_BF16, ASR_F32
from ....utils.parameter import ParamStore
from .blstm import BLSTMEncoder
from .lstm import LSTMEncoder

CONVS = [('VGG1/conv1', 3, 64), ('VGG1/conv2', 64, 64), ('VGG2/conv1', 64, 128), ('VGG2/conv2', 128, 128)]
CHUNK_FRAMES = 4096
VGG_FWD_CHUNK_MIN = 2048     # images per run below which the single pass is kept
VGG_FWD_CHUNKS = int(_os.environ.get('ASR_VGG_FWD_CHUNKS', '2'))        # runs of images the fused bf16 forward goes through on separate lanes
VGG_WGRAD_SIDE = _os.environ.get('ASR_VGG_WGRAD_SIDE', '1') != '0'   # weight gradients of the implicit-GEMM layers on side lane 1


def _trunc_normal(rng, std, shape):
    x = rng.normal(0.0, std, size=shape)
    bad = np.abs(x) > 2 * std
    while bad.any():
        x[bad] = rng.normal(0.0, std, size=int(bad.sum()))
        bad = np.abs(x) > 2 * std
    return x


class _VGGFrontEnd(object):
    """conv/pool/bridge part; owns no LSTM."""

    def __init__(self, input_size, splice, num_stack, parameter_init, dtype):
        assert input_size % 3 == 0
        self.F = input_size // 3
        self.W = splice * num_stack
        self.parameter_init = parameter_init
        self.dtype = dtype
        self.out_dim = 256
        self.prefix = ''
        # dropout in the epilogue of the producing kernels (forward()); ASR_VGG_FUSED_DROP=0 / .fused_drop = False: every
        # dropout as its own pass over the stored activation (A/B and the equality test)
        self.fused_drop = os.environ.get('ASR_VGG_FUSED_DROP', '1') != '0'

    def build(self, store, rng, prefix=''):
        self.store, self.prefix = store, prefix
        for name, cin, cout in CONVS:
            store.declare(prefix + name + '/weight', (3, 3, cin, cout), _trunc_normal(rng, self.parameter_init, (3, 3, cin, cout)))
            store.declare(prefix + name + '/bias', (cout,), np.zeros(cout))
        H2, W2 = (self.F + 1) // 2, (self.W + 1) // 2
        H4, W4 = (H2 + 1) // 2, (W2 + 1) // 2
        self.flat = H4 * W4 * 128
        store.declare(prefix + 'bridge/weights', (self.flat, 256), _trunc_normal(rng, self.parameter_init, (self.flat, 256)))
        store.declare(prefix + 'bridge/biases', (256,), np.zeros(256))

    # one conv layer on a chunk: x [n,H,W,Cin] (operand dtype) -> relu(conv + b) same dtype
    def _conv_fwd(self, x, name, cin, cout, sh):
        n, H, W, _ = x.shape
        ldp = (9 * cin + 7) // 8 * 8                     # 16-B aligned rows for the GEMM loader
        patches = ops.im2col3x3(x, ldp)
        w2d = sh[self.prefix + name + '/weight'].view(9 * cin, cout)
        return ops.gemm(patches[:, :9 * cin], w2d, bias=self.store[self.prefix + name + '/bias'], relu=True)

    def _pack_index(self, lens, T, dev):
        """(valid-frame gather index, inverse index) on the device for the lengths `lens`: built once per distinct batch
        geometry (an epoch on bucketed batches repeats them; bench loops reuse one) and uploaded through pinned memory,
        so the step never waits on it."""
        key = (lens.tobytes(), int(T), str(dev))
        cache = self.__dict__.setdefault('_pack_cache', {})
        hit = cache.get(key)
        if hit is None:
            B = len(lens)
            t = np.arange(T, dtype=np.int64)[None, :]
            mask = t < lens[:, None]
            valid = np.flatnonzero(mask.reshape(-1)).astype(np.int32)      # b * T + t of every valid frame, row-major
            inv = np.full(B * T, len(valid), dtype=np.int32)                # padded frames -> the extra zero row
            inv[valid] = np.arange(len(valid), dtype=np.int32)
            if len(cache) >= 16:
                cache.clear()
            hit = cache[key] = (ops.to_device(valid, torch.int32, dev), ops.to_device(inv, torch.int32, dev))
        return hit

    def forward(self, x_btd, keep_prob, is_training, rng_state=None, seq_len=None):
        """x [B,T,F*W*3] fp32 cuda -> [B,T,256] fp32; keeps what backward needs.
        seq_len (host ints): only the valid frames go through the convolutions -- every frame is an
        independent image and the recurrent stack never reads a padded one (dynamic_rnn(sequence_length)),
        so dropping them changes nothing but the work (44 % of a LibriSpeech-shaped batch is padding)."""
        st = self.store
        sh = st.shadow(self.dtype)
        B, T, Dd = x_btd.shape
        assert Dd == self.F * self.W * 3
        pack = None
        if seq_len is not None:
            lens = np.minimum(np.maximum(np.asarray(seq_len, dtype=np.int64), 0), T)
            if int(lens.sum()) < B * T:
                pack = self._pack_index(lens, T, x_btd.device)
        if pack is not None and pack[0].numel() > 0:
            x_rows = ops.embedding_gather(x_btd.reshape(B * T, Dd), pack[0])
        else:
            pack = None
            x_rows = x_btd.reshape(B * T, Dd)
        N = x_rows.shape[0]
        tdt = torch.bfloat16 if self.dtype == ASR_BF16 else torch.float32
        x0 = ops.cast_from_f32(x_rows.reshape(N, self.F, self.W, 3).contiguous(), self.dtype) \
            if self.dtype == ASR_BF16 else x_rows.reshape(N, self.F, self.W, 3).contiguous()
        drop = is_training and keep_prob < 1.0
        self.ctx = dict(N=N, B=B, T=T, x0=x0, acts=[], masks={}, args=[], pack=pack)
        self._drop_i = 0

        def dropout(t, key):
            if not drop:
                return t
            seed, off = rng_state
            self._drop_i += 1
            # no mask tensor: the mask is formed from (seed, offset) where it is applied, forward and backward
            d = (float(keep_prob), seed + 7, off + (self._drop_i << 32))
            self.ctx['masks'][key] = d
            return ops.dropout_apply(t, *d)
        def descriptor(key):
            """The dropout of tensor `key` as (keep, seed, offset) -- same numbering as dropout() -- for the kernels that
            apply it in their epilogue."""
            seed, off = rng_state
            self._drop_i += 1
            d = (float(keep_prob), seed + 7, off + (self._drop_i << 32))
            self.ctx['masks'][key] = d
            return d
        # Round 4: on the bf16 MFMA path tf.nn.dropout is applied in the epilogue of the kernel that produces the tensor
        # (first-layer and implicit-GEMM convolutions, max-pool): the undropped ReLU output is never written and there is no
        # pass over the gigabyte activations just to mask them (3.6 ms of an 83 ms cfg C step).  Same masks, same values:
        # the dropped tensors are bit-identical to dropout_apply of the separate outputs; the backward takes "active and
        # kept" from the sign of the dropped tensor.
        fused = (drop and self.dtype == ASR_BF16 and self.fused_drop and self._direct(*CONVS[0][1:]) and
                 all(self._implicit(*c[1:]) for c in CONVS[1:]))      # (.fused_drop = False: the separate passes, A/B + tests)
        self.ctx['fused_drop'] = fused
        if fused and VGG_FWD_CHUNKS > 1 and N >= VGG_FWD_CHUNK_MIN * VGG_FWD_CHUNKS:
            # Round 5: the frames are independent images, and the chain alternates kernels bound by different things (the
            # first layer by its Philox rounds, the 64 / 128-channel convolutions by the matrix pipe and the LDS, the pools
            # by HBM).  The images go through in VGG_FWD_CHUNKS contiguous runs, run k on its own lane: two kernels of the
            # SAME stage cannot share a CU (157 KB of LDS each), so the lanes fall one stage apart by themselves and a
            # pool of one run streams under the multiplies of the other.  Same kernels on sub-ranges of the same buffers,
            # the dropout counters shifted by the run's first element: bit-identical to the single pass.
            F_, W_ = self.F, self.W
            H2, W2 = (F_ + 1) // 2, (W_ + 1) // 2
            H4, W4 = (H2 + 1) // 2, (W2 + 1) // 2
            dev = x0.device
            bf = torch.bfloat16
            a1d = torch.empty((N, F_, W_, 64), dtype=bf, device=dev)
            a2 = torch.empty((N, F_, W_, 64), dtype=bf, device=dev)
            p1d = torch.empty((N, H2, W2, 64), dtype=bf, device=dev)
            arg1 = torch.empty((N, H2, W2, 64), dtype=torch.uint8, device=dev)
            a3d = torch.empty((N, H2, W2, 128), dtype=bf, device=dev)
            a4 = torch.empty((N, H2, W2, 128), dtype=bf, device=dev)
            p2d = torch.empty((N, H4, W4, 128), dtype=bf, device=dev)
            arg2 = torch.empty((N, H4, W4, 128), dtype=torch.uint8, device=dev)
            w1 = sh[self.prefix + CONVS[0][0] + '/weight'].view(9 * CONVS[0][1], CONVS[0][2])
            b = [st[self.prefix + c[0] + '/bias'] for c in CONVS]
            wi = [None, self._conv_images(CONVS[1][0])[0], self._conv_images(CONVS[2][0])[0],
                  self._conv_images(CONVS[3][0])[0]]
            d_a1, d_p1, d_a3, d_p2 = descriptor('a1'), descriptor('p1'), descriptor('a3'), descriptor('p2')

            def at(d, first_elem):          # the same dropout stream, entered at element first_elem (a multiple of 4)
                return (d[0], d[1], d[2] + first_elem // 4)

            def chain(c0, c1):
                ops.conv3x3_smallc_fwd_drop(x0[c0:c1], w1, b[0], at(d_a1, c0 * F_ * W_ * 64), out=a1d[c0:c1])
                ops.conv3x3_fwd(a1d[c0:c1], wi[1], b[1], relu=True, out=a2[c0:c1])
                ops.maxpool2x2_fwd_drop(a2[c0:c1], at(d_p1, c0 * H2 * W2 * 64), out=p1d[c0:c1], arg=arg1[c0:c1])
                ops.conv3x3_fwd_drop(p1d[c0:c1], wi[2], b[2], at(d_a3, c0 * H2 * W2 * 128), out=a3d[c0:c1])
                ops.conv3x3_fwd(a3d[c0:c1], wi[3], b[3], relu=True, out=a4[c0:c1])
                ops.maxpool2x2_fwd_drop(a4[c0:c1], at(d_p2, c0 * H4 * W4 * 128), out=p2d[c0:c1], arg=arg2[c0:c1])
            bounds = [N * k // VGG_FWD_CHUNKS for k in range(VGG_FWD_CHUNKS + 1)]
            keep = (x0, a1d, a2, p1d, arg1, a3d, a4, p2d, arg2)
            for k in range(1, VGG_FWD_CHUNKS):
                with ops.side_lane(dev, keep=keep, lane=2 + k):
                    chain(bounds[k], bounds[k + 1])
            chain(bounds[0], bounds[1])
            ops.join_side(dev)
            a1, a3 = a1d, a3d
        elif fused:
            w1 = sh[self.prefix + CONVS[0][0] + '/weight'].view(9 * CONVS[0][1], CONVS[0][2])
            a1 = a1d = ops.conv3x3_smallc_fwd_drop(x0, w1, st[self.prefix + CONVS[0][0] + '/bias'], descriptor('a1'))
            a2 = self._layer(a1d, CONVS[1], sh)
            p1d, arg1 = ops.maxpool2x2_fwd_drop(a2, descriptor('p1'))
            a3 = a3d = ops.conv3x3_fwd_drop(p1d, self._conv_images(CONVS[2][0])[0], st[self.prefix + CONVS[2][0] + '/bias'],
                                            descriptor('a3'))
            a4 = self._layer(a3d, CONVS[3], sh)
            p2d, arg2 = ops.maxpool2x2_fwd_drop(a4, descriptor('p2'))
        else:
            a1 = self._layer(x0, CONVS[0], sh)
            a1d = dropout(a1, 'a1')
            a2 = self._layer(a1d, CONVS[1], sh)
            p1, arg1 = ops.maxpool2x2_fwd(a2)
            p1d = dropout(p1, 'p1')
            a3 = self._layer(p1d, CONVS[2], sh)
            a3d = dropout(a3, 'a3')
            a4 = self._layer(a3d, CONVS[3], sh)
            p2, arg2 = ops.maxpool2x2_fwd(a4)
            p2d = dropout(p2, 'p2')
        flat = p2d.reshape(N, self.flat)
        br = ops.gemm(flat, sh[self.prefix + 'bridge/weights'], bias=st[self.prefix + 'bridge/biases'], relu=True)
        brd = dropout(br, 'br')
        self.ctx.update(a1=a1, a1d=a1d, a2=a2, arg1=arg1, p1d=p1d, a3=a3, a3d=a3d, a4=a4, arg2=arg2, p2d=p2d,
                        flat=flat, br=br)
        out = ops.cast_to_f32(brd) if brd.dtype != torch.float32 else brd
        if pack is not None:   # back to the padded [B,T] grid: padded frames read the appended zero row
            table = torch.cat([out, out.new_zeros(1, 256)], 0)
            out = ops.embedding_gather(table, pack[1])
        return out.view(B, T, 256)

    def _implicit(self, cin, cout):
        import os
        if os.environ.get('ASR_VGG_IMPLICIT', '1') == '0':      # A-B switch: im2col + GEMM everywhere
            return False
        return self.dtype == ASR_BF16 and cin % 64 == 0 and cout % 64 == 0

    def _conv_images(self, name):
        """bf16 weight images of the implicit GEMMs (forward / flipped-tap), built once per step."""
        cache = self.ctx.setdefault('wimg', {})
        if name not in cache:
            cache[name] = ops.conv3x3_prep_weights(self.store[self.prefix + name + '/weight'])
        return cache[name]

    def _direct(self, cin, cout):
        """The few-channel first layer (K = 27) without a patch matrix (asr_conv3x3_smallc_*)."""
        import os
        return self.dtype == ASR_BF16 and 9 * cin <= 32 and cout == 64 and os.environ.get('ASR_VGG_IMPLICIT', '1') != '0'

    def _layer(self, x, conv, sh):
        name, cin, cout = conv
        if self._implicit(cin, cout):
            return ops.conv3x3_fwd(x, self._conv_images(name)[0], self.store[self.prefix + name + '/bias'], relu=True)
        if self._direct(cin, cout):
            return ops.conv3x3_smallc_fwd(x, sh[self.prefix + name + '/weight'].view(9 * cin, cout),
                                          self.store[self.prefix + name + '/bias'], relu=True)
        N, H, W, _ = x.shape
        out = torch.empty((N, H, W, cout), dtype=x.dtype, device=x.device)
        for c0 in range(0, N, CHUNK_FRAMES):
            xc = x[c0:c0 + CHUNK_FRAMES]
            out[c0:c0 + CHUNK_FRAMES] = self._conv_fwd(xc, name, cin, cout, sh).view(xc.shape[0], H, W, cout)
        return out

    # conv backward on the full batch, chunked: dout fp32 [N,H,W,Cout] -> din fp32 [N,H,W,Cin]; fills dW, db
    def _conv_bwd(self, dout, out, mask, x_in, conv, sh, need_dx=True, pooled=None, below=None, dout_is_dpre=False,
                  out_dropped=False):
        """pooled = (argmax, drop or None): dout is the gradient of the POOLED output (before its dropout); un-pooling,
        that dropout and this convolution's ReLU backward run as one kernel (implicit-GEMM layers).
        below = (act, drop or None) of the layer below (implicit-GEMM layers): the data gradient comes back as that
        layer's pre-activation gradient in the operand dtype (its ReLU / dropout backward in the convolution's
        epilogue) and is passed to its _conv_bwd with dout_is_dpre=True.
        Fused-dropout forward (ctx['fused_drop']): pooled = (argmax, drop, pooled activation AFTER its dropout) -- that
        tensor replaces `out` and the mask in the un-pooling pass; below = (DROPPED act, drop, True); out_dropped: `out`
        is the dropped ReLU output, the mask is its sign and the scale 1 / keep."""
        name, cin, cout = conv
        st = self.store
        N, H, W, _ = out.shape
        gw = st.g(self.prefix + name + '/weight').view(9 * cin, cout)
        gb = st.g(self.prefix + name + '/bias')
        if pooled is not None and not (self._implicit(cin, cout) and mask is None and cout % 4 == 0):
            d = dout.contiguous()
            if pooled[1] is not None:
                d = ops.dropout_apply(d, *pooled[1])
            dout, pooled = ops.maxpool2x2_bwd(d, pooled[0], H, W), None
        if self._implicit(cin, cout):
            if dout_is_dpre:
                dpre = dout
            elif pooled is not None and len(pooled) > 2:
                dpre = ops.maxpool2x2_relu_bwd(dout.contiguous(), pooled[0], None, drop=pooled[1], pooled=pooled[2], hw=(H, W))
            elif pooled is not None:
                dpre = ops.maxpool2x2_relu_bwd(dout.contiguous(), pooled[0], out, drop=pooled[1])
            elif out_dropped and mask is not None:
                dpre = ops.relu_bwd_scaled(dout.contiguous(), out, mask[0])
            else:
                dpre = ops.relu_bwd(dout.contiguous(), out, drop=mask)         # [N,H,W,cout] bf16
            if VGG_WGRAD_SIDE and need_dx:
                # nobody waits for the weight gradient: side lane, beside the data gradient and the (HBM-bound) un-pooling
                # pass that follow on the main stream; joined at the end of backward()
                with ops.side_lane(dpre.device, keep=(x_in, dpre), lane=1):
                    ops.conv3x3_bwd_weight_bias(x_in, dpre, gw, gb)
                self._side_used = True
            else:
                ops.conv3x3_bwd_weight_bias(x_in, dpre, gw, gb)
            if not need_dx:
                return None
            if below is not None:
                return ops.conv3x3_bwd_data_relu(dpre, self._conv_images(name)[1], below[0], drop=below[1],
                                                 dropped=len(below) > 2 and below[2] and below[1] is not None)
            return ops.conv3x3_bwd_data(dpre, self._conv_images(name)[1])
        if self._direct(cin, cout) and not need_dx:
            if dout_is_dpre:
                dpre = dout
            elif out_dropped and mask is not None:
                dpre = ops.relu_bwd_scaled(dout.contiguous(), out, mask[0])
            else:
                dpre = ops.relu_bwd(dout.contiguous(), out, drop=mask)
            ops.conv3x3_smallc_bwd_weight_bias(x_in, dpre, gw, gb)
            return None
        w2d = sh[self.prefix + name + '/weight'].view(9 * cin, cout)
        ldp = (9 * cin + 7) // 8 * 8
        din = torch.empty((N, H, W, cin), dtype=torch.float32, device=dout.device) if need_dx else None
        gb_acc = torch.zeros_like(gb)
        for ci, c0 in enumerate(range(0, N, CHUNK_FRAMES)):
            sl = slice(c0, c0 + CHUNK_FRAMES)
            n = out[sl].shape[0]
            # a chunk's elements start at element c0*H*W*cout of the layer's tensor = Philox block offset + that / 4
            if dout_is_dpre:
                dpre = dout[sl].contiguous().view(n * H * W, cout)
            else:
                dchunk = None if mask is None else (mask[0], mask[1], mask[2] + (c0 * H * W * cout) // 4)
                dpre = ops.relu_bwd(dout[sl].contiguous(), out[sl].contiguous(), drop=dchunk).view(n * H * W, cout)
            patches = ops.im2col3x3(x_in[sl].contiguous(), ldp)
            if self.dtype == ASR_BF16 and ldp % 8 == 0 and cout % 8 == 0:
                # all ldp columns (the padding ones are zero): M = 32 instead of 27 keeps this K = frames*F*W ~ 2 M
                # contraction on the lean reduction-major kernel (the generic one runs it at 0.3 TB/s: 1.1 ms per chunk)
                full = ops.gemm(patches, dpre, transA=True, out_dtype=ASR_F32)
                if ci == 0:
                    gw.copy_(full[:9 * cin])
                else:
                    gw.add_(full[:9 * cin])
            else:
                ops.gemm(patches[:, :9 * cin], dpre, transA=True, out=gw, accumulate=(ci > 0))
            gb_acc += ops.colsum(dpre)
            if need_dx:
                dpat = ops.gemm(dpre, w2d, transB=True, out_dtype=ASR_F32)
                din[sl] = ops.col2im3x3(dpat, n, H, W, cin)
        gb.copy_(gb_acc)
        return din

    def backward(self, dout_btd):
        """dout [B,T,256] fp32 (gradient w.r.t. the front-end output)."""
        c, st = self.ctx, self.store
        sh = st.shadow(self.dtype)
        N = c['N']
        m = c['masks']
        d = dout_btd.reshape(c['B'] * c['T'], 256).contiguous()
        if c['pack'] is not None:
            d = ops.embedding_gather(d, c['pack'][0])
        dpre = ops.relu_bwd(d, c['br'], drop=m.get('br'))
        ops.gemm(c['flat'], dpre, transA=True, out=st.g(self.prefix + 'bridge/weights'))
        ops.colsum(dpre, out=st.g(self.prefix + 'bridge/biases'))
        dflat = ops.gemm(dpre, sh[self.prefix + 'bridge/weights'], transB=True, out_dtype=ASR_F32)
        H2, W2 = (self.F + 1) // 2, (self.W + 1) // 2
        H4, W4 = (H2 + 1) // 2, (W2 + 1) // 2
        dp2 = dflat.view(N, H4, W4, 128)
        # where both neighbours are implicit-GEMM layers the ReLU / dropout backward of the lower one sits in the data
        # gradient's epilogue (no fp32 activation gradient is written): conv4 -> a3, conv2 -> a1
        fuse43 = self._implicit(*CONVS[3][1:]) and self._implicit(*CONVS[2][1:])
        fuse21 = self._implicit(*CONVS[1][1:]) and self.dtype == ASR_BF16
        fd = bool(c.get('fused_drop'))      # a1 / a3 are the DROPPED outputs, the pooled tensors carry their masks
        # (the pooled tensor as the mask source of the un-pooling pass works with or without dropout on the implicit path)
        pool_src = (self._implicit(*CONVS[3][1:]) and self._implicit(*CONVS[1][1:]) and
                    (fd or self.fused_drop))
        da3d = self._conv_bwd(dp2, c['a4'], None, c['a3d'], CONVS[3], sh,
                              pooled=(c['arg2'], m.get('p2'), c['p2d']) if pool_src else (c['arg2'], m.get('p2')),
                              below=(c['a3'], m.get('a3'), fd) if fuse43 else None)
        dp1d = self._conv_bwd(da3d, c['a3'], m.get('a3'), c['p1d'], CONVS[2], sh, dout_is_dpre=fuse43, out_dropped=fd)
        da1d = self._conv_bwd(dp1d, c['a2'], None, c['a1d'], CONVS[1], sh,
                              pooled=(c['arg1'], m.get('p1'), c['p1d']) if pool_src else (c['arg1'], m.get('p1')),
                              below=(c['a1'], m.get('a1'), fd) if fuse21 else None)
        self._conv_bwd(da1d, c['a1'], m.get('a1'), c['x0'], CONVS[0], sh, need_dx=False, dout_is_dpre=fuse21,
                       out_dropped=fd)
        if getattr(self, '_side_used', False):
            ops.join_side(dout_btd.device)
            self._side_used = False
        self.ctx = None


class _VGGRecurrentMixin(object):
    """VGG front-end + the recurrent stack of the base class."""

    def _vgg_init(self, input_size, splice, num_stack, parameter_init, dtype):
        self.front = _VGGFrontEnd(input_size, splice, num_stack, parameter_init, ops.dtype_id(dtype))
        self.input_size, self.splice, self.num_stack = input_size, splice, num_stack

    def build(self, store, input_dim, rng, scope_prefix=''):
        assert input_dim == self.front.F * self.front.W * 3
        self.front.build(store, rng, scope_prefix)
        return super(_VGGRecurrentMixin, self).build(store, 256, rng, scope_prefix)

    def __call__(self, inputs, inputs_seq_len, keep_prob, is_training, drop_masks=None, rng_state=None):
        if self.layers is None:
            store = ParamStore(inputs.device)
            self.build(store, inputs.shape[-1], np.random.RandomState(self.seed))
            store.finalize()
        # the lengths as the caller gave them on the host (the model classes leave them in _lens_host); a device vector
        # seen for the first time costs one read-back, remembered by ops.host_ints
        lens_host = getattr(self, '_lens_host', None)
        self._lens_host = None
        if lens_host is None or len(lens_host) != inputs.shape[0]:
            lens_host = ops.host_ints(inputs_seq_len)
        x = self.front.forward(inputs.contiguous(), float(keep_prob), is_training,
                               rng_state or (self.seed, 1 << 50), seq_len=lens_host)
        return super(_VGGRecurrentMixin, self).__call__(x, inputs_seq_len, keep_prob, is_training, drop_masks,
                                                        rng_state)

    def backward(self, d_outputs, d_final=None):
        dx = super(_VGGRecurrentMixin, self).backward(d_outputs, d_final, need_input_grad=True)   # [T,Bp,256]
        B = self.batch
        self.front.backward(dx[:, :B].transpose(0, 1).contiguous())
        return None


class VGGBLSTMEncoder(_VGGRecurrentMixin, BLSTMEncoder):
    """models/encoders/core/vgg_blstm.py:17 VGGBLSTMEncoder."""

    def __init__(self, input_size, splice, num_stack, num_units, num_proj, num_layers, lstm_impl, use_peephole,
                 parameter_init, clip_activation, time_major=False, name='vgg_blstm_encoder', dtype=ASR_F32, seed=0):
        BLSTMEncoder.__init__(self, num_units, num_proj, num_layers, lstm_impl, use_peephole, parameter_init,
                              clip_activation, time_major, name, dtype, seed)
        self._vgg_init(input_size, splice, num_stack, parameter_init, dtype)


class VGGLSTMEncoder(_VGGRecurrentMixin, LSTMEncoder):
    """models/encoders/core/vgg_lstm.py VGGLSTMEncoder (same front-end, unidirectional stack)."""

    def __init__(self, input_size, splice, num_stack, num_units, num_proj, num_layers, lstm_impl, use_peephole,
                 parameter_init, clip_activation, time_major=False, name='vgg_lstm_encoder', dtype=ASR_F32, seed=0):
        LSTMEncoder.__init__(self, num_units, num_proj, num_layers, lstm_impl, use_peephole, parameter_init,
                             clip_activation, time_major, name, dtype, seed)
        self._vgg_init(input_size, splice, num_stack, parameter_init, dtype)

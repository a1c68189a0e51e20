"""GRU encoders on the HIP kernels of csrc/gru.hip.

Mirror of models/encoders/core/gru.py (class GRUEncoder :9-76, class BGRUEncoder :79-152 of the reference): same
constructor arguments (num_units, num_layers, parameter_init, time_major, name) and
__call__(inputs, inputs_seq_len, keep_prob, is_training) -> (outputs, final_state).
tf.contrib.rnn.GRUCell under DropoutWrapper(output_keep_prob) and (bidirectional_)dynamic_rnn(sequence_length).
Variable names follow TF 1.3:
    bgru_hidden<i>/{fw,bw}/gru_cell/{gates,candidate}/{kernel,bias}                      (BGRUEncoder)
    multi_gru/rnn/multi_rnn_cell/cell_<i-1>/gru_cell/{gates,candidate}/{kernel,bias}     (GRUEncoder)
gates/kernel [(Din+H), 2H] (columns r | u), gates/bias [2H] initialised to 1, candidate/kernel [(Din+H), H],
candidate/bias [H] zero; kernels uniform(+-parameter_init) (the variable_scope initializer).
fp32 (the recurrence kernels are fp32; a bf16 request runs them all the same).
"""
import numpy as np
import torch

from .... import ops
from ...._lib import ASR_F32
from ....utils.parameter import ParamStore

DIRS = ('fw', 'bw')


class GRULayer(object):
    def __init__(self, store, bases, din, H):
        self.store, self.bases = store, bases
        self.ndir = len(bases)
        self.din, self.H = din, H
        self.ctx = None
        self.grad_event = None

    def var_names(self):
        names = []
        for b in self.bases:
            names += [b + '/gates/kernel', b + '/gates/bias', b + '/candidate/kernel', b + '/candidate/bias']
        return names

    def forward(self, x, seq_len, tmax, mask=None, save=True):
        """x [T,B,din] fp32 time-major -> out [T,B,ndir*H] (times the dropout mask), h_final [ndir,B,H]."""
        st = self.store
        T, B, din = x.shape
        H, ndir = self.H, self.ndir
        x2d = x.view(T * B, din)
        xg = torch.empty((T, B, ndir * 2 * H), dtype=torch.float32, device=x.device)
        xc = torch.empty((T, B, ndir * H), dtype=torch.float32, device=x.device)
        wgh = torch.empty((ndir, H, 2 * H), dtype=torch.float32, device=x.device)
        wch = torch.empty((ndir, H, H), dtype=torch.float32, device=x.device)
        for d, b in enumerate(self.bases):
            wg, wc = st[b + '/gates/kernel'], st[b + '/candidate/kernel']
            ops.gemm(x2d, wg[:din], bias=st[b + '/gates/bias'], out=xg.view(T * B, -1)[:, d * 2 * H:(d + 1) * 2 * H])
            ops.gemm(x2d, wc[:din], bias=st[b + '/candidate/bias'], out=xc.view(T * B, -1)[:, d * H:(d + 1) * H])
            wgh[d].copy_(wg[din:])
            wch[d].copy_(wc[din:])
        sv = ops.gru_fwd(xg, xc, wgh, wch, seq_len, tmax, H, ndir)
        out = sv['hout'] if mask is None else ops.apply_mask(sv['hout'], mask)
        if save:
            self.ctx = dict(x=x, sv=sv, wgh=wgh, wch=wch, seq_len=seq_len, tmax=tmax, mask=mask)
        return out, sv['h_final']

    def backward(self, dout, d_h_final=None, need_dx=True):
        """dout [T,B,ndir*H] fp32 -> dx [T,B,din] (or None).  Fills store.grad."""
        c, st = self.ctx, self.store
        x, sv = c['x'], c['sv']
        T, B, din = x.shape
        H, ndir = self.H, self.ndir
        if c['mask'] is not None:
            dout = ops.apply_mask(dout, c['mask'])
        wghT = torch.stack([ops.transpose2d(c['wgh'][d]) for d in range(ndir)])          # [ndir, 2H, H]
        wchT = torch.stack([ops.transpose2d(c['wch'][d]) for d in range(ndir)])          # [ndir, H, H]
        dgate, dcand = ops.gru_bwd(dout.contiguous(), d_h_final, sv, wghT, wchT, c['seq_len'], c['tmax'], H, ndir)
        x2d = x.view(T * B, din)
        h2d = sv['hout'].view(T * B, ndir * H)
        rh2d = sv['rh'].view(T * B, ndir * H)
        dg2d = dgate.view(T * B, ndir * 2 * H)
        dc2d = dcand.view(T * B, ndir * H)
        dx = torch.zeros((T, B, din), dtype=torch.float32, device=x.device) if need_dx else None
        # ONE marker behind the recurrence kernel: the weight gradients start there on side lanes (one per direction) and run
        # beside the dx products and the BPTT of the layer below -- nobody waits for them before the clip (the encoder's
        # backward joins the lanes); the main stream carries only what the layer below needs (as LSTMLayer.backward)
        fork = ops.stream_event()
        if need_dx:
            for d, b in enumerate(self.bases):
                dg = dg2d[:, d * 2 * H:(d + 1) * 2 * H]
                dc = dc2d[:, d * H:(d + 1) * H]
                ops.gemm(dg, st[b + '/gates/kernel'][:din], transB=True, out=dx.view(T * B, din), accumulate=True)
                ops.gemm(dc, st[b + '/candidate/kernel'][:din], transB=True, out=dx.view(T * B, din), accumulate=True)
        done = []
        for d, b in enumerate(self.bases):
            with ops.side_lane(x.device, keep=(x, sv['hout'], sv['rh'], dgate, dcand), lane=1 + (d % 2), after=fork):
                dg = dg2d[:, d * 2 * H:(d + 1) * 2 * H]
                dc = dc2d[:, d * H:(d + 1) * H]
                gk, ck = st.g(b + '/gates/kernel'), st.g(b + '/candidate/kernel')
                ops.gemm(x2d, dg, transA=True, out=gk[:din])
                ops.gemm(x2d, dc, transA=True, out=ck[:din])
                if T > 1:
                    if d == 0:   # forward direction: h_prev(t) = h(t-1)
                        ops.gemm(h2d[:(T - 1) * B, d * H:(d + 1) * H], dg[B:], transA=True, out=gk[din:])
                    else:        # backward direction: h_prev(t) = h(t+1) (zero beyond len-1)
                        ops.gemm(h2d[B:, d * H:(d + 1) * H], dg[:(T - 1) * B], transA=True, out=gk[din:])
                else:
                    gk[din:].zero_()
                ops.gemm(rh2d[:, d * H:(d + 1) * H], dc, transA=True, out=ck[din:])          # r * h_prev, same frame
                ops.colsum(dg, out=st.g(b + '/gates/bias'))
                ops.colsum(dc, out=st.g(b + '/candidate/bias'))
                if d % 2 > 0:
                    done.append(ops.stream_event())
        with ops.side_lane(x.device, lane=1, after=fork):
            for ev in done:
                ops.wait_event(ev)
            # every gradient of this layer is complete at this point of side lane 1 (the data-parallel step hangs the
            # layer's clip + all-reduce on it)
            self.grad_event = ops.stream_event()
        self.ctx = None
        return dx


class _GRUEncoderBase(object):
    ndir = 1

    def __init__(self, num_units, num_layers, parameter_init, time_major=False, name='gru_encoder', dtype=ASR_F32, seed=0):
        self.num_units = num_units
        self.num_layers = num_layers
        self.parameter_init = parameter_init
        self.time_major = time_major
        self.name = name
        self.dtype = ASR_F32                             # the GRU kernels are fp32
        self.seed = seed
        self.layers = None
        self.store = None
        self.grad_ready_hook = None
        self.want_f32_outputs = True
        self.num_layers_sub = None

    def _bases(self, i, prefix):
        raise NotImplementedError

    def build(self, store, input_dim, rng, scope_prefix=''):
        self.store = store
        self.layers = []
        din, H = input_dim, self.num_units
        u = lambda *s: rng.uniform(-self.parameter_init, self.parameter_init, size=s)
        for i in range(1, self.num_layers + 1):
            bases = self._bases(i, scope_prefix)
            for b in bases:
                store.declare(b + '/gates/kernel', (din + H, 2 * H), u(din + H, 2 * H))
                store.declare(b + '/gates/bias', (2 * H,), np.ones(2 * H))
                store.declare(b + '/candidate/kernel', (din + H, H), u(din + H, H))
                store.declare(b + '/candidate/bias', (H,), np.zeros(H))
            self.layers.append(GRULayer(store, bases, din, H))
            din = self.ndir * H
        self.output_dim = self.ndir * H
        return self.output_dim

    def __call__(self, inputs, inputs_seq_len, keep_prob, is_training, drop_masks=None, rng_state=None):
        """inputs [B,T,input_size] fp32 (cuda); inputs_seq_len [B].  Returns outputs [T,B,ndir*H] if time_major else
        [B,T,ndir*H] and the final state: GRUEncoder -> one [B,H] tensor per layer (MultiRNNCell's state tuple),
        BGRUEncoder -> (h_fw, h_bw) of the last layer (gru.py:75, :150)."""
        if self.layers is None:
            store = ParamStore(inputs.device)
            self.build(store, inputs.shape[-1], np.random.RandomState(self.seed))
            store.finalize()
        B, T, _ = inputs.shape
        lens_host = getattr(self, '_lens_host', None)    # the caller's host lengths (model classes), else one read-back
        self._lens_host = None
        if lens_host is None or len(lens_host) != B:
            lens_host = ops.host_ints(inputs_seq_len)
        pad = (-B) % 16
        if pad:  # the kernels tile 16 utterances; pad with zero-length rows
            inputs = torch.cat([inputs, inputs.new_zeros(pad, T, inputs.shape[2])], 0)
            inputs_seq_len = torch.cat([inputs_seq_len, inputs_seq_len.new_zeros(pad)], 0)
        self.batch = B
        seq_len = inputs_seq_len.to(torch.int32).contiguous()
        tmax = int(min(max(int(lens_host.max()), 0), T)) if len(lens_host) else 0
        x = ops.bt_to_tb(inputs.contiguous(), ASR_F32)
        Bp = x.shape[1]
        if rng_state is None and drop_masks is None and is_training and keep_prob is not None and keep_prob < 1.0:
            self._dropout_calls = getattr(self, '_dropout_calls', 0) + 1
            rng_state = (self.seed, self._dropout_calls << 40)
        finals = []
        for li, layer in enumerate(self.layers):
            mask = None
            if is_training and drop_masks is not None:
                mask = drop_masks[li]
            elif is_training and rng_state is not None and keep_prob < 1.0:
                mask = ops.dropout_mask((T, Bp, self.ndir * self.num_units), keep_prob, rng_state[0],
                                        rng_state[1] + li * (1 << 32), x.device)
            x, hf = layer.forward(x, seq_len, tmax, mask)
            finals.append(hf)
        self.seq_len_padded = seq_len
        self._out_op = self._out_tm = x
        self._finals = finals
        out_user = x[:, :B]
        if not self.time_major:
            out_user = out_user.transpose(0, 1)
        if self.ndir == 2:
            final_state = (finals[-1][0, :B], finals[-1][1, :B])
        else:
            final_state = tuple(f[0, :B] for f in finals)
        return out_user, final_state

    def backward(self, d_outputs, d_final=None, need_input_grad=False, d_outputs_sub=None):
        """d_outputs [T,Bpad,ndir*H] fp32 (time-major, padded batch)."""
        dx = d_outputs
        for li in reversed(range(len(self.layers))):
            dx = self.layers[li].backward(dx.contiguous(), None, need_dx=(li > 0 or need_input_grad))
            if self.grad_ready_hook is not None:
                self.grad_ready_hook(li, self.layers[li])
        # the model's head gradients were issued on side lane 1 (CTC._backward): clip / all-reduce / update on the main
        # stream must be ordered after them, and the lane's keep list is only released here (blstm.py does the same)
        ops.join_side(d_outputs.device)
        return dx


class GRUEncoder(_GRUEncoderBase):
    """models/encoders/core/gru.py:9 GRUEncoder."""
    ndir = 1

    def _bases(self, i, prefix):
        return ['%smulti_gru/rnn/multi_rnn_cell/cell_%d/gru_cell' % (prefix, i - 1)]


class BGRUEncoder(_GRUEncoderBase):
    """models/encoders/core/gru.py:79 BGRUEncoder."""
    ndir = 2

    def __init__(self, num_units, num_layers, parameter_init, time_major=False, name='bgru_encoder', dtype=ASR_F32, seed=0):
        _GRUEncoderBase.__init__(self, num_units, num_layers, parameter_init, time_major, name, dtype, seed)

    def _bases(self, i, prefix):
        return ['%sbgru_hidden%d/%s/gru_cell' % (prefix, i, d) for d in DIRS]

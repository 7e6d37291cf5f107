"""MaskTransUnet on MI355X: the nn.Module surface of model/trans_3DUnet.py:150-222 over HIP kernels.

Drop-in contract:
  * `get_model_dict('MaskTransUnet')(num_layers, roi_size_list, is_roi_list, dim_input, dim_output,
    kernel_size=3, dropout=0.3)` builds a module whose `state_dict()` has exactly the reference's
    614 keys / shapes (checkpoints load with strict=True, including the 14 unused
    `decode.bridge_list.4.transformer.pos_encoders.{1..7}` tensors).
  * `forward(x[B,1,H,W,D])` returns `(probs[B,C,H,W,D], [mask]*4)` in training mode and the one-hot
    arg-max tensor in eval mode.  Returned tensors are channels-last in memory, exposed through a
    permuted view so indexing/shape follow the reference.

The parameter containers are plain torch modules (they provide names, initialisation and the
optimizer/DDP plumbing); their own forward() is never called - all arithmetic goes through
`ops` (libltu_hip.so).  The module fails loudly on CPU tensors: there is no fallback path.
"""
import os

import torch
import torch.nn as nn

from . import ops

HEAD_DIM = 32      # model/Unet_3Dblock.py:1294
N_LAYERS = 8


def _cl(t):
    """logical [B,C,H,W,D] view of a channels-last [B,H,W,D,C] tensor"""
    return t.permute(0, 4, 1, 2, 3)


class _Group(nn.Module):
    """anonymous parameter container"""


def _transformer_layer(d):
    """parameter names of model/trans_block.py:127-211"""
    lay = _Group()
    lay.self_attn = _Group()
    lay.self_attn.linears = nn.ModuleList([nn.Linear(d, d) for _ in range(4)])
    lay.linear1 = nn.Linear(d, 2 * d)
    lay.linear2 = nn.Linear(2 * d, d)
    lay.layer_norm1 = nn.LayerNorm(d, eps=1e-6)
    lay.layer_norm2 = nn.LayerNorm(d, eps=1e-6)
    return lay


def _pos_encoder(d):
    pe = _Group()
    pe.proj = nn.Conv3d(d, d, 3, padding=1, groups=d)
    return pe


class _SeedStream:
    """Distinct dropout stream ids inside one forward pass."""

    def __init__(self, base):
        self.base = int(base) & 0xFFFFFFFFFF
        self.n = 0

    def next(self):
        self.n += 1
        return (self.base << 20) + self.n


WPREP_CHUNK = 4096          # LTU_WPREP_CHUNK of include/ltu_hip.h
# width (workgroups) of the operand refreshes that run on the side stream beside the encoder / the bottleneck transformer (0: full;
# 32 / 64 / 128 made the refreshes the critical path: +1.0 / +0.25 / +0.1 ms per step)
SIDE_REFRESH_BLOCKS = int(os.environ.get('LTU_SIDE_REFRESH_BLOCKS', '0'))


class _WeightStore:
    """Every GEMM weight operand of the model in the activation dtype, refreshed by ONE kernel launch per step.

    conv: wf [CoP][27][CiP] + wd [CiP][27][CoP] (+ a padded fp32 bias for the mask heads);
    linear / 1x1x1 conv groups: forward operands (bf16 copies; the fp32 masters themselves in fp32 mode) and cat(W)^T.
    """

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.recs = []          # (src, dst, kind, R, C, p0, p1) of whole tensors; raw = the same with pointers and a pad field
        self.raw = []
        self.pair = {}
        self.conv, self.lin = {}, {}
        self.table = None

    def add_conv(self, conv, cip=None, cop=None):
        w = conv.weight
        Co, Ci = w.shape[0], w.shape[1]
        cip, cop = cip or Ci, cop or Co
        wf = torch.empty((cop, 27, cip), device=self.device, dtype=self.dtype)
        wd = torch.empty((cip, 27, cop), device=self.device, dtype=self.dtype)
        self.recs.append((w, wf, 2, Co, Ci, cop, cip))
        self.recs.append((w, wd, 3, Co, Ci, cop, cip))
        bias = conv.bias
        if cop != Co:
            bias = torch.zeros(cop, device=self.device, dtype=torch.float32)
            self.recs.append((conv.bias, bias, 4, 1, Co, 0, 0))
        self.conv[id(conv)] = ops.ConvPrep(wf, wd, bias)

    def add_upconv(self, conv):
        w = conv.weight
        Co, Ci = w.shape[0], w.shape[1]
        wf = torch.empty((8, Co, 8, Ci), device=self.device, dtype=self.dtype)
        wd = torch.empty((Ci, 64, Co), device=self.device, dtype=self.dtype)
        self.recs.append((w, wf, 5, Co, Ci, Co, Ci))
        self.recs.append((w, wd, 6, Co, Ci, Co, Ci))
        self.conv[id(conv)] = ops.UpConvPrep(wf, wd)

    def add_conv_pair(self, key, conv_a, conv_b, n1):
        """conv_a and conv_b (3x3x3, stride 1, same input) fused: conv_b's outputs are zero-padded to n1 channels"""
        wa, wb = conv_a.weight, conv_b.weight
        Ca, Ci = wa.shape[0], wa.shape[1]
        wf = torch.empty((Ca + n1, 27, Ci), device=self.device, dtype=self.dtype)
        wd = torch.empty((Ci, 27, Ca + n1), device=self.device, dtype=self.dtype)
        bias = torch.zeros(Ca + n1, device=self.device, dtype=torch.float32)
        self.raw.append((conv_a, conv_b, n1, wf, wd, bias))
        self.pair[key] = ops.PairPrep(wf, wd, bias, Ca, n1)

    def add_linear(self, key, weights, group=None, frag=False):
        """weights: list of [N,K(,1,1,1)] parameters sharing K (a fused q,k,v group or a single projection)"""
        N, K = weights[0].shape[0], weights[0].shape[1]
        if self.dtype == torch.float32:
            fw = list(weights)
        else:
            fw = []
            for w in weights:
                c = torch.empty((N, K), device=self.device, dtype=self.dtype)
                self.recs.append((w, c, 0, N, K, 0, 0))
                fw.append(c)
        wt = torch.empty((K, N * len(weights)), device=self.device, dtype=self.dtype)
        for i, w in enumerate(weights):
            self.recs.append((w, wt, 1, N, K, N * len(weights), i * N))
        fr = frT = None
        if frag and self.dtype == torch.bfloat16 and N % 32 == 0 and K % 32 == 0:
            nw = len(weights)
            fr = torch.empty(nw * N * K, device=self.device, dtype=self.dtype)      # MFMA fragment order (kind 8): column tiles of
            for i, w in enumerate(weights):                                          # cat(W) = the weights' tiles back to back
                self.recs.append((w, fr[i * N * K:(i + 1) * N * K], 8, N, K, 0, 0))
            if nw == 1:
                frT = torch.empty(N * K, device=self.device, dtype=self.dtype)      # ... and of the transpose (kind 9)
                self.recs.append((weights[0], frT, 9, N, K, 0, 0))
        self.lin[key] = ops.LinPrep(fw, wt, group, fr, frT)

    def finalize(self):
        import numpy as np
        rows = [(src.data_ptr(), dst.data_ptr(), kind, R, C, p0, p1, 0) for src, dst, kind, R, C, p0, p1 in self.recs]
        for conv_a, conv_b, n1, wf, wd, bias in self.raw:
            rows += ops.pair_records(conv_a.weight, conv_a.bias, conv_b.weight, conv_b.bias, n1, wf, wd, bias)
        rec = np.zeros(len(rows), dtype=ops.WPREP_DTYPE)
        for i, r in enumerate(rows):
            rec[i] = r
        self.ptrs = [r[0] for r in rows]
        self.srcs = [src for src, *_ in self.recs] + [t for conv_a, conv_b, *_ in self.raw
                                                      for t in (conv_a.weight, conv_b.weight, conv_a.weight, conv_b.weight, conv_a.bias, conv_b.bias)]
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
        # work list: (record, chunk) pairs of WPREP_CHUNK destination elements, one workgroup each
        chunks = []
        for i, (_, _, kind, R, C, p0, p1, pad) in enumerate(rows):
            if kind in (0, 1, 4, 8, 9):
                n = R * C
            elif kind in (2, 3):
                n = p0 * 27 * p1
            elif kind == 7:
                n = (pad & 0xffff) * 27 * p1
            else:
                n = 64 * p0 * p1
            chunks += [(i, c) for c in range((n + WPREP_CHUNK - 1) // WPREP_CHUNK)]
        self.nchunks = len(chunks)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(self.device)
        # the records added before `mark_first()` (the encoder's convs) form part 0 of the work list, the rest part 1: the model
        # refreshes part 1 on the side stream, beside the encoder (ops.Context.side_run)
        # part p = the chunks of the records added before the (p + 1)-th `mark()`: bounds[p] .. bounds[p + 1]
        marks = list(getattr(self, 'marks', [])) + [len(rows)]
        self.bounds = [0] + [sum(1 for i, _ in chunks if i < m) for m in marks]

    def stale(self):
        """parameter storage moved (e.g. .to(), load with assign): the table must be rebuilt"""
        return any(src.data_ptr() != p0 for src, p0 in zip(self.srcs, self.ptrs))

    def mark(self):
        """the records added so far (since the previous mark) form one part of the work list; the pair records of `add_conv_pair`
        are appended behind all others by `finalize` and therefore belong to the LAST part"""
        self.marks = getattr(self, 'marks', []) + [len(self.recs)]

    def refresh(self, part=None, blocks=None):
        """part None: everything; p: the operands added between the p-th and the (p + 1)-th `mark()` (the last part: the rest).
        blocks: width of the launch (workgroups) - a refresh that runs on the side stream beside the step stays narrow, so that the
        step's own kernels find free workgroup slots (SIDE_REFRESH_BLOCKS)"""
        lo, hi = (0, self.nchunks) if part is None else (self.bounds[part], self.bounds[part + 1])
        if hi > lo:
            if blocks:
                ops._lib.config_set('LTU_WPREP_BLOCKS', blocks)
            try:
                ops._lib.call('ltu_weight_prep_chunks', self.table.data_ptr(), self.chunks.data_ptr() + lo * 8, hi - lo,
                              ops.F32 if self.dtype == torch.float32 else ops.BF16, torch.cuda.current_stream().cuda_stream)
            finally:
                if blocks:
                    ops._lib.config_set('LTU_WPREP_BLOCKS', None)


class MaskTransUnet(nn.Module):
    def __init__(self, num_layers, roi_size_list, is_roi_list, dim_input, dim_output, kernel_size=3, dropout=0.3,
                 act_dtype=torch.float32):
        super().__init__()
        if kernel_size != 3:
            raise ValueError('only the reference default kernel_size=3 is supported')
        if dim_input != 1:
            raise ValueError('window embedding of the reference assumes dim_input=1 (Unet_3Dblock.py:131-132)')
        L = list(num_layers)
        nl = len(L)
        self.num_layers, self.roi_size_list, self.is_roi_list = L, list(roi_size_list), list(is_roi_list)
        self.dim_input, self.dim_output, self.kernel_size, self.dropout = dim_input, dim_output, kernel_size, dropout
        self.act_dtype = act_dtype
        self._step = 0

        enc = self.encode = _Group()
        enc.block_list = nn.ModuleList()
        for i in range(1, nl):
            blk = _Group()
            blk.conv1 = nn.Conv3d(L[i - 1], L[i - 1], 3, padding=1)
            blk.conv2 = nn.Conv3d(L[i - 1], L[i], 3, stride=(2, 2, (i - 1) % 2 + 1), padding=1)
            enc.block_list.append(blk)
        enc.input_block = nn.Conv3d(dim_input * 4, L[0], 3, padding=1)

        dec = self.decode = _Group()
        dec.bridge_list = nn.ModuleList()
        for i in range(nl - 1):
            br = _Group()
            if self.is_roi_list[i]:
                d = min(4 * L[i], 256)
                tr = br.transformer = _Group()
                tr.down_embed = _Group()
                tr.down_embed.module_list = nn.Sequential(nn.Sequential(nn.Conv3d(L[i], d, 3, stride=2, padding=1)))
                tr.up_embed = _Group()
                tr.up_embed.module_list = nn.Sequential(nn.Sequential(nn.Identity(), nn.Conv3d(d, L[i], 3, padding=1)))
                tr.pos_encoder = _pos_encoder(d)
                tr.layers = nn.ModuleList([_transformer_layer(d) for _ in range(N_LAYERS)])
            dec.bridge_list.append(br)
        bott = _Group()
        bott.transformer = _Group()
        bott.transformer.pos_encoders = nn.ModuleList([_pos_encoder(L[-1]) for _ in range(N_LAYERS)])
        bott.transformer.layers = nn.ModuleList([_transformer_layer(L[-1]) for _ in range(N_LAYERS)])
        dec.bridge_list.append(bott)
        dec.mask_conv_list = nn.ModuleList([nn.Conv3d(L[i], dim_output, 3, padding=1) for i in range(1, nl)])
        dec.att_conv_list = nn.ModuleList()
        for i in range(1, nl):
            g = _Group()
            g.W_x = nn.Sequential(nn.Conv3d(L[i - 1], L[i - 1], 1))
            g.W_g = nn.Sequential(nn.Conv3d(L[i], L[i - 1], 1))
            g.psi = nn.Sequential(nn.Conv3d(L[i - 1], 1, 1))
            dec.att_conv_list.append(g)
        dec.block_list = nn.ModuleList()
        for i in range(1, nl):
            blk = _Group()
            blk.conv1 = nn.Conv3d(L[-i], L[-i - 1], 3, padding=1)
            blk.conv2 = nn.Conv3d(2 * L[-i - 1], L[-i - 1], 3, padding=1)
            dec.block_list.append(blk)
        dec.final_block = nn.Conv3d(L[0], dim_output * 4, 3, padding=1)
        self.last_boxes = []
        self._store = None
        self._infer_ctx = None       # ops.Context of this model's no-grad forwards (own scratch arena)

    def _final_cop(self):
        n = 4 * self.dim_output
        return n if self.act_dtype == torch.float32 else (n + 7) // 8 * 8

    # ------------------------------------------------------------------ prepared weight operands
    def _weights(self, device):
        st = self._store
        if st is not None and st.device == device and st.dtype == self.act_dtype and not st.stale():
            return st
        st = _WeightStore(device, self.act_dtype)
        enc, dec = self.encode, self.decode
        st.add_conv(enc.input_block, cip=8)
        for blk in enc.block_list:
            st.add_conv(blk.conv1)
            st.add_conv(blk.conv2)
        st.mark()                            # part 0: the encoder (refreshed on the main stream at the start of the step)
        # part 1: the ROI bridges - needed last in the forward pass; refreshed beside the bottleneck transformer
        def add_transformer(tr):
            if hasattr(tr, 'down_embed'):
                st.add_conv(tr.down_embed.module_list[0][0])
                st.add_upconv(tr.up_embed.module_list[0][1])
            for lay in tr.layers:
                lin = lay.self_attn.linears
                # weight gradients of a layer's projections go out as one group, launched by the qkv backward (the layer's last)
                st.add_linear((id(lay), 'qkv'), [lin[0].weight, lin[1].weight, lin[2].weight], group='flush', frag=True)
                st.add_linear((id(lay), 'o'), [lin[3].weight], group='collect', frag=True)
                st.add_linear((id(lay), 'f1'), [lay.linear1.weight], group='collect', frag=True)
                st.add_linear((id(lay), 'f2'), [lay.linear2.weight], group='collect', frag=True)
        nl = len(self.num_layers)
        for br in dec.bridge_list[:nl - 1]:
            if hasattr(br, 'transformer'):
                add_transformer(br.transformer)
        st.mark()
        # part 2 (the rest): the bottleneck transformer, the decoder's convs, heads and gates - refreshed beside the encoder's
        # deeper levels.  A decoder level's conv1 and its mask head read the same upsampled tensor: one fused conv (head padded to
        # 16 / 32 columns so that the pair's data gradient keeps 32-channel chunks)
        add_transformer(dec.bridge_list[nl - 1].transformer)
        for i, blk in enumerate(dec.block_list):
            lvl = nl - 2 - i
            st.add_conv_pair(lvl, blk.conv1, dec.mask_conv_list[lvl], 16 if lvl == 0 else 32)
            st.add_conv(blk.conv2)
        # bf16 GEMM operands need channel counts in multiples of 8: 4C = 12 output channels (3 labels) are padded to 16
        st.add_conv(dec.final_block, cop=self._final_cop())
        for ag in dec.att_conv_list:
            st.add_linear(id(ag.W_x[0]), [ag.W_x[0].weight])
            st.add_linear(id(ag.W_g[0]), [ag.W_g[0].weight])
        st.finalize()
        self._store = st
        return st

    # ------------------------------------------------------------------ pieces
    def _conv_in_act(self, x, conv, stride=(1, 1, 1), res=None, p=0.0, seeds=None, x1=None, ups=False, fork=1, res_dup=None):
        """conv + InstanceNorm + LeakyReLU (+ residual, + dropout).  fork: output ports, one per consumer of the result (their
        gradients are summed inside the InstanceNorm backward kernel instead of by stand-alone add passes)"""
        y = ops.conv3d(x, conv.weight, conv.bias, stride=stride, x1=x1, ups=ups, prep=self._store.conv[id(conv)])
        return ops.instnorm_act(y, res=res, act=ops.ACT_LRELU, p=p, seed=seeds.next() if p > 0 else 0, fork=fork, res_dup=res_dup)

    def _chain_ok(self, lay, B, N, d):
        """the row-block chain kernels (csrc/tlayer.hip) serve this layer"""
        po = self._store.lin[(id(lay), 'o')]
        return (po is not None and po.frag is not None and d in (128, 256) and 64 <= B * N <= ops.TAIL_MAX_TOKENS and ops.USE_LAYER_TAIL)

    def _layer(self, lay, t, tres, B, N, d, p, seeds, last, qkv=None, nxt=None):
        """post-norm transformer layer on tokens t [B*N, d] (model/trans_block.py:148-166, 203-211).  `t` feeds the
        projections, `tres` (same values) the residual: two autograd edges whose gradients the LayerNorm backward sums.
        qkv: this layer's fused q|k|v projection, already formed by the previous layer's chain kernel (then t is None);
        nxt: the next layer, whose q|k|v projection this layer's chain kernel forms.  Returns (t, tres, qkv_next)."""
        lin = lay.self_attn.linears
        wl = self._store.lin
        if qkv is None:
            if ops.flush_per_layer(B * N):
                t = ops.wgrad_flush_point(t)     # the qkv data gradient is the last backward op of a layer
            qkv = ops.linear(t, [lin[0].weight, lin[1].weight, lin[2].weight], [lin[0].bias, lin[1].bias, lin[2].bias],
                             prep=wl[(id(lay), 'qkv')])
        po, p1, p2 = wl[(id(lay), 'o')], wl[(id(lay), 'f1')], wl[(id(lay), 'f2')]
        if self._chain_ok(lay, B, N, d):
            # the rest of the layer as one launch (csrc/tlayer.hip); with N a multiple of the kernel's 32-row blocks the
            # attention's phase B runs inside it too (the chain kernel reads its q rows and applies the merged context)
            s1, sg, s2 = ((seeds.next(), seeds.next(), seeds.next()) if p > 0 else (0, 0, 0))
            fuse = ops.FUSE_ATTN_APPLY and N % 32 == 0 and qkv.dtype == torch.bfloat16 and B * N <= ops.FUSE_ATTN_MAX_TOKENS
            a = qkv if fuse else ops.linear_attention(qkv, B, N, d)
            nx = None
            if nxt is not None:
                nl = nxt.self_attn.linears
                nx = ([nl[0].weight, nl[1].weight, nl[2].weight], [nl[0].bias, nl[1].bias, nl[2].bias], wl[(id(nxt), 'qkv')])
            out = ops.layer_tail(a, tres, (lin[3].weight, lin[3].bias, lay.linear1.weight, lay.linear1.bias, lay.linear2.weight,
                                           lay.linear2.bias, lay.layer_norm1.weight, lay.layer_norm1.bias, lay.layer_norm2.weight,
                                           lay.layer_norm2.bias), (po, p1, p2), 1e-6, p, (s1, sg, s2), fork=not last and nx is None,
                                 attn=(B, N) if fuse else None, nxt=nx)
            if nx is not None:
                return None, out[0], out[1]
            return (out, out, None) if last else (out[0], out[1], None)
        a = ops.linear_attention(qkv, B, N, d)
        a = ops.linear(a, [lin[3].weight], [lin[3].bias], prep=wl[(id(lay), 'o')])
        t, tres = ops.res_layernorm(tres, a, lay.layer_norm1.weight, lay.layer_norm1.bias, 1e-6, p, seeds.next() if p > 0 else 0,
                                    fork=True)
        f = ops.linear_gelu(t, lay.linear1.weight, lay.linear1.bias, p, seeds.next() if p > 0 else 0, prep=wl[(id(lay), 'f1')])
        f = ops.linear(f, [lay.linear2.weight], [lay.linear2.bias], prep=wl[(id(lay), 'f2')])
        out = ops.res_layernorm(tres, f, lay.layer_norm2.weight, lay.layer_norm2.bias, 1e-6, p, seeds.next() if p > 0 else 0,
                                fork=not last)
        return (out, out, None) if last else (out[0], out[1], None)

    def _token_transformer(self, layers, pos, x, p, seeds, x_res=None):
        """8 layers over the voxels of x [B,H,W,D,d]; positional conv after layer 0.  Token order does not
        matter to the layers (per-token ops + a set reduction over tokens), so voxels stay in place.
        x_res: a second port of x for the residual path of layer 0 (see ops._ports)."""
        B, H, W, D, d = x.shape
        N = H * W * D
        x = ops.wgrad_flush_point(x)         # backward: the transformer's weight gradients go out as one branch from here
        t = x.reshape(B * N, d)
        tres = t if x_res is None else x_res.reshape(B * N, d)
        qkv = None
        for n, lay in enumerate(layers):
            # layer n's chain kernel also forms layer n + 1's q|k|v projection whenever nothing lies between the two layers (the
            # positional conv follows layer 0) and both run the chain kernels
            nxt = None
            if (ops.FUSE_NEXT_QKV and 1 <= n < len(layers) - 1 and x.dtype == torch.bfloat16 and self._chain_ok(lay, B, N, d)
                    and ops.FUSE_QKV_MIN_TOKENS <= B * N <= ops.FUSE_QKV_MAX_TOKENS
                    and self._store.lin[(id(layers[n + 1]), 'qkv')].frag is not None and not ops.WGRAD_FLUSH_PER_LAYER):
                nxt = layers[n + 1]
            # the positional conv after layer 0 consumes a single tensor; so does whatever follows the last layer
            t, tres, qkv = self._layer(lay, t, tres, B, N, d, p, seeds, last=(n == 0 or n == len(layers) - 1), qkv=qkv, nxt=nxt)
            if n == 0:
                g, g_res = ops.pos_conv(t.view(B, H, W, D, d), pos.proj.weight, pos.proj.bias, p, seeds.next() if p > 0 else 0, fork=2)
                t, tres = g.view(B * N, d), g_res.view(B * N, d)
        # backward enters the transformer here: the weight gradients queued so far go out as a batch (ops.WQ_SCHEDULE 'start')
        return ops.wgrad_flush_point(tres.view(B, H, W, D, d), 'out')

    def _roi_bridge(self, br, skip, mask, roi_size, p, seeds):
        """model/Unet_3Dblock.py:717-755"""
        plan = ops.RoiPlan(mask, roi_size, 0.5)
        self.last_boxes.append(plan.box)
        tr = br.transformer
        g = ops.roi_warp(skip, plan)
        e, e_res = self._conv_in_act(g, tr.down_embed.module_list[0][0], stride=(2, 2, 2), p=p, seeds=seeds, fork=2)
        e = self._token_transformer(tr.layers, tr.pos_encoder, e, p, seeds, x_res=e_res)
        up = tr.up_embed.module_list[0][1]       # nearest x2 + conv as a sub-pixel conv (3.4x fewer multiply-adds)
        e = ops.upconv3d(e, up.weight, up.bias, prep=self._store.conv[id(up)])
        e = ops.instnorm_act(e, act=ops.ACT_LRELU, p=p, seed=seeds.next() if p > 0 else 0)
        # the un-embedded grid is 2*ceil(n/2) wide; the warp-back plan samples it with the reference's normalisation
        return ops.roi_unwarp(e, plan)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('lintransunet_amd.MaskTransUnet runs on MI355X only (no CPU fallback); move the input to cuda')
        if torch.is_grad_enabled():
            return self._forward(x)              # training: the caller's context (train.train_step recycles its arena per step)
        # inference: nothing of an earlier forward is needed any more -> recycle (and re-zero) a scratch arena.  Unless the caller
        # brought a context of its own (infer.GraphedPredictor), that is this model's private inference context, so an evaluation
        # between a training forward and its backward cannot clobber the statistics the backward still needs.
        lc = ops.current()
        if lc is ops._DEFAULT_CTX:
            if self._infer_ctx is None:
                self._infer_ctx = ops.Context()
            lc = self._infer_ctx
        with ops.use(lc):
            lc.begin_step(x.device)
            return self._forward(x)

    def _forward(self, x):
        L, nl, C = self.num_layers, len(self.num_layers), self.dim_output
        p = float(self.dropout) if self.training else 0.0
        self._step += 1
        seeds = _SeedStream(torch.initial_seed() * 1000003 + self._step)
        self.last_boxes = []
        B, _, H, W, D = x.shape
        if H % 2 or W % 2:
            raise ValueError('H and W must be even')
        enc, dec = self.encode, self.decode
        store = self._weights(x.device)
        # every cast / transposed / repacked weight of this step: the encoder's operands here, everything else in a second launch
        # that a context with a side stream (train.GraphedStep) runs beside the encoder
        lc = ops.current()
        store.refresh(0)

        # Tensors with several consumers are produced with one output port per consumer (ops._ports): a block input feeds the
        # block's conv1 and its residual (whose gradient arrives in two parts, because the block output itself has two consumers:
        # the strided conv2 and the decoder's attention gate), the encoder output feeds the bottleneck transformer's first
        # projection and its first residual.
        t = ops.window_embed(x.contiguous().float(), self.act_dtype)
        t, t_r, t_r2 = self._conv_in_act(t, enc.input_block, seeds=seeds, fork=3)
        skips = []
        nblk = len(enc.block_list)
        for i, blk in enumerate(enc.block_list):
            if i == 0 and ops.WQ_FLUSH_IN_ENCODER:
                # backward reaches this point after the first block's conv1: its weight gradient goes out beside the stem's
                # InstanceNorm backward instead of behind the last kernel of the step (only the stem's own weight gradient is left there)
                t = ops.wgrad_flush_point(t, 'enc')
            s, s_skip = self._conv_in_act(t, blk.conv1, res=t_r, res_dup=t_r2, seeds=seeds, fork=2)
            if i == 0:
                # the operands of the bottleneck transformer and the decoder: beside the encoder's deeper (latency-bound) levels - not
                # beside its first, bandwidth-bound one, whose kernels a streaming side kernel slowed down 2-5x
                lc.side_run(lambda: store.refresh(2, SIDE_REFRESH_BLOCKS))
            if ops.WQ_FLUSH_IN_ENCODER:
                # backward reaches this point after conv2's (and the deeper block's conv1's) backward: their queued weight gradients go
                # out as a batch beside this block's data gradients instead of piling up behind the last kernel of the step
                s = ops.wgrad_flush_point(s, 'enc')
            if i < nblk - 1:
                t, t_r, t_r2 = self._conv_in_act(s, blk.conv2, stride=(2, 2, i % 2 + 1), p=p, seeds=seeds, fork=3)
            else:
                t, t_r = self._conv_in_act(s, blk.conv2, stride=(2, 2, i % 2 + 1), p=p, seeds=seeds, fork=2)
            skips.append(s_skip)

        lc.side_join()                       # the operands of the bottleneck transformer and the decoder are ready
        lc.side_run(lambda: store.refresh(1, SIDE_REFRESH_BLOCKS))      # the ROI bridges' operands: beside the bottleneck transformer
        bt = dec.bridge_list[nl - 1].transformer
        t = self._token_transformer(bt.layers, bt.pos_encoders[0], t, p, seeds, x_res=t_r)
        masks = []
        for i in range(1, nl):
            lvl = nl - 1 - i
            t, t_gate = ops.trilinear_up(t, 2 if (nl - i) % 2 == 0 else 1, fork=2)      # consumers: the conv pair and the attention gate
            mc = dec.mask_conv_list[lvl]
            blk = dec.block_list[i - 1]
            t1, zm = ops.conv3d_pair(t, blk.conv1.weight, blk.conv1.bias, mc.weight, mc.bias, store.pair[lvl])
            m = ops.head_softmax(zm, C)
            masks.append(m)
            ag = dec.att_conv_list[lvl]
            skip = ops.attention_gate(skips[-i], t_gate, ag.W_x[0].weight, ag.W_x[0].bias, ag.W_g[0].weight, ag.W_g[0].bias,
                                      ag.psi[0].weight, ag.psi[0].bias, store.lin[id(ag.W_x[0])], store.lin[id(ag.W_g[0])])
            if self.is_roi_list[lvl]:
                lc.side_join()               # (first bridge only: the bridges' operands are ready)
                skip = self._roi_bridge(dec.bridge_list[lvl], skip, m.detach(), self.roi_size_list[lvl], p, seeds)
            t = ops.instnorm_act(t1, act=ops.ACT_LRELU)
            t = self._conv_in_act(t, blk.conv2, x1=skip, p=p, seeds=seeds)
        z = ops.conv3d(t, dec.final_block.weight, dec.final_block.bias, cop=self._final_cop(), prep=store.conv[id(dec.final_block)])
        out = ops.final_softmax(z, C)
        if self.training:
            return _cl(out), [_cl(m) for m in masks]
        return _cl(ops.onehot_argmax(out.detach()))


Model_Dict = {'MaskTransUnet': MaskTransUnet}


def get_model_dict(name: str):
    """Same contract as model/trans_3DUnet.py:215-222.  Only MaskTransUnet exists: the reference's other
    four registry entries cannot run."""
    return Model_Dict[name]

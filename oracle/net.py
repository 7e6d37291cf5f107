"""Oracle: functional fp32 restatement of MaskTransUnet (test infrastructure).

The network is a pure function of a flat parameter dict keyed by the reference's
`state_dict` names, so fixtures can be generated from a
seed and no module tree is needed.  Layout is the reference's [B, C, H, W, D].

Restates (paths relative to the reference root):
  model/trans_3DUnet.py:161-204   MaskTransUnet ctor/forward (train vs eval return)
  model/Unet_3Dblock.py:123-152   2x2 window (un)embedding
  model/Unet_3Dblock.py:290-341   encoder residual/strided block
  model/Unet_3Dblock.py:560-607   encoder
  model/Unet_3Dblock.py:194-221   attention gate
  model/Unet_3Dblock.py:224-274   bottleneck transformer
  model/Unet_3Dblock.py:343-501   token embed / transformer / un-embed of an ROI
  model/Unet_3Dblock.py:504-557   decoder block
  model/Unet_3Dblock.py:673-755   ROI bridge
  model/Unet_3Dblock.py:1277-1396 decoder
  model/trans_block.py:41-67      linear attention (softmax feature maps)
  model/trans_block.py:70-96      depthwise-conv positional embedding
  model/trans_block.py:127-211    multi-head wrapper and post-norm layer
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import roi as _roi

HEAD_DIM = 32      # Unet_3Dblock.py:1294 (nhead_lens) and 1313-1314
N_LAYERS = 8       # Unet_3Dblock.py:1294 (N)
LN_EPS = 1e-6      # trans_block.py:182
IN_EPS = 1e-5      # nn.InstanceNorm3d default
LRELU = 0.01       # nn.LeakyReLU default


@dataclass
class NetConfig:
    """Constructor arguments of MaskTransUnet (trans_3DUnet.py:161-162, defaults train3D.py:54-61)."""
    num_layers: List[int] = field(default_factory=lambda: [16, 32, 64, 128, 256])
    roi_size_list: List[int] = field(default_factory=lambda: [100, 65, 40, 25, 10])
    is_roi_list: List[bool] = field(default_factory=lambda: [False, True, True, True, True])
    dim_input: int = 1
    dim_output: int = 2
    dropout: float = 0.0   # 0 = the deterministic recipe used for parity; 0.3 = training


def param_shapes(cfg: NetConfig) -> Dict[str, tuple]:
    """Every state_dict key with its shape, in registration order."""
    L, C = cfg.num_layers, cfg.dim_output
    nl = len(L)
    out: Dict[str, tuple] = {}

    def conv(name, co, ci, k=3):
        out[name + '.weight'] = (co, ci, k, k, k)
        out[name + '.bias'] = (co,)

    def lin(name, o, i):
        out[name + '.weight'] = (o, i)
        out[name + '.bias'] = (o,)

    def layers(prefix, d):
        for n in range(N_LAYERS):
            p = f'{prefix}.layers.{n}'
            for j in range(4):
                lin(f'{p}.self_attn.linears.{j}', d, d)
            lin(f'{p}.linear1', 2 * d, d)
            lin(f'{p}.linear2', d, 2 * d)
            for j in (1, 2):
                out[f'{p}.layer_norm{j}.weight'] = (d,)
                out[f'{p}.layer_norm{j}.bias'] = (d,)

    for i in range(1, nl):
        conv(f'encode.block_list.{i-1}.conv1', L[i-1], L[i-1])
        conv(f'encode.block_list.{i-1}.conv2', L[i], L[i-1])
    conv('encode.input_block', L[0], cfg.dim_input * 4)

    for i in range(nl - 1):
        if not cfg.is_roi_list[i]:
            continue
        d = min(4 * L[i], 256)
        t = f'decode.bridge_list.{i}.transformer'
        conv(f'{t}.down_embed.module_list.0.0', d, L[i])
        conv(f'{t}.up_embed.module_list.0.1', L[i], d)
        out[f'{t}.pos_encoder.proj.weight'] = (d, 1, 3, 3, 3)
        out[f'{t}.pos_encoder.proj.bias'] = (d,)
        layers(t, d)
    t = f'decode.bridge_list.{nl-1}.transformer'
    for n in range(N_LAYERS):
        out[f'{t}.pos_encoders.{n}.proj.weight'] = (L[-1], 1, 3, 3, 3)
        out[f'{t}.pos_encoders.{n}.proj.bias'] = (L[-1],)
    layers(t, L[-1])
    for i in range(1, nl):
        conv(f'decode.mask_conv_list.{i-1}', C, L[i])
    for i in range(1, nl):
        a = f'decode.att_conv_list.{i-1}'
        conv(f'{a}.W_x.0', L[i-1], L[i-1], 1)
        conv(f'{a}.W_g.0', L[i-1], L[i], 1)
        conv(f'{a}.psi.0', 1, L[i-1], 1)
    for i in range(1, nl):
        conv(f'decode.block_list.{i-1}.conv1', L[-i-1], L[-i])
        conv(f'decode.block_list.{i-1}.conv2', L[-i-1], 2 * L[-i-1])
    conv('decode.final_block', 4 * C, L[0])
    return out


# ----------------------------------------------------------------------------- small pieces

def _drop(x, p, channelwise=False):
    """nn.Dropout / nn.Dropout3d in training mode; p == 0 is the parity recipe (identity)."""
    if p <= 0:
        return x
    return F.dropout3d(x, p, True) if channelwise else F.dropout(x, p, True)


def window_embed(x, k=2):
    """[B,1,H,W,D] -> [B,k*k,H/k,W/k,D], channel = kh*k+kw (Unet_3Dblock.py:123-136)."""
    B, _, H, W, D = x.shape
    return x.reshape(B, H // k, k, W // k, k, D).permute(0, 2, 4, 1, 3, 5).reshape(B, k * k, H // k, W // k, D)


def window_unembed(x, k=2):
    """[B,c*k*k,h,w,D] -> [B,c,h*k,w*k,D], channel c*k*k+kh*k+kw -> (c, h*k+kh, w*k+kw) (Unet_3Dblock.py:138-152)."""
    B, ch, h, w, D = x.shape
    x = x.reshape(B, ch // (k * k), k, k, h, w, D).permute(0, 1, 4, 2, 5, 3, 6)
    return x.reshape(B, ch // (k * k), h * k, w * k, D)


def _conv(P, name, x, stride=1, padding=1):
    return F.conv3d(x, P[name + '.weight'], P[name + '.bias'], stride=stride, padding=padding)


def _in_act(x):
    """InstanceNorm3d (no affine, biased var, eps 1e-5) then LeakyReLU(0.01)."""
    return F.leaky_relu(F.instance_norm(x, eps=IN_EPS), LRELU)


def linear_attention(q, k, v):
    """Softmax-feature linear attention (trans_block.py:41-67): [B,h,N,dk] x3 -> [B,h,N,dk].

    q: softmax over dk, scaled 1/sqrt(dk); k: softmax over the TOKEN axis; ctx = k^T v; out = q ctx.
    """
    dk = q.shape[-1]
    qs = torch.softmax(q, dim=-1) / math.sqrt(dk)
    ks = torch.softmax(k, dim=-2)
    ctx = torch.einsum('bhnd,bhne->bhde', ks, v)
    return torch.einsum('bhnd,bhde->bhne', qs, ctx)


def attn_layer(P, pre, x, p_drop=0.0):
    """Post-norm transformer layer on tokens [B,N,d] (trans_block.py:148-166, 203-211)."""
    B, N, d = x.shape
    h = d // HEAD_DIM

    def proj(j, t):
        return F.linear(t, P[f'{pre}.self_attn.linears.{j}.weight'], P[f'{pre}.self_attn.linears.{j}.bias'])

    q, k, v = (proj(j, x).view(B, N, h, HEAD_DIM).transpose(1, 2) for j in range(3))
    if p_drop > 0:                     # trans_block.py:62-63: dropout result is discarded, only RNG is consumed
        F.dropout(q, p_drop, True)
    a = linear_attention(q, k, v).transpose(1, 2).reshape(B, N, d)
    a = proj(3, a)
    x = F.layer_norm(x + _drop(a, p_drop), (d,), P[f'{pre}.layer_norm1.weight'], P[f'{pre}.layer_norm1.bias'], LN_EPS)
    f = F.linear(x, P[f'{pre}.linear1.weight'], P[f'{pre}.linear1.bias'])
    f = F.linear(_drop(F.gelu(f), p_drop), P[f'{pre}.linear2.weight'], P[f'{pre}.linear2.bias'])
    return F.layer_norm(x + _drop(f, p_drop), (d,), P[f'{pre}.layer_norm2.weight'], P[f'{pre}.layer_norm2.bias'], LN_EPS)


def token_transformer(P, pre, pos_name, x, p_drop=0.0):
    """8 layers over the voxels of x [B,d,H,W,D] as tokens, depthwise positional conv after layer 0.

    Token order is depth-major (Unet_3Dblock.py:259-264 / 480-482); the positional conv therefore
    runs on a [B,d,D,H,W] grid (Unet_3Dblock.py:267-270 / 487-490, trans_block.py:94-96).
    """
    B, d, H, W, D = x.shape
    t = x.permute(0, 4, 2, 3, 1).reshape(B, D * H * W, d)
    for n in range(N_LAYERS):
        t = attn_layer(P, f'{pre}.layers.{n}', t, p_drop)
        if n == 0:
            g = t.transpose(1, 2).reshape(B, d, D, H, W)
            pe = F.conv3d(g, P[pos_name + '.proj.weight'], P[pos_name + '.proj.bias'], padding=1, groups=d)
            g = _drop(g + pe, p_drop, channelwise=True)
            t = g.flatten(2).transpose(1, 2)
    return t.transpose(1, 2).reshape(B, d, D, H, W).permute(0, 1, 3, 4, 2)


def roi_transformer(P, pre, x, p_drop=0.0):
    """Strided-conv token embed -> transformer -> nearest x2 + conv un-embed (Unet_3Dblock.py:469-501)."""
    e = _drop(_in_act(_conv(P, f'{pre}.down_embed.module_list.0.0', x, stride=2)), p_drop)
    e = token_transformer(P, pre, f'{pre}.pos_encoder', e, p_drop)
    e = F.interpolate(e, scale_factor=2)             # nn.Upsample default mode = nearest
    return _drop(_in_act(_conv(P, f'{pre}.up_embed.module_list.0.1', e)), p_drop)


def roi_bridge(P, pre, skip, fg, roi_size, p_drop=0.0, boxes_out=None):
    """Box from fg >= 0.5, warp skip to the fixed grid, transform, warp back; no residual (Unet_3Dblock.py:717-755)."""
    geo = _roi.roi_geometry(roi_size)
    with torch.no_grad():
        box = _roi.find_boxes(fg >= 0.5, geo['min_h'], geo['min_w'])
    if boxes_out is not None:
        boxes_out.append(box)
    grid = _roi.warp_to_roi(skip, box, geo)
    grid = roi_transformer(P, f'{pre}.transformer', grid, p_drop)
    return _roi.warp_from_roi(skip, grid, box, geo)


def attention_gate(P, pre, skip, up):
    """sigmoid(psi(relu(IN(Wx skip) + IN(Wg up)))) -> [B,1,H,W,D] (Unet_3Dblock.py:217-221)."""
    a = F.instance_norm(_conv(P, f'{pre}.W_x.0', skip, padding=0), eps=IN_EPS)
    b = F.instance_norm(_conv(P, f'{pre}.W_g.0', up, padding=0), eps=IN_EPS)
    return torch.sigmoid(_conv(P, f'{pre}.psi.0', F.relu(a + b), padding=0))


def encoder(P, cfg: NetConfig, x):
    """Stem + 4 x [residual conv block, strided conv] (Unet_3Dblock.py:596-607, 325-341)."""
    L = cfg.num_layers
    x = _in_act(_conv(P, 'encode.input_block', window_embed(x)))
    skips = []
    for i in range(len(L) - 1):
        pre = f'encode.block_list.{i}'
        s = _in_act(_conv(P, pre + '.conv1', x)) + x
        stride = (2, 2, i % 2 + 1)                  # Unet_3Dblock.py:584 with i shifted by one
        x = _drop(_in_act(_conv(P, pre + '.conv2', s, stride=stride)), cfg.dropout)
        skips.append(s)
    return x, skips


def decoder(P, cfg: NetConfig, x, skips, boxes_out=None):
    """Bottleneck transformer, then coarse->fine decoding with mask heads, gates and ROI bridges (Unet_3Dblock.py:1359-1396)."""
    L = cfg.num_layers
    nl = len(L)
    masks = []
    pre = f'decode.bridge_list.{nl-1}.transformer'
    x = token_transformer(P, pre, f'{pre}.pos_encoders.0', x, cfg.dropout)
    for i in range(1, nl):
        scale = (2, 2, 2) if (nl - i) % 2 == 0 else (2, 2, 1)
        x = F.interpolate(x, scale_factor=scale, mode='trilinear', align_corners=True)
        m = torch.softmax(_conv(P, f'decode.mask_conv_list.{nl-1-i}', x), dim=1)
        masks.append(m)
        skip = skips[-i]
        skip = skip * attention_gate(P, f'decode.att_conv_list.{nl-1-i}', skip, x)
        lvl = nl - 1 - i
        if cfg.is_roi_list[lvl]:
            fg = (1 - m[:, 0]).unsqueeze(1)
            skip = roi_bridge(P, f'decode.bridge_list.{lvl}', skip, fg, cfg.roi_size_list[lvl], cfg.dropout, boxes_out)
        b = f'decode.block_list.{i-1}'
        x = _in_act(_conv(P, b + '.conv1', x))
        x = _in_act(_conv(P, b + '.conv2', torch.cat((x, skip), dim=1)))
        x = _drop(x, cfg.dropout)
    x = window_unembed(_conv(P, 'decode.final_block', x))
    return torch.softmax(x, dim=1), masks


def forward(P: Dict[str, torch.Tensor], cfg: NetConfig, x: torch.Tensor, training: bool = True, boxes_out=None):
    """MaskTransUnet.forward (trans_3DUnet.py:181-202): (probs, mask_list) when training, else one-hot argmax."""
    bottom, skips = encoder(P, cfg, x)
    out, masks = decoder(P, cfg, bottom, skips, boxes_out)
    if training:
        return out, masks
    idx = torch.argmax(out, dim=1, keepdim=True)
    return torch.zeros_like(out).scatter_(1, idx, 1)

"""ORACLE (test infrastructure, not product code).

CPU restatement, in plain functional PyTorch fp32, of the three networks on RapidDoc's page hot path
whose definitions are readable in the reference:

* PP-OCRv6 det small  = PPLCNetV4(det) + RepLKFPN + DBHead(ppocrv6)
* PP-OCRv6 rec small  = PPLCNetV4(rec) + LightSVTR + Linear(120 -> n_classes)  (raw CTC logits)
* PPHGNetV2-B4(det=True) = the PP-DocLayout-L/plus-L/V2/V3 backbone

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product path (``rapiddoc_amd``) never does: it fails loudly when the HIP library is absent.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against golden vectors
minted by importing the reference's own ``nn.Module`` definitions in the build container
(``tests/golden/make_golden.py``); the reference itself holds no numeric vectors for this path
(SURVEY.md fact 5).

All functions take a ``state`` mapping using the reference's state-dict names (the names found in the
shipped ``.safetensors`` after the ``model.`` prefix is stripped) and NCHW float32 input.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5  # nn.BatchNorm2d default, reference rec_lcnetv4.py:109
LN_EPS = 1e-6  # reference necks/rnn.py:306-308,363


def as_torch_state(state: Mapping[str, object]) -> Dict[str, Tensor]:
    out = {}
    for k, v in state.items():
        out[k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(__import__("numpy").ascontiguousarray(v))
    return out


def _bn(state, prefix: str, x: Tensor) -> Tensor:
    return F.batch_norm(
        x,
        state[prefix + ".running_mean"],
        state[prefix + ".running_var"],
        state[prefix + ".weight"],
        state[prefix + ".bias"],
        training=False,
        eps=BN_EPS,
    )


# --------------------------------------------------------------------------------------------------
# PPLCNetV4  (reference rec_lcnetv4.py)
# --------------------------------------------------------------------------------------------------
def _lc_conv_bn_act(state, p: str, x: Tensor, stride=1, groups=1, act="relu") -> Tensor:
    """PPLCNetV4ConvLayer, rec_lcnetv4.py:87-118: conv(pad=(k-1)//2, no bias) -> BN -> act."""
    w = state[p + ".convolution.weight"]
    k = w.shape[-1]
    x = F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2, groups=groups)
    x = _bn(state, p + ".normalization", x)
    if act == "relu":
        x = F.relu(x)
    return x


def _lc_se(state, p: str, x: Tensor) -> Tensor:
    """PPLCNetV4SqueezeExcitationModule, rec_lcnetv4.py:120-142 (torch Hardsigmoid = clamp(x/6+.5,0,1))."""
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.relu(F.conv2d(s, state[p + ".convolutions.0.weight"], state[p + ".convolutions.0.bias"]))
    s = F.hardsigmoid(F.conv2d(s, state[p + ".convolutions.2.weight"], state[p + ".convolutions.2.bias"]))
    return x * s


def _lc_stem(state, p: str, x: Tensor) -> Tensor:
    """PPLCNetV4LargeStem.forward, rec_lcnetv4.py:158-169."""
    e = _lc_conv_bn_act(state, p + ".stem1", x, stride=2)
    e = F.pad(e, (0, 1, 0, 1))
    a = _lc_conv_bn_act(state, p + ".stem2a", e)  # k=2, pad 0
    a = F.pad(a, (0, 1, 0, 1))
    a = _lc_conv_bn_act(state, p + ".stem2b", a)
    pooled = F.max_pool2d(e, kernel_size=2, stride=1, ceil_mode=True)
    e = torch.cat([pooled, a], dim=1)
    e = _lc_conv_bn_act(state, p + ".stem3", e, stride=2)
    e = _lc_conv_bn_act(state, p + ".stem4", e)
    return e


def _lc_block(state, p: str, x: Tensor, cin: int, cout: int, stride, use_se: bool) -> Tensor:
    """PPLCNetV4DepthwiseSeparableConvLayer.forward, rec_lcnetv4.py:226-236."""
    stride_t = tuple(stride) if isinstance(stride, (list, tuple)) else (stride, stride)
    rep = stride_t == (1, 1) and cin == cout
    if rep:
        w = state[p + ".token_conv.weight"]
        x = F.conv2d(x, w, state[p + ".token_conv.bias"], stride=1, padding=w.shape[-1] // 2, groups=cin)
    else:
        x = _lc_conv_bn_act(state, p + ".token_conv", x, stride=stride_t, groups=cin, act=None)
    if use_se:
        x = _lc_se(state, p + ".token_squeeze_excitation", x)
    res = x
    x = _lc_conv_bn_act(state, p + ".channel_conv1", x, act=None)
    x = F.gelu(x)  # erf form
    x = _lc_conv_bn_act(state, p + ".channel_conv2", x, act=None)
    if rep:
        x = res + x
    return x


# (kernel, cin, cout, stride, use_se) per block - reference rec_lcnetv4.py:7-43
LCNETV4_DET_SMALL = [
    [[3, 48, 48, 1, True], [3, 48, 48, 1, False]],
    [[3, 48, 96, 2, False], [3, 96, 96, 1, True], [3, 96, 96, 1, False]],
    [[3, 96, 192, 2, False], [3, 192, 192, 1, True], [3, 192, 192, 1, False], [3, 192, 192, 1, True],
     [3, 192, 192, 1, False]],
    [[3, 192, 384, 2, False], [3, 384, 384, 1, True], [3, 384, 384, 1, False]],
]
LCNETV4_REC_SMALL = [
    [[3, 96, 96, 1, True]],
    [[3, 96, 96, 1, False], [3, 96, 96, 1, False]],
    [[3, 96, 192, (2, 1), False], [3, 192, 192, 1, True], [3, 192, 192, 1, False], [3, 192, 192, 1, True],
     [3, 192, 192, 1, False], [3, 192, 192, 1, True], [3, 192, 192, 1, False]],
    [[3, 192, 384, (2, 1), False], [3, 384, 384, 1, True], [3, 384, 384, 1, False]],
]


def lcnetv4_features(state, x: Tensor, cfg, prefix="backbone.encoder") -> List[Tensor]:
    h = _lc_stem(state, prefix + ".convolution", x)
    feats = []
    for si, stage in enumerate(cfg):
        for bi, (k, cin, cout, stride, se) in enumerate(stage):
            h = _lc_block(state, f"{prefix}.blocks.{si}.blocks.{bi}", h, cin, cout, stride, se)
        feats.append(h)
    return feats


# --------------------------------------------------------------------------------------------------
# PP-OCRv6 det: RepLKFPN (db_fpn.py:288-415) + DBHead ppocrv6 (det_db_head.py:95-149)
# --------------------------------------------------------------------------------------------------
def _fpn_se(state, p: str, x: Tensor) -> Tensor:
    """RepLKFPNSqueezeExcitationModule, db_fpn.py:288-308 (Paddle hard-sigmoid: clamp(.2x+.5,0,1))."""
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.conv2d(F.relu(F.conv2d(s, state[p + ".conv1.weight"], state[p + ".conv1.bias"])),
                 state[p + ".conv2.weight"], state[p + ".conv2.bias"])
    s = torch.clamp(0.2 * s + 0.5, 0.0, 1.0)
    return x * s


def replkfpn(state, feats: List[Tensor], prefix="neck") -> Tensor:
    fused = []
    for i, f in enumerate(feats):
        y = F.conv2d(f, state[f"{prefix}.insert_conv.{i}.in_conv.weight"])
        y = y + _fpn_se(state, f"{prefix}.insert_conv.{i}.squeeze_excitation_block", y)
        fused.append(y)
    for i in range(2, -1, -1):
        fused[i] = fused[i] + F.interpolate(fused[i + 1], scale_factor=2, mode="nearest")
    outs = []
    for i, f in enumerate(fused):
        p = f"{prefix}.input_conv.{i}"
        w = state[p + ".depthwise_convolution.weight"]
        y = F.conv2d(f, w, state[p + ".depthwise_convolution.bias"], padding=w.shape[-1] // 2, groups=w.shape[0])
        y = F.conv2d(y, state[p + ".pointwise_convolution.weight"])
        y = y + _fpn_se(state, p + ".squeeze_excitation_module", y)
        outs.append(y)
    proc = [outs[0]] + [F.interpolate(outs[i], scale_factor=2 ** i, mode="nearest") for i in (1, 2, 3)]
    return torch.cat(proc[::-1], dim=1)


def dbhead_v6(state, x: Tensor, prefix="head") -> Tensor:
    y = F.conv2d(x, state[prefix + ".conv_down.convolution.weight"], None, padding=1)
    y = F.relu(_bn(state, prefix + ".conv_down.norm", y))
    y = F.conv_transpose2d(y, state[prefix + ".conv_up.convolution.weight"],
                           state[prefix + ".conv_up.convolution.bias"], stride=2)
    y = F.relu(_bn(state, prefix + ".conv_up.norm", y))
    y = F.conv_transpose2d(y, state[prefix + ".conv_final.weight"], state[prefix + ".conv_final.bias"], stride=2)
    return torch.nan_to_num(torch.sigmoid(y))


def det_forward(state, x: Tensor, return_all: bool = False):
    """PP-OCRv6 det small: [B,3,H,W] (H,W multiples of 32) -> prob map [B,1,H,W]."""
    feats = lcnetv4_features(state, x, LCNETV4_DET_SMALL)
    neck = replkfpn(state, feats)
    maps = dbhead_v6(state, neck)
    if return_all:
        return {"feats": feats, "neck": neck, "maps": maps}
    return maps


# --------------------------------------------------------------------------------------------------
# PP-OCRv6 rec: PPLCNetV4(rec) + LightSVTR (necks/rnn.py:238-379) + Linear (rec_multi_head.py:66-75)
# --------------------------------------------------------------------------------------------------
def _svtr_conv(state, p: str, x: Tensor, padding=0, groups=1) -> Tensor:
    x = F.conv2d(x, state[p + ".convolution.weight"], None, padding=padding, groups=groups)
    return F.silu(_bn(state, p + ".normalization", x))


def lightsvtr(state, x: Tensor, prefix="head.encoder", num_heads=8) -> Tensor:
    """EncoderWithLightSVTR.forward, necks/rnn.py:366-379. x: [B,384,1,T] -> [B,120,1,T]."""
    res = _svtr_conv(state, prefix + ".conv_block.0", x)
    h = _svtr_conv(state, prefix + ".conv_block.1", x)
    wk = state[prefix + ".conv_block.2.convolution.weight"]
    h = h + _svtr_conv(state, prefix + ".conv_block.2", h, padding=(0, wk.shape[-1] // 2), groups=wk.shape[0])
    b, c, hh, ww = h.shape
    t = h.flatten(2).permute(0, 2, 1)  # [B,T,C]
    depth = 0
    while f"{prefix}.svtr_block.{depth}.layer_norm1.weight" in state:
        depth += 1
    for d in range(depth):
        p = f"{prefix}.svtr_block.{d}"
        y = F.layer_norm(t, (c,), state[p + ".layer_norm1.weight"], state[p + ".layer_norm1.bias"], LN_EPS)
        qkv = F.linear(y, state[p + ".self_attn.qkv.weight"], state[p + ".self_attn.qkv.bias"])
        hd = c // num_heads
        qkv = qkv.reshape(b, -1, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = torch.matmul(att, v).transpose(1, 2).reshape(b, -1, c)
        o = F.linear(o, state[p + ".self_attn.projection.weight"], state[p + ".self_attn.projection.bias"])
        t = t + o
        y = F.layer_norm(t, (c,), state[p + ".layer_norm2.weight"], state[p + ".layer_norm2.bias"], LN_EPS)
        y = F.silu(F.linear(y, state[p + ".mlp.fc1.weight"], state[p + ".mlp.fc1.bias"]))
        y = F.linear(y, state[p + ".mlp.fc2.weight"], state[p + ".mlp.fc2.bias"])
        t = t + y
    t = F.layer_norm(t, (c,), state[prefix + ".norm.weight"], state[prefix + ".norm.bias"], LN_EPS)
    h = t.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return h + res


def rec_forward(state, x: Tensor, return_all: bool = False):
    """PP-OCRv6 rec small: [B,3,48,W] -> raw CTC logits [B, W/8, n_classes]."""
    feats = lcnetv4_features(state, x, LCNETV4_REC_SMALL)
    f = F.avg_pool2d(feats[-1], [3, 2])  # rec_lcnetv4.py:311
    n = lightsvtr(state, f)
    seq = n.squeeze(2).permute(0, 2, 1)
    logits = F.linear(seq, state["head.head.weight"], state["head.head.bias"])
    if return_all:
        return {"backbone": f, "neck": n, "logits": logits}
    return logits


def ctc_greedy_stats(logits: Tensor):
    """What the host decode needs per time step: argmax class and its softmax probability.

    Reference: session wrapper applies softmax over classes (ocr/torch.py:186-187), rapidocr
    CTCLabelDecode takes argmax / max of it.
    """
    prob = torch.softmax(logits, dim=2)
    p, idx = prob.max(dim=2)
    return idx.to(torch.int32), p


# --------------------------------------------------------------------------------------------------
# PPHGNetV2 (reference formula/.../rec_pphgnetv2.py:860-1360)
# --------------------------------------------------------------------------------------------------
def _hg_conv_bn_act(state, p: str, x: Tensor, stride=1, groups=1, act=True, same=False) -> Tensor:
    """ConvBNAct, rec_pphgnetv2.py:860-918. padding='same' (stem 2x2) pads right/bottom only."""
    w = state[p + ".conv.weight"]
    k = w.shape[-1]
    if same:
        tot = k - 1
        lo = tot // 2
        x = F.pad(x, (lo, tot - lo, lo, tot - lo))
        x = F.conv2d(x, w, None, stride=stride, groups=groups)
    else:
        x = F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2, groups=groups)
    x = _bn(state, p + ".bn", x)
    return F.relu(x) if act else x


def _hg_stem(state, p: str, x: Tensor) -> Tensor:
    """StemBlock.forward, rec_pphgnetv2.py:1045-1056 (max-pool k2 s1 with zero 'same' pad, :962-976)."""
    x = _hg_conv_bn_act(state, p + ".stem1", x, stride=2)
    x2 = _hg_conv_bn_act(state, p + ".stem2a", x, same=True)
    x2 = _hg_conv_bn_act(state, p + ".stem2b", x2, same=True)
    x1 = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), kernel_size=2, stride=1, ceil_mode=True)
    x = torch.cat([x1, x2], dim=1)
    x = _hg_conv_bn_act(state, p + ".stem3", x, stride=2)
    return _hg_conv_bn_act(state, p + ".stem4", x)


def _hg_block(state, p: str, x: Tensor, layer_num: int, light: bool, identity: bool) -> Tensor:
    """HGV2_Block.forward, rec_pphgnetv2.py:1124-1136."""
    ident = x
    outs = [x]
    for i in range(layer_num):
        lp = f"{p}.layers.{i}"
        if light:
            x = _hg_conv_bn_act(state, lp + ".conv1", x, act=False)
            x = _hg_conv_bn_act(state, lp + ".conv2", x, groups=x.shape[1], act=True)
        else:
            x = _hg_conv_bn_act(state, lp, x)
        outs.append(x)
    x = torch.cat(outs, dim=1)
    x = _hg_conv_bn_act(state, p + ".aggregation_squeeze_conv", x)
    x = _hg_conv_bn_act(state, p + ".aggregation_excitation_conv", x)
    return x + ident if identity else x


# in, mid, out, blocks, downsample, light, kernel, layers, stride  (rec_pphgnetv2.py:1463-1468 / 1601-1607)
HGV2_B4_DET = [
    [48, 48, 128, 1, False, False, 3, 6, 2],
    [128, 96, 512, 1, True, False, 3, 6, 2],
    [512, 192, 1024, 3, True, True, 5, 6, 2],
    [1024, 384, 2048, 1, True, True, 5, 6, 2],
]


def pphgnetv2_features(state, x: Tensor, cfg=HGV2_B4_DET, prefix="") -> List[Tensor]:
    """PPHGNetV2(det=True).forward -> 4 stage outputs (strides 4/8/16/32)."""
    x = _hg_stem(state, prefix + "stem", x)
    outs = []
    for si, (cin, mid, cout, nblk, down, light, k, nl, stride) in enumerate(cfg):
        sp = f"{prefix}stages.{si}"
        if down:
            x = _hg_conv_bn_act(state, sp + ".downsample", x, stride=stride, groups=cin, act=False)
        for bi in range(nblk):
            x = _hg_block(state, f"{sp}.blocks.{bi}", x, nl, light, identity=bi > 0)
        outs.append(x)
    return outs


HGV2_B6_FORMULA = [
    [96, 96, 192, 2, False, False, 3, 6, 2],
    [192, 192, 512, 3, True, False, 3, 6, 2],
    [512, 384, 1024, 6, True, True, 5, 6, 2],
    [1024, 768, 2048, 3, True, True, 5, 6, 2],
]


def formula_encoder_forward(state, x: Tensor, prefix="backbone.pphgnet_b6.") -> Tensor:
    """PPHGNetV2_B6_Formula.forward (rec_pphgnetv2.py:1616-1642): grey input repeated to 3 channels, last stage
    output [B,2048,h,w] -> [B, h*w, 2048]."""
    if x.shape[1] == 1:
        x = torch.repeat_interleave(x, repeats=3, dim=1)
    f = pphgnetv2_features(state, x, HGV2_B6_FORMULA, prefix)[-1]
    b, c, h, w = f.shape
    return f.reshape(b, c, h * w).permute(0, 2, 1)


# --------------------------------------------------------------------------------------------------
# algorithmic FLOPs (2*MAC, conv + linear) used by bench.py's roofline bookkeeping - SURVEY.md section 8d
# --------------------------------------------------------------------------------------------------
def conv_flops(cin, cout, kh, kw, oh, ow, groups=1, batch=1) -> int:
    return 2 * batch * oh * ow * cout * (cin // groups) * kh * kw

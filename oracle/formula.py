"""ORACLE (test infrastructure): CPU restatement of the PP-FormulaNet_plus decoder head in plain PyTorch fp32.

Reference: rapid_doc/model/formula/rapid_formula_self/networks/heads/rec_ppformulanet_head.py
  PPFormulaNet_Head.forward -> generate_export (:1054-1176), generate_single_iter (:919-962, enc_to_dec_proj 2048->512),
  CustomMBartDecoder.forward (:407-630) and rec_unimernet_head.py MBartAttention (:502-628), MBartDecoderLayer (:635-746),
  MBartLearnedPositionalEmbedding (offset 2, :440-456), ForcedEOSTokenLogitsProcessor(max_length 1537, :1545-1572).
Pinned by tests/golden/formula_seed0_*.npz, minted by running the reference's BaseModel (make_golden.py).
The restatement recomputes the whole prefix every step (no KV cache): slow, obviously correct.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

DEC = "head.decoder.model.decoder."
EOS, PAD, START = 2, 1, 0
FORCED_EOS_LEN = 1537
HEADS = 16


def _lin(state, p, x):
    return F.linear(x, state[p + ".weight"], state.get(p + ".bias"))


def _ln(state, p, x):
    return F.layer_norm(x, (x.shape[-1],), state[p + ".weight"], state[p + ".bias"], 1e-5)


def _attn(state, p, x, kv, causal):
    b, t, d = x.shape
    hd = d // HEADS
    q = (_lin(state, p + ".q_proj", x) * hd ** -0.5).reshape(b, t, HEADS, hd).transpose(1, 2)
    k = _lin(state, p + ".k_proj", kv).reshape(b, -1, HEADS, hd).transpose(1, 2)
    v = _lin(state, p + ".v_proj", kv).reshape(b, -1, HEADS, hd).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if causal:
        w = w + torch.full((t, t), float("-inf")).triu(1)
    o = (torch.softmax(w, dim=-1) @ v).transpose(1, 2).reshape(b, t, d)
    return _lin(state, p + ".out_proj", o)


def decoder_logits(state, enc_proj, ids, all_positions=False):
    """Logits of the LAST position for token prefix `ids` [B,L] (all positions [B,L,V] with `all_positions`: the causal
    mask makes position t's logits those of the prefix ids[:, :t+1] - teacher forcing, one pass instead of L)."""
    d = state[DEC + "embed_tokens.weight"].shape[1]
    L = ids.shape[1]
    x = state[DEC + "embed_tokens.weight"][ids] * math.sqrt(d) + state[DEC + "embed_positions.weight"][torch.arange(L) + 2]
    x = _ln(state, DEC + "layernorm_embedding", x)
    n_layers = 0
    while f"{DEC}layers.{n_layers}.fc1.weight" in state:
        n_layers += 1
    for l in range(n_layers):
        p = f"{DEC}layers.{l}"
        h = _ln(state, p + ".self_attn_layer_norm", x)
        x = x + _attn(state, p + ".self_attn", h, h, True)
        h = _ln(state, p + ".encoder_attn_layer_norm", x)
        x = x + _attn(state, p + ".encoder_attn", h, enc_proj, False)
        h = _ln(state, p + ".final_layer_norm", x)
        x = x + _lin(state, p + ".fc2", F.gelu(_lin(state, p + ".fc1", h)))
    x = _ln(state, DEC + "layer_norm", x)
    return F.linear(x if all_positions else x[:, -1], state["head.decoder.lm_head.weight"])


def teacher_forced_logits(state, enc, ids):
    """Logits [B, L-1, V] the decoder produces at every step when it is fed the prefix of `ids` [B,L] (start token included):
    row t is the distribution the (t+1)-th token was chosen from.  Forced-EOS rule not applied (sequences shorter than 1536)."""
    enc_proj = _lin(state, "head.enc_to_dec_proj", enc)
    return decoder_logits(state, enc_proj, ids[:, :-1], all_positions=True)


def formula_decode(state, enc, max_new_tokens, return_logits=False):
    """enc: encoder states [B,S,2048] -> token ids [B,L] int64 (start token included), like generate_export."""
    enc_proj = _lin(state, "head.enc_to_dec_proj", enc)
    B = enc.shape[0]
    ids = torch.full((B, 1), START, dtype=torch.int64)
    unfinished = torch.ones(B, dtype=torch.int64)
    all_logits = []
    for _ in range(max_new_tokens):
        lg = decoder_logits(state, enc_proj, ids)
        if ids.shape[1] == FORCED_EOS_LEN - 1:
            forced = torch.full_like(lg, -math.inf)
            forced[:, EOS] = 0
            lg = forced
        all_logits.append(lg)
        nxt = lg.argmax(-1) * unfinished + PAD * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], 1)
        unfinished = unfinished & (nxt != EOS).to(torch.int64)
        if bool(((ids == EOS).cumsum(1)[:, -1] >= 1).all()):
            break
    return (ids, all_logits) if return_logits else ids

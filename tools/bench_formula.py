#!/usr/bin/env python3
"""PP-FormulaNet_plus-M on the GPU box: encoder TFLOP/s and decoder ms/step (developer tool, synthetic weights)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from rapiddoc_amd import weights as W
from rapiddoc_amd.engine import RdEngine

man = W.load_manifest(ROOT / "tests/golden/manifest_ppformulanet_plus_m_m8.json")
man = [(n, (2562, 512) if n.endswith("embed_positions.weight") else s, d) for n, s, d in man]
st = W.synth_state_dict(man, 0)
enc_eng = RdEngine("pphgnetv2_b6_formula").load_weights({k: v for k, v in st.items() if k.startswith("backbone.")})
dec_eng = RdEngine("ppformulanet_head").load_weights({k: v for k, v in st.items() if k.startswith("head.")})
BATCHES = [int(a) for a in sys.argv[1:]] or [1, 8, 32]
for B in BATCHES:
    x = torch.rand((B, 1, 384, 384), device="cuda") * 2 - 1
    enc = enc_eng.formula_encoder_forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        enc = enc_eng.formula_encoder_forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"encoder B={B:3d} 384x384: {dt*1e3:8.2f} ms  {98.945e9*B/dt/1e12:6.1f} TFLOP/s")
    enc_r = torch.randn((B, 144, 2048), device="cuda") * 3
    for n in (64,):
        dec_eng.formula_decode(enc_r, 8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = dec_eng.formula_decode(enc_r, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"decoder B={B:3d} S=144 {ids.shape[1]-1} steps: {dt*1e3:8.2f} ms  {dt*1e3/(ids.shape[1]-1):6.3f} ms/step  {B*(ids.shape[1]-1)/dt:9.0f} tok/s")

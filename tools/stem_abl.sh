#!/bin/bash
# Fused stem front: phase ablations (RD_STEM_DBG bits: 1 no stem1, 2 no stem2a, 4 no pool, 8 no stem2b, 16 no patch traffic; results
# garbage, timing only) on the recogniser's widest launch, the det / layout stems, and the kernel's parity tests.
cd $GRAFT_REPO_ROOT
for d in ${STEM_ABL:-0 1 2 8 16 15}; do
  echo "RD_STEM_DBG=$d: $(RD_STEM_DBG=$d timeout 120 python tools/op_profile.py ppocrv6_rec 140 48 1248 60 2>/dev/null | grep stem_fused | cut -c1-70)"
done
echo "det: $(timeout 120 python tools/op_profile.py ppocrv6_det 32 960 704 80 2>/dev/null | grep stem_fused | cut -c1-70)"
echo "layout: $(timeout 120 python tools/op_profile.py pphgnetv2_b4 32 800 800 80 2>/dev/null | grep stem_fused | cut -c1-70)"
[ "${STEM_TESTS:-1}" = "1" ] && python -m pytest tests/test_gpu_round3.py -q -x -k "stem" 2>&1 | tail -3

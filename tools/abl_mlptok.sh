#!/bin/bash
# timing ablations of the token-stationary MLP kernel (tuning build must be in the tree): SPEI_TOK_DBG bits, see swin_tok16.hip
for d in 0 16 17 20; do
  echo "== SPEI_TOK_DBG=$d"
  SPEI_TOK_DBG=$d python tools/stamp_phases.py mlptok 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# Same-call A/B (runs on the GPU box): centre_norm as one fma (default) vs add + multiply (-DBT709_NO_FMA_CENTRE)
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2 3; do
  for v in fma nofma; do echo "== decode_lab $v (round $round)"; tools/bin/decode_lab_$v 0 5 | head -1; done
done
tools/ab_half.sh tools/bin/libbt709hip_nofma.so 3

#!/bin/bash
# A/B of the tile kernel at the two residual-epilogue c_proj sites (lab library): default (chooser -> persistent 4-wave 160 x 256) vs the 8-wave staggered 160 x 256 (mode 6)
# and 256 x 256 (mode 4), alternating on one box.   gpurun -- 'bash tools/diag/ab_resid_tiles.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--steps 30 --warmup 5 --regions 3 --no-cpu-baseline --no-sub-benches --no-roofline-pass"
run() { env CLIPCAP_HIP_LIB=lab "$@" python $R/bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
  echo "default: $(run X=1)"
  echo "PROJ=6 (attn c_proj on 8-wave 160x256): $(run CC_TILE_PROJ=6)"
  echo "PROJ2=6 (mlp c_proj on 8-wave 160x256): $(run CC_TILE_PROJ2=6)"
  echo "PROJ2=4 (mlp c_proj on 256x256): $(run CC_TILE_PROJ2=4)"
done

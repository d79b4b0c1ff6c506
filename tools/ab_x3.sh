#!/bin/bash
# A/B chain of the split-bf16 (fp32 parity) training step on ONE GPU box: each round-4 switch off, then everything on, and the bf16 step
# of the same box (run through gpurun from the repo root: `gpurun -- 'bash tools/ab_x3.sh'`).  Prints ms per step; two alternations.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env CLIPCAP_HIP_LIB=lab "$@" python $R/bench.py --precision 32 --steps 30 --warmup 3 --regions 3 --no-cpu-baseline --no-sub-benches --no-roofline-pass 2>/dev/null | tail -1 |
        python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
    echo "all on:               $(run CC_X3_IMG=1)"
    echo "CC_X3_SHARE=0:        $(run CC_X3_SHARE=0)"
    echo "CC_LM_EXPFORM=0:      $(run CC_LM_EXPFORM=0)"
    echo "CC_X3_FUSED=0:        $(run CC_X3_FUSED=0)"
    echo "CC_ATTN_X3MFMA=0:     $(run CC_ATTN_X3MFMA=0)"
    echo "CC_X3_IMG=0:          $(run CC_X3_IMG=0)"
    echo "all five off:         $(run CC_X3_SHARE=0 CC_LM_EXPFORM=0 CC_X3_FUSED=0 CC_ATTN_X3MFMA=0 CC_X3_IMG=0)"
done
echo "bf16 step, same box:  $(python $R/bench.py --steps 30 --warmup 3 --regions 3 --no-cpu-baseline --no-sub-benches --no-roofline-pass 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"

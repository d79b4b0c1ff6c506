#!/bin/bash
# A/B of a lab-build environment switch on ONE GPU box, alternating: tools/ab_env.sh <VAR> "<bench.py args>" <json key path> [rounds]
#   gpurun -- 'bash tools/ab_env.sh CC_SKINNY_80 "--mode decode --steps 3 --warmup 1 --regions 3" ms_per_step'
#   gpurun -- 'bash tools/ab_env.sh CC_X3_CHOOSE192 "--precision 32 --steps 30 --warmup 3 --regions 3 --no-cpu-baseline --no-sub-benches --no-roofline-pass" ms_per_step'
# The switches exist only in the lab library (CLIPCAP_HIP_LIB=lab; the product never reads the environment).
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; ARGS=$2; KEY=${3:-ms_per_step}; N=${4:-2}
run() { env CLIPCAP_HIP_LIB=lab "$@" python $R/bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['$KEY'])"; }
for i in $(seq 1 $N); do
    echo "$VAR=0: $(run $VAR=0)"
    echo "$VAR=1: $(run $VAR=1)"
done

#!/bin/bash
# kernel-trace statistics of one beam-5 decode (bench.py --mode decode), to gpurun_out/<tag>_decode.md; extra env passes through
TAG=${1:-dec}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/prof_$TAG -- python $ROOT/bench.py --mode decode --steps 1 --warmup 1 > $OUT/${TAG}_decode.log 2>&1
python $ROOT/tools/rocpd_stats.py $(find $OUT/prof_$TAG -name "*.db" | head -1) > $OUT/${TAG}_decode.md
rm -rf $OUT/prof_$TAG
cd $ROOT

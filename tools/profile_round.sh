#!/bin/bash
# Collects the per-round rocprofv3 evidence on a GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03     -> gpurun_out/<tag>_{train,mapper,decode,x3_train}.md (kernel-trace stats), <tag>_pmc_{FETCH,WRITE}_SIZE.txt and <tag>_decode_pmc_*.txt
# Counters are collected in their own passes (one TCC counter per pass; never combined with other trace domains).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TRAIN="python $ROOT/bench.py --steps 8 --warmup 2 --regions 1 --no-cpu-baseline --no-sub-benches"
run_trace() {  # name, command...
    local name=$1; shift
    rm -rf $OUT/prof_$name
    rocprofv3 --kernel-trace --output-format rocpd -d $OUT/prof_$name -- "$@" > $OUT/${TAG}_$name.log 2>&1
    python $ROOT/tools/rocpd_stats.py $(find $OUT/prof_$name -name "*.db" | head -1) > $OUT/${TAG}_$name.md
    rm -rf $OUT/prof_$name
}
run_trace train $TRAIN
run_trace mapper python $ROOT/bench.py --mode mapper --steps 10 --warmup 3
run_trace decode python $ROOT/bench.py --mode decode --steps 1 --warmup 1 --regions 1
# counter passes: one TCC counter per pass; the databases are kept until pmc_constants.py has read them
for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/prof_pmc_train_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format rocpd -d $OUT/prof_pmc_train_$ctr -- python $ROOT/bench.py --steps 2 --warmup 1 --regions 1 --no-cpu-baseline --no-sub-benches --no-roofline-pass > /dev/null 2>&1
    python $ROOT/tools/rocpd_pmc.py $(find $OUT/prof_pmc_train_$ctr -name "*.db" | head -1) > $OUT/${TAG}_pmc_$ctr.txt
done
# decode: the same two counter passes on the beam-5 decode (bench.py --mode decode --steps 1 --warmup 1 --regions 1 = 3 decodes: warm-up, timed, and the
# untimed one that counts the distinct KV rows; per generated position = totals / (3 decodes x 67 positions))
for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/prof_pmc_decode_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format rocpd -d $OUT/prof_pmc_decode_$ctr -- python $ROOT/bench.py --mode decode --steps 1 --warmup 1 --regions 1 > /dev/null 2>&1
    python $ROOT/tools/rocpd_pmc.py $(find $OUT/prof_pmc_decode_$ctr -name "*.db" | head -1) > $OUT/${TAG}_decode_pmc_$ctr.txt
done
python $ROOT/tools/pmc_constants.py $TAG $(find $OUT/prof_pmc_train_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/prof_pmc_train_WRITE_SIZE -name "*.db" | head -1) 3 \
    $(find $OUT/prof_pmc_decode_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/prof_pmc_decode_WRITE_SIZE -name "*.db" | head -1) 3 67 > $OUT/${TAG}_pmc_constants.json
cp $ROOT/profiles/pmc_constants.json $OUT/pmc_constants.json
rm -rf $OUT/prof_pmc_train_* $OUT/prof_pmc_decode_*
# the training step in the split-bf16 (fp32 parity) mode
run_trace x3_train python $ROOT/bench.py --precision 32 --steps 6 --warmup 2 --no-cpu-baseline --no-sub-benches --no-roofline-pass
# MFMA utilisation (north_star: "rocprof counters reporting ... MFMA utilisation"): SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in their own passes over the
# bf16 step, the split-bf16 step and one beam decode; tools/mfma_util.py -> <tag>_mfma_utilisation.md and the "mfma_busy" entry of pmc_constants.json
for leg in train x3_train decode; do
    case $leg in
        train) CMD="python $ROOT/bench.py --steps 2 --warmup 1 --regions 1 --no-cpu-baseline --no-sub-benches --no-roofline-pass" ;;
        x3_train) CMD="python $ROOT/bench.py --precision 32 --steps 2 --warmup 1 --regions 1 --no-cpu-baseline --no-sub-benches --no-roofline-pass" ;;
        decode) CMD="python $ROOT/bench.py --mode decode --steps 1 --warmup 1 --regions 1" ;;
    esac
    rm -rf $OUT/prof_mfma_$leg
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format rocpd -d $OUT/prof_mfma_$leg -- $CMD > /dev/null 2>&1
done
python $ROOT/tools/mfma_util.py train=$(find $OUT/prof_mfma_train -name "*.db" | head -1) x3_train=$(find $OUT/prof_mfma_x3_train -name "*.db" | head -1) \
    decode=$(find $OUT/prof_mfma_decode -name "*.db" | head -1) --json $ROOT/profiles/pmc_constants.json > $OUT/${TAG}_mfma_utilisation.md
cp $ROOT/profiles/pmc_constants.json $OUT/pmc_constants.json
rm -rf $OUT/prof_mfma_*
cd $ROOT

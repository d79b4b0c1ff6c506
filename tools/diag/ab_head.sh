R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--steps 30 --warmup 5 --regions 3 --no-cpu-baseline --no-sub-benches --no-roofline-pass"
run() { env "$@" python $R/bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
  echo "head: $(run CLIPCAP_HIP_LIB=$R/build_ab/libclipcap_hip_head.so)"
  echo "new : $(run X=1)"
done
for i in 1 2; do
  echo "lab default: $(run CLIPCAP_HIP_LIB=lab)"
  echo "lab DACT=7: $(run CLIPCAP_HIP_LIB=lab CC_TILE_DACT=7)"
  echo "lab FC=7: $(run CLIPCAP_HIP_LIB=lab CC_TILE_FC=7)"
  echo "lab PROJ2=7: $(run CLIPCAP_HIP_LIB=lab CC_TILE_PROJ2=7)"
done

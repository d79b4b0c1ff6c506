R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--steps 30 --warmup 5 --regions 3 --no-cpu-baseline --no-sub-benches --no-roofline-pass"
run() { env CLIPCAP_HIP_LIB=lab "$@" python $R/bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  echo "default: $(run X=1)"
  echo "DACT=5 (320x256): $(run CC_TILE_DACT=5)"
  echo "DACT=4 (256x256): $(run CC_TILE_DACT=4)"
  echo "DACT=3 (256x192): $(run CC_TILE_DACT=3)"
  echo "FC=4 (256x256): $(run CC_TILE_FC=4)"
  echo "FC=0 (128x128): $(run CC_TILE_FC=0)"
done

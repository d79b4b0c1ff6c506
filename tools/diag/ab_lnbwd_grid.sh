R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env CLIPCAP_HIP_LIB=lab "$@" python $R/bench.py --mode mapper --steps 30 --warmup 5 --regions 3 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for g in 256 128 160 192 224; do echo "CC_LNBWD_GRID=$g: $(run CC_LNBWD_GRID=$g)"; done
done

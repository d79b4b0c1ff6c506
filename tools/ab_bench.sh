# A/B of two builds on ONE GPU box (box-to-box spread is ~2 %): clipcap_amd/libclipcap_hip_old.so (e.g. built from `git archive HEAD`
# with `make OUT=.../libclipcap_hip_old.so`) against the current library; alternates them three times.
# usage: gpurun -- bash tools/ab_bench.sh ["--config 2" "--config 3" "--mode mapper" ...]
cd ${GRAFT_REPO_ROOT:-.}
cp clipcap_amd/libclipcap_hip.so /tmp/new.so; cp clipcap_amd/libclipcap_hip_old.so /tmp/old.so
[ $# -eq 0 ] && set -- "--config 2" "--config 3"
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so clipcap_amd/libclipcap_hip.so
    line="$v:"
    for a in "$@"; do line="$line  [$a] $(python bench.py $a --no-cpu-baseline --no-sub-benches --no-roofline-pass --steps 60 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)"; done
    echo "$line"
  done
done
cp /tmp/new.so clipcap_amd/libclipcap_hip.so

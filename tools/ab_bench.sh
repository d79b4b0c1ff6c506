# A/B of two builds on ONE GPU box (box-to-box spread is ~2 %): clipcap_amd/libclipcap_hip_old.so (e.g. built from `git archive HEAD`
# with `make OUT=.../libclipcap_hip_old.so`) against the current library; alternates them three times.  usage: gpurun -- bash tools/ab_bench.sh
cd $GRAFT_REPO_ROOT
cp clipcap_amd/libclipcap_hip.so /tmp/new.so; cp clipcap_amd/libclipcap_hip_old.so /tmp/old.so
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so clipcap_amd/libclipcap_hip.so
    echo "$v: $(python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')  cfg3 $(python bench.py --config 3 --no-cpu-baseline --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
  done
done
cp /tmp/new.so clipcap_amd/libclipcap_hip.so

#!/bin/bash
# Copies what tools/profile_round.sh left in gpurun_out/ into profiles/ (tracked) with a one-line provenance header per file.
# usage: tools/collect_profiles.sh r04
TAG=${1:-rXX}
G=gpurun_out
P=profiles
hdr() { echo "# round ${TAG#r}: $1"; echo; }
{ hdr "training step, config 2: \`rocprofv3 --kernel-trace -- python bench.py --steps 8 --warmup 2 --regions 1 --no-cpu-baseline --no-sub-benches\` (10 + 25 event-bracketed roofline steps in the trace)"; cat $G/${TAG}_train.md; } > $P/${TAG}_b_train_step_kernel_stats.md
{ hdr "mapper forward+backward only: \`rocprofv3 --kernel-trace -- python bench.py --mode mapper --steps 10 --warmup 3\` (B = 256; the at::native fill / normal / uniform kernels are the bench's one-time initialisation)"; cat $G/${TAG}_mapper.md; } > $P/${TAG}_c_mapper_fwd_bwd_kernel_stats.md
{ hdr "beam-5 decode, configs[4]: \`rocprofv3 --kernel-trace -- python bench.py --mode decode --steps 1 --warmup 1\` (3 decodes in the trace: warm-up, timed, and the untimed one that counts distinct KV rows; 66 single-position steps each; beam-group attention on, per-op launches)"; cat $G/${TAG}_decode.md; } > $P/${TAG}_d_decode_kernel_stats.md
{ hdr "\`rocprofv3 --kernel-trace --pmc FETCH_SIZE\` / \`--pmc WRITE_SIZE\` (one counter per pass) on \`python bench.py --steps 2 --warmup 1 --regions 1 --no-cpu-baseline --no-sub-benches --no-roofline-pass\` (3 steps); KB per dispatch; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section).  Totals and the source hash of the build: profiles/pmc_constants.json"; echo "## FETCH_SIZE"; echo '```'; cat $G/${TAG}_pmc_FETCH_SIZE.txt; echo '```'; echo; echo "## WRITE_SIZE"; echo '```'; cat $G/${TAG}_pmc_WRITE_SIZE.txt; echo '```'; } > $P/${TAG}_e_pmc_fetch_write_train.md
{ hdr "the same two counter passes on \`python bench.py --mode decode --steps 1 --warmup 1\` (3 decodes x 67 positions); KB per dispatch"; echo "## FETCH_SIZE"; echo '```'; cat $G/${TAG}_decode_pmc_FETCH_SIZE.txt; echo '```'; echo; echo "## WRITE_SIZE"; echo '```'; cat $G/${TAG}_decode_pmc_WRITE_SIZE.txt; echo '```'; } > $P/${TAG}_f_pmc_fetch_write_decode.md
{ hdr "the config-2 training step with split-bf16 operands: \`rocprofv3 --kernel-trace -- python bench.py --precision 32 --steps 6 --warmup 2 --no-cpu-baseline --no-sub-benches --no-roofline-pass\`"; cat $G/${TAG}_x3_train.md; } > $P/${TAG}_g_train_step_split_bf16_kernel_stats.md
[ -f $G/${TAG}_mfma_utilisation.md ] && { hdr "MFMA pipe utilisation: \`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE\` (own passes) over the bf16 step, the split-bf16 step and one beam decode; busy = SQ_VALU_MFMA_BUSY_CYCLES / 32 / GRBM_GUI_ACTIVE per dispatch, cycle-weighted (tools/mfma_util.py)"; cat $G/${TAG}_mfma_utilisation.md; } > $P/${TAG}_i_mfma_utilisation.md
[ -f $G/${TAG}_vendor_gemm_trace.md ] && { hdr "vendor GEMM (hipBLASLt through torch.matmul) vs this library, GPU-side dispatch durations, one shape per process: \`python tools/vendor_gemm_trace.py\` (measuring stick only; nothing in the product calls a vendor GEMM)"; cat $G/${TAG}_vendor_gemm_trace.md; } > $P/${TAG}_j_vendor_gemm_trace.md
cp $G/pmc_constants.json $P/pmc_constants.json
ls -la $P | grep ${TAG}_

#!/bin/bash
# Audit of the 64-rows-per-wave prefill kernel's build (mini-vllm_amd/csrc/prefill_mfma.hip): the kernel owns a[0:191] by
# name, so hipcc itself must not touch the accumulator file, spill, or use scratch.  Usage: tools/dbg/audit_pf64.sh
set -e
cd "$(dirname "$0")/../../mini-vllm_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-slp-vectorize -save-temps=obj -c prefill_mfma.hip -o $T/pf.o 2>/dev/null
S=$T/prefill_mfma-hip-amdgcn-amd-amdhsa-gfx950.s
rc=0
for kern in prefill_mfma64_kernel; do for tag in 8bf16_tag 7f16_tag; do
  awk -v k="${kern}I${tag}Li0E" '$0 ~ "^_ZN.*"k".*:" {f=1} f {print} f && /^; Occupancy/ {exit}' $S > $T/k.s
  # strip the asm blocks, then look for accumulator-file traffic of the compiler's own
  awk '/;;#ASMSTART/{a=1} !a{print} /;;#ASMEND/{a=0}' $T/k.s > $T/k_noasm.s
  n_acc=$(grep -c "accvgpr\| a\[[0-9]\|, a[0-9]" $T/k_noasm.s || true)
  n_scr=$(grep -c "scratch_" $T/k.s || true)
  vg=$(grep "; NumVgprs" $T/k.s | head -1); ag=$(grep "; NumAgprs" $T/k.s | head -1); sc=$(grep "; ScratchSize" $T/k.s | head -1)
  echo "$kern $tag: $vg $ag $sc | compiler AGPR uses outside asm: $n_acc, scratch ops: $n_scr, MFMAs: $(grep -c v_mfma $T/k.s)"
  if [ "$n_acc" != "0" ] || [ "$n_scr" != "0" ]; then rc=1; fi
done; done
rm -rf $T
exit $rc

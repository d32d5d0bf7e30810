#!/bin/bash
# Per-kernel count of compiler-inserted full drains (s_waitcnt vmcnt(0)) against the kernel's loads / LDS-DMA pieces / MFMAs, from the
# gfx950 assembly of the three translation units of libbp_hip.so.  Round 6 found three launches this way whose loads the compiler had
# serialised (load, s_waitcnt vmcnt(0), use -- once per load): the bf16 dgrad's y_{l-1} prologue (32 round trips in front of the k-loop),
# the output layer's slab reduce and the bf16-segment exchange kernel; and one experiment whose main loop was drained in front of every
# LDS read because the kernel had a second __shared__ object next to the LDS-DMA ring (profiles/r06_wgrad_tail_split.txt).
# A kernel whose first column is close to its second deserves a look at the listing (the .s files stay in $OUT).
#   usage: tools/asm_waitcnt_scan.sh [outdir]        (no GPU needed; ~1 min)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); OUT=${1:-/tmp/bp_asm}; mkdir -p $OUT
cd $R/dnn-for-speech-enhancement_amd/csrc
for f in bp_step bp_dp bp_profile; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o $OUT/$f.s $f.hip 2>/dev/null
  awk -v tu=$f '/^_Z[A-Za-z0-9_]+:|^bp_[a-z_0-9]+:/{k=$1; w0[k]=0; ld[k]=0; dma[k]=0; mf[k]=0}
       /s_waitcnt vmcnt\(0\)/{w0[k]++} /global_load_lds/{dma[k]++; next} /global_load|buffer_load/{ld[k]++} /v_mfma/{mf[k]++}
       END{for(k in w0) printf "%4d vmcnt(0) %4d loads %4d lds-dma %4d mfma  %s %s\n", w0[k], ld[k], dma[k], mf[k], tu, k}' $OUT/$f.s
done | sort -k1,1nr | c++filt | cut -c1-200

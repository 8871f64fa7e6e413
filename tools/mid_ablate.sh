#!/bin/bash
# bma_gemm_mid with parts of its k loop compiled out (WRONG results, timing only): what the loop costs without its DMA
# pieces / fragment reads -- in cycles AND in the clock the chip holds (tools/mid_stamps.py on stamped builds).  Build first:
#   for ab in 1 2 3; do make -C bimodalattack_amd/csrc OUTDIR=$PWD/bimodalattack_amd/lib_ab$ab EXTRA="-DBMA_MID_STAMPS -DBMA_MID_ABLATE=$ab"; done
# bits: 1 = no DMA pieces inside the loop, 2 = no fragment reads, 4 = no MFMAs
set -u
for ab in 3 1 2; do
  L=bimodalattack_amd/lib_ab$ab/libbma_hip.so
  [ -f "$L" ] || continue
  echo "== BMA_MID_ABLATE=$ab (1 = no DMA in the loop, 2 = no fragment reads, 3 = neither)"
  BMA_LIB=$PWD/$L python3 tools/mid_stamps.py "$@" 2>&1 | grep -v amdgpu.ids | head -7
done

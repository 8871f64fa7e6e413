#!/bin/bash
# bma_gemm_mid with parts of its k loop compiled out (WRONG results, timing only): what the loop costs without its DMA
# pieces / fragment reads / MFMAs / priority flips.  Build first:
#   for ab in 1 2 4 8 3; do make -C bimodalattack_amd/csrc OUTDIR=$PWD/bimodalattack_amd/lib_ab$ab EXTRA=-DBMA_MID_ABLATE=$ab; done
# bits: 1 = no DMA pieces inside the loop, 2 = no fragment reads, 4 = no MFMAs, 8 = no s_setprio
set -u
ONLY=${1:-gate_up}
for ab in 0 1 2 4 8 3; do
  L=bimodalattack_amd/lib_ab$ab/libbma_hip.so
  [ "$ab" = 0 ] && L=bimodalattack_amd/lib/libbma_hip.so
  [ -f "$L" ] || continue
  echo "== BMA_MID_ABLATE=$ab"
  BMA_LIB=$PWD/$L python3 tools/mid_pad_probe.py --pads 0 --only "$ONLY" 2>&1 | grep "pad w"
done

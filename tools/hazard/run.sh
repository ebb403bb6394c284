#!/bin/bash
# GPU side of tools/hazard_ab.sh: one GEMM (M K N) against fp64 per variant build, three repetitions each.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for v in ${VARIANTS:-isa_rne isa_pk isa_padall isa_cvtpk isa_mix isa_sdwa isa_pk32 isa_preds isa_waitds}; do
  echo "== $v"
  MTMC_DBG_LIB=tools/_hazard/$v/libmtmc_mpn.so timeout -k 10 120 python tools/dbg_gemm.py ${SHAPE:-9000 2048 1024} 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
R=r05c
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
bash tools/pmc_any.sh $R c3 grp16 "$CNT" HARC_AMD_GRP=1 > gpurun_out/$R/pmc_grp.log 2>&1; grep -A9 "k_steps_grp<4, 16" gpurun_out/$R/pv_grp16.txt
CNT2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY"
bash tools/pmc_any.sh $R c3 grp16b "$CNT2" HARC_AMD_GRP=1 > gpurun_out/$R/pmc_grpb.log 2>&1; grep -A9 "k_steps_grp<4, 16" gpurun_out/$R/pv_grp16b.txt

#!/usr/bin/env bash
# PMC passes (one counter group per pass) over tools/kone.py for one kernel name pattern.  usage: tools/pmc_bf.sh <kone mode> <kernel pattern> <out file>
set -u
R=$(cd "$(dirname "$0")/.." && pwd); MODE=$1; PAT=$2; OUT=$3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_x
i=0
for grp in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_x/p_$i -- python $R/tools/kone.py $MODE 4 > /dev/null 2>&1
done
python $R/tools/pmc_parse.py /tmp/pmc_x $PAT > $OUT 2>&1

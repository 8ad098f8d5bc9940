#!/bin/bash
# SQ counter passes on one GEMM shape (separate passes, kernel-trace off).  usage: tools/pmc_gemm.sh TAG M N K [epilogue]
set -e
TAG=$1; M=$2; N=$3; K=$4; EPI=${5:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/p1 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py $M $N $K $EPI > $OUT.p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py $M $N $K $EPI > $OUT.p2.log 2>&1
echo done

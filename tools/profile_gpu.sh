#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-extras"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_tcc -o pmc -- $CMD > $OUT/pmc_tcc.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_BF16 --kernel-trace -d $OUT/pmc_pipe -o pmc -- $CMD > $OUT/pmc_pipe.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d $OUT/pmc_f16 -o pmc -- $CMD > $OUT/pmc_f16.log 2>&1
timeout 300 rocprofv3 --pmc VALUBusy MfmaUtil --kernel-trace -d $OUT/pmc_util -o pmc -- $CMD > $OUT/pmc_util.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -50
# keep only what is small
find $OUT -size +8M -delete

#!/bin/bash
# usage: tools/gpu_job_pmc_mode.sh <mode of tools/stage_only.py> <tag>   -> gpurun_out/<tag>_{stats,pmc_fetch,pmc_write,pmc_sq,pmc_sq2}
# separate passes, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc)
R=$PWD; S=tools/stage_only.py; MODE=$1; T=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o s -- python3 $R/$S 20 $MODE > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -o s -- python3 $R/$S 6 $MODE > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_write -o s -- python3 $R/$S 6 $MODE > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_pmc_sq -o s -- python3 $R/$S 6 $MODE > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/${T}_pmc_sq2 -o s -- python3 $R/$S 6 $MODE > /dev/null 2>&1
cd $R

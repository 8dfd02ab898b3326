#!/bin/bash
# Counter passes over the three large tower GEMM shapes (one rocprofv3 run per pass); CSVs land in gpurun_out/gemm_pmc/
set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/gemm_pmc
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"
timeout -s KILL 120 rocprofv3 --pmc $P1 --output-format csv -d $R/gpurun_out/gemm_pmc/p1 -- python3 $R/tools/gemm_bench.py one fwd0 dx0 dw0 fwd1
timeout -s KILL 120 rocprofv3 --pmc $P2 --output-format csv -d $R/gpurun_out/gemm_pmc/p2 -- python3 $R/tools/gemm_bench.py one fwd0 dx0 dw0 fwd1

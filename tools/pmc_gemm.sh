#!/bin/bash
# per-kernel hardware counters of the GEMM microbenchmark (separate --pmc passes, kernel-trace only)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1; tail -12 gpurun_out/gemm_bench.log
rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/p$i -- python tools/gemm_bench.py dec_ffn1 enc_conv2 square attn_qk wgrad > gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i rc=$?"
done

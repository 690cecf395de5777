#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of single gemm_bench shapes (cold), with and without the m-fastest tile order
mkdir -p gpurun_out/pmcf
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for mf in 0 1; do
  for shape in dec_ffn2 dec_ffn1; do
    export FS2_GEMM_MFAST=$mf
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf/${shape}_mf${mf}_f -- python tools/gemm_bench.py cold $shape > gpurun_out/pmcf/${shape}_mf${mf}_f.log 2>&1
    echo "$shape mf=$mf done"
  done
done

#!/bin/bash
mkdir -p gpurun_out
for a in 0 1 2 4 8 6 14 15; do
  echo "=== ablate=$a" | tee -a gpurun_out/ablate.log
  FS2_GEMM_ABLATE=$a timeout -k 10 200 python tools/gemm_bench.py square dec_ffn1 enc_conv2 wgrad 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ablate.log
done

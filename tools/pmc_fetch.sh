#!/bin/bash
# FETCH_SIZE of single gemm_bench shapes (cold)
mkdir -p gpurun_out/pmcf
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for shape in "wgrad 1024" conv_wgrad; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf/${tag}_f -- python tools/gemm_bench.py cold "$shape" > gpurun_out/pmcf/${tag}_f.log 2>&1
  echo "$shape done"
done

#!/bin/bash
# Run the GPU-side checks in sequence on the gpurun box; stop at the first step that times out / is killed.
mkdir -p gpurun_out
STAMP=$(date +%Y%m%d_%H%M%S)
run() {  # name, timeout, command...
  local name=$1 tmo=$2; shift 2
  # every call keeps its own log (gpurun_out/<name>.<stamp>.log; <name>.log is a copy of the latest): a log that shows a GPU fault is
  # evidence and must not be overwritten by the next call
  local log="gpurun_out/$name.$STAMP.log"
  echo "=== $name  ($(date +%H:%M:%S))  $*" | tee -a gpurun_out/ci.log
  timeout -k 10 "$tmo" "$@" > "$log" 2>&1
  local rc=$?
  cp -f "$log" "gpurun_out/$name.log"
  echo "$name rc=$rc" | tee -a gpurun_out/ci.log
  tail -3 "$log" | tee -a gpurun_out/ci.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "ABORT after $name (timeout/kill)" | tee -a gpurun_out/ci.log; exit 1; fi
  # a step that died on the GPU (HSA error, memory fault, abort) ends the sequence: no further GPU step in this call
  if [ $rc -ne 0 ] && grep -q -E "HSA_STATUS_ERROR|Memory access fault|MEMORY_APERTURE|Aborted|core dumped|hipErrorIllegal|GPU core dump" "$log"; then
    echo "ABORT after $name (GPU fault signature in $log)" | tee -a gpurun_out/ci.log; grep -m5 -E "HSA_STATUS|fault|Abort" "$log" | tee -a gpurun_out/ci.log; exit 1
  fi
}
echo "##### gpu_ci $STAMP: $*" >> gpurun_out/ci.log
for step in "$@"; do
  case $step in
    kernels) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 180 -p no:cacheprovider ;;
    model)   run model 900 python -m pytest tests/test_model_gpu.py -q -m gpu --timeout 400 -p no:cacheprovider ;;
    trainer) run trainer 900 python -m pytest tests/test_trainer_gpu.py -q -m gpu --timeout 400 -p no:cacheprovider ;;
    bf16cfg) run bf16cfg 600 python -m pytest tests/test_model_gpu.py -q -m gpu --timeout 400 -p no:cacheprovider -k "benchmark_config" ;;
    smoke)   run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)   run bench 600 python bench.py ;;
    bench3)  run bench3 500 python bench.py --workload cfg3 --steps 24 --warmup 8 --no-cpu-baseline ;;
    bench4)  run bench4_bf16 500 python bench.py --workload cfg4 --steps 24 --warmup 16 --no-cpu-baseline
             run bench4_fp8 500 python bench.py --workload cfg4 --fp8 --steps 24 --warmup 16 --no-cpu-baseline ;;
    bench3a) run bench3a 500 python bench.py --workload cfg3 --return-attn --steps 24 --warmup 8 --no-cpu-baseline ;;
    prof4)   cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
             run prof4_bf16 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4_bf16 -- python bench.py --workload cfg4 --steps 3 --warmup 4 --no-cpu-baseline --no-graph --no-gemm-timer
             run prof4_fp8 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4_fp8 -- python bench.py --workload cfg4 --fp8 --steps 3 --warmup 4 --no-cpu-baseline --no-graph --no-gemm-timer ;;
    aten)    run aten 300 python tools/aten_ops_in_step.py ;;
    benchq)  run benchq 400 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --gemm-report gpurun_out/gemm_report.txt ;;
    benche)  run benche 400 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-graph ;;
    benchee) run benchee 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-graph --no-gemm-timer ;;
    prof)    cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
             run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 4 --warmup 8 --no-cpu-baseline --no-graph --no-overlap ;;
    ncclgraph) run ncclgraph 200 python tools/exp_nccl_graph.py ;;
    gemmbench) run gemmbench 300 python tools/gemm_bench.py ;;
    wsplit)  run wsplit 300 python tools/gemm_bench.py wsplit ;;
    gemmcold) run gemmcold 300 python tools/gemm_bench.py cold dec_ffn enc_conv post_conv attn square "wgrad 1024" ;;
    gemmepi) run gemmepi 300 python tools/gemm_bench.py cold epi ;;
    attnph)  FS2_ATTN_QB=64 run attnph64 200 python tools/attn_phases.py
             FS2_ATTN_QB=32 run attnph 200 python tools/attn_phases.py ;;
    absk)    FS2_SPLITK_FWD=0 run absk0 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_SPLITK_FWD=1 run absk1 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_SPLITK_FWD=0 run absk0b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_SPLITK_FWD=1 run absk1b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline ;;
    abmf)    FS2_GEMM_MFAST=0 run abmf0 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_MFAST=1 run abmf1 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_MFAST=0 run abmf0b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_MFAST=1 run abmf1b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline ;;
    abstream) FS2_KM_STREAM_SLICED=0 run abstream0 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-gemm-timer
             FS2_KM_STREAM_SLICED=1 run abstream1 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-gemm-timer
             FS2_KM_STREAM_SLICED=0 run abstream0b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-gemm-timer
             FS2_KM_STREAM_SLICED=1 run abstream1b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-gemm-timer ;;
    shapes)  FS2_GRAPH_POLICY=lru run shapes_lru 600 python bench.py --shapes 200 --no-cpu-baseline
             run shapes 600 python bench.py --shapes 200 --no-cpu-baseline ;;
    overlap) run overlap 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-gemm-timer --overlap ;;
    wgradk)  run wgradk 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 180 -p no:cacheprovider -k "wgrad or big_km" ;;
    abring)  FS2_GEMM_RING=1 run abring1 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_RING=2 run abring2 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_WS=2 run abws2 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_RING=1 run abring1b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_RING=2 run abring2b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_GEMM_WS=2 run abws2b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline ;;
    abwg)    FS2_WGRAD_SLICED=0 run abwg0 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_WGRAD_SLICED=1 run abwg1 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_WGRAD_SLICED=0 run abwg0b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline
             FS2_WGRAD_SLICED=1 run abwg1b 400 python bench.py --steps 32 --warmup 16 --no-cpu-baseline ;;
    ab)      FS2_FUSED_ATTN=0 run ab0 400 python bench.py --steps 24 --warmup 8 --no-cpu-baseline
             FS2_FUSED_ATTN=1 run ab1 400 python bench.py --steps 24 --warmup 8 --no-cpu-baseline
             FS2_FUSED_ATTN=0 run ab0b 400 python bench.py --steps 24 --warmup 8 --no-cpu-baseline
             FS2_FUSED_ATTN=1 run ab1b 400 python bench.py --steps 24 --warmup 8 --no-cpu-baseline ;;
    pmcbench) cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
             run pmcb1 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcb1 -- python bench.py --steps 4 --warmup 8 --no-cpu-baseline --no-graph --no-overlap
             run pmcb2 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcb2 -- python bench.py --steps 4 --warmup 8 --no-cpu-baseline --no-graph --no-overlap ;;
    bounds)  # every GEMM / weight-gradient / flash-attention descriptor of the model tests and of one eager step per workload through the host-side validator
             FS2_CHECK_BOUNDS=1 run bounds_model 900 python -m pytest tests/test_model_gpu.py tests/test_ar_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider -x
             FS2_CHECK_BOUNDS=1 run bounds_kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 300 -p no:cacheprovider -x -k "flash or wgrad or big_km or gemm or linear or conv"
             FS2_CHECK_BOUNDS=1 run bounds_cfg2 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph
             FS2_CHECK_BOUNDS=1 run bounds_cfg3 400 python bench.py --workload cfg3 --steps 3 --warmup 2 --no-cpu-baseline --no-graph
             FS2_CHECK_BOUNDS=1 run bounds_cfg4 400 python bench.py --workload cfg4 --fp8 --steps 3 --warmup 2 --no-cpu-baseline --no-graph ;;
    pmcsq)   # SQ counters of every kernel of the step, two passes (8 SQ slots each), kernel trace only
             cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
             run pmcsq1 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcsq1 -- python bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-graph --no-overlap
             run pmcsq2 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmcsq2 -- python bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-graph --no-overlap
             run pmcsq3 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmcsq3 -- python bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-graph --no-overlap ;;
    benchdp) FS2_FORCE_DP=1 run benchdp 400 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-graph
             FS2_FORCE_DP=1 FS2_GRAPH_DP=1 run benchdpg 400 python bench.py --steps 8 --warmup 4 --no-cpu-baseline
             run benchtr 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline ;;
    *) echo "unknown step $step" ;;
  esac
done

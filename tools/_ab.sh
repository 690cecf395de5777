for v in 256 512 384 256 512; do FS2_RED_BLOCKS=$v python bench.py --no-cpu-baseline --steps 60 --warmup 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('red $v', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"; done

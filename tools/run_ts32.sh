#!/bin/bash
# sweep: 32 x 32 tiles from a given level size up, with and without level 0 among the fused levels
export PYLAMP_BENCH_NO_4097=1
run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'])"; }
run base
PYLAMP_MG_TS32=1000000 run ts32_L1
PYLAMP_MG_TS32=260000 run ts32_L1L2
PYLAMP_MG_TS32=1000000 PYLAMP_MG_FUSED_MAX=5000000 run ts32_L0L1_fusedL0
PYLAMP_MG_TS32=4000000 PYLAMP_MG_FUSED_MAX=5000000 run ts32_L0_fusedL0

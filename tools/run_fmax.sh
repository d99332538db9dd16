#!/bin/bash
# sweep of the largest fused multigrid level
export PYLAMP_BENCH_NO_4097=1
for m in 1 70000 300000 1100000 5000000; do
  PYLAMP_MG_FUSED_MAX=$m python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused_max $m', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'])"
done

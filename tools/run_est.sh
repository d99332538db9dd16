#!/bin/bash
export PYLAMP_BENCH_NO_4097=1
for e in 0 1; do
  PYLAMP_EST_EXACT=$e python bench.py --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exact=$e', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], d['stokes_precond_applies'], d['stokes_operator_applies'])"
done

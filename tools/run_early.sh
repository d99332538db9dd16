for k in -1 0 2 3 4; do
  PYLAMP_MG_EARLY=$k python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('early=$k', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], d['stokes_rel_residual'], d['stokes_converged'])
"
done

for sc in 1.0 0.7 0.5 0.35 1.4; do
  PYLAMP_SCHUR_SCALE=$sc python bench.py --steps 8 --warmup 2 --no-cpu-baseline --apply-reps 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('schur_scale=$sc', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], min(d['stokes_converged']))
"
done

# Initial-guess experiment on the GPU box: linear (order 1, the default) against quadratic extrapolation in time.
for o in 2 3; do
  PYLAMP_X0_ORDER=$o python bench.py --steps 12 --warmup 4 --no-cpu-baseline --apply-reps 2 2> gpurun_out/x0_o$o.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('order=$o', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], min(d['stokes_converged']))
"
done

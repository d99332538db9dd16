# Shadow-residual experiment on the GPU box: 0 = seeded random (default), 1 = the initial residual when the solve starts from a guess
for m in 0 1; do
  PYLAMP_SHADOW=$m PYLAMP_SOLVER_TRACE=1 python bench.py --steps 16 --warmup 6 --no-cpu-baseline --apply-reps 2 2> gpurun_out/shadow_$m.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); it = d['stokes_iterations']; print('shadow=$m', d['ms_per_step'], d['stage_ms']['ms_stokes'], sum(it) / len(it), it, d['heat_iterations'], min(d['stokes_converged']))
"
done

for w in 0 0.5 1.0; do
  PYLAMP_X0_EXTRAP=$w PYLAMP_SOLVER_TRACE=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --apply-reps 2 2> gpurun_out/x0_$w.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('extrap=$w', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], min(d['stokes_converged']))
"
  grep "it   1 " gpurun_out/x0_$w.err | grep -v pylamp3 | awk '{print $6}' | tr '\n' ' ' | cut -c1-400; echo
done

run() {
  env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --apply-reps 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$*', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], min(d['stokes_converged']))
"
}
run X=1
run PYLAMP_MG_TAIL_NU=2,2
run PYLAMP_MG_TAIL_NU=2,1
run PYLAMP_MG_COARSE=6
run PYLAMP_MG_TAIL_NODES=289
run PYLAMP_MG_TAIL_NODES=4225

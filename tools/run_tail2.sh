run() {
  env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --apply-reps 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$*', d['ms_per_step'], d['stage_ms']['ms_stokes'], sum(d['stokes_iterations']), min(d['stokes_converged']))
"
}
for rep in 1 2; do
run X=1
run PYLAMP_MG_TAIL_NODES=289
run PYLAMP_MG_TAIL_NODES=81
done

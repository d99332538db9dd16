# A/B of solver knobs on the bench problem, 8 timed steps each, every configuration twice
run() {
  env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$*', d['ms_per_step'], d['stage_ms']['ms_stokes'], sum(d['stokes_iterations'])/len(d['stokes_iterations']), min(d['stokes_converged']))
"
}
for rep in 1 2; do
  run X=1
  run PYLAMP_MG_POWER=1
  run PYLAMP_MG_NU0=1,1 PYLAMP_MG_NU=2,3
  run PYLAMP_MG_NU0=1,1 PYLAMP_MG_NU=1,4
  run PYLAMP_MG_NU0=1,1 PYLAMP_MG_NU=2,2
  run PYLAMP_MG_NU0=1,2 PYLAMP_MG_NU=3,3
done

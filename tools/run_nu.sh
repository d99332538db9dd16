# smoothing-count sweep on the bench problem: levels >= 1 (PYLAMP_MG_NU), finest level V(1,1)
run() {
  env PYLAMP_MG_NU0=1,1 "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$*', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], d['stokes_converged'])
"
}
for nu in 3,3 0,6 0,5 0,4 1,5 1,4 2,4 1,3 2,3; do
  run PYLAMP_MG_EARLY=0 PYLAMP_MG_NU=$nu
done
for nu in 1,5 1,4 2,4; do
  run PYLAMP_MG_EARLY=4 PYLAMP_MG_NU=$nu
  run PYLAMP_MG_EARLY=3 PYLAMP_MG_NU=$nu
done
run PYLAMP_MG_EARLY=0 PYLAMP_MG_NU=3,3 PYLAMP_MG_TAIL_NU=2,2
run PYLAMP_MG_EARLY=0 PYLAMP_MG_NU=3,3 PYLAMP_MG_TAIL_NU=1,3
run PYLAMP_MG_EARLY=0 PYLAMP_MG_NU=3,3 PYLAMP_MG_TAIL_NU=0,4

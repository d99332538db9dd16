# Initial-guess experiment on the GPU box: degree of the extrapolating polynomial x number of older solutions it is fitted to.
for op in "2 2" "2 3" "2 4" "2 5" "1 2" "1 3" "3 5"; do
  set -- $op
  PYLAMP_X0_ORDER=$1 PYLAMP_X0_POINTS=$2 python bench.py --steps 16 --warmup 6 --no-cpu-baseline --apply-reps 2 2> gpurun_out/x0_o$1_p$2.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); it = d['stokes_iterations']; print('degree=$1 older=$2', d['ms_per_step'], d['stage_ms']['ms_stokes'], sum(it) / len(it), it, min(d['stokes_converged']))
"
done

# 3-D smoothing experiment on the GPU box: sweeps on the finest level / on the others
for c in "0 2" "1 2" "1 3" "2 3" "1 4"; do
  set -- $c
  PYLAMP_MG_NU3_FINE=$1 PYLAMP_MG_NU3=$2 python bench.py --config 3d257 --steps 2 --warmup 1 2> gpurun_out/nu3_$1_$2.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('fine=$1 others=$2', d['ms_per_step'], d['stokes_iterations'], d['stokes_rel_residual'], d['stokes_converged'])
"
done

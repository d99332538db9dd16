# Heat initial-guess experiment on the GPU box: 0 = zero start, 1 = old nodal temperature, 2 = old temperature + scaled last increment
for m in 0 1 2; do
  PYLAMP_HEAT_X0=$m python bench.py --steps 10 --warmup 4 --no-cpu-baseline --apply-reps 2 2> gpurun_out/heatx0_$m.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('heat_x0=$m', d['ms_per_step'], d['stage_ms']['ms_heat'], d['heat_iterations'], d['stokes_iterations'])
"
done

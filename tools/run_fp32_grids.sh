for n in 513 1025; do
  for f in 1 0; do
    PYLAMP_MG_FP32=$f PYLAMP_MG_FP32_NODES=1000 python bench.py --grid $n --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('grid $n fp32=$f', d['ms_per_step'], d['stage_ms']['ms_stokes'], d['stokes_iterations'], d['stokes_rel_residual'])
"
  done
done

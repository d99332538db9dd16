#!/bin/bash
# Stokes iteration counts of both bench models under an environment setting:  bash tools/iter_probe.sh "VAR=value ..." [steps]
# (experiments on the multigrid hierarchy: the headline model must not pay for what helps the falling block)
envs="$1"; steps=${2:-6}
for m in block mantle; do
  env $envs PYLAMP_BENCH_NO_4097=1 python bench.py --model $m --steps $steps --warmup 2 --no-cpu-baseline > gpurun_out/ip_$m.log 2>&1
  python - "$m" "$envs" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/ip_%s.log'%sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[1], sys.argv[2], "FAILED"); sys.exit(0)
d=json.loads(l[-1])
print("%-7s %-40s ms/step %.2f stokes_ms %s iters %s conv %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d.get("stage_ms",{}).get("stokes"), d.get("stokes_iterations_mean", d.get("stokes_iterations")), d.get("stokes_converged")))
PY
done

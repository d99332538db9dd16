export PYLAMP_SOLVER_TRACE=1
python tools/traj_rtol.py 2>&1 | grep -E "x n =|traj_" | head -60
for n in 129 513; do python tools/tune_nu.py mantle $n 1,1 3,3 2>&1 | grep -E "x n =|mantle" | tail -4; done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --apply-reps 2 2>&1 | grep -E "x n =" | tail -3

for v in 0 8 4; do
echo "== PYLAMP_APPLY_LDS=$v"
PYLAMP_APPLY_LDS=$v python -m pytest tests/test_hip_parity.py tests/test_hip_solve.py -x -q -m gpu 2>&1 | tail -1
PYLAMP_APPLY_LDS=$v python tools/apply_bench.py 2049 200 2>&1 | tail -1
PYLAMP_APPLY_LDS=$v python tools/apply_bench.py 4097 50 2>&1 | tail -1
done

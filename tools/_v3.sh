python -m pytest tests/test_hip_parity.py tests/test_hip_solve.py tests/test_hip_step.py -x -q -m gpu 2>&1 | tail -1
python tools/tune.py nu 2049 2>&1 | grep step | tail -2

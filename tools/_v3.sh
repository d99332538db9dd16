for r in 2 4 8 16; do
echo "== ROWS=$r"
PYLAMP_APPLY_ROWS=$r python tools/apply_bench.py 2049 200 2>&1 | tail -1
PYLAMP_APPLY_ROWS=$r python tools/apply_bench.py 4097 50 2>&1 | tail -1
PYLAMP_APPLY_ROWS=$r python tools/apply_bench.py 1025 400 2>&1 | tail -1
done

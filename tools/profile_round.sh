#!/bin/bash
# Profiles of one round, run on the GPU box from the repository root:
#     PYLAMP_COMMIT=<short hash> bash tools/profile_round.sh <tag>        (e.g. r02)
# 1. kernel trace + stats of the bench command, 2./3. FETCH_SIZE and WRITE_SIZE in separate PMC passes over tools/pmc_run.py
# (never combined with a trace), then tools/make_traffic.py; the small results land in gpurun_out/profiles_<tag>/ -- copy them
# into profiles/ and commit.
set -e -o pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
out=gpurun_out/profiles_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -- python3 bench.py --steps 5 --warmup 2 > $out/${tag}_bench_line.json 2> gpurun_out/prof_kt.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -- python3 tools/pmc_run.py > gpurun_out/prof_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -- python3 tools/pmc_run.py > gpurun_out/prof_write.log 2>&1
echo "write pass done"
F=$(find gpurun_out/prof_fetch -name "*.db" | head -1); W=$(find gpurun_out/prof_write -name "*.db" | head -1); K=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python3 tools/make_traffic.py $F $W $tag $K > gpurun_out/make_traffic.log 2>&1
cp profiles/traffic.json profiles/${tag}_pmc_summary.csv profiles/${tag}_kernel_stats.csv $out/
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
grep '^{' $out/${tag}_bench_line.json > $out/line.tmp && mv $out/line.tmp $out/${tag}_bench_line.json
ls -la $out

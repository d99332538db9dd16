#!/bin/bash
# kernel trace of a short bench run: bash tools/kt.sh <tag> [bench args...]   -> gpurun_out/kt_<tag>.csv + line
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYLAMP_BENCH_NO_4097=1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/kt_$tag.json 2> gpurun_out/kt_$tag.err
K=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python3 tools/kernel_stats.py $K > gpurun_out/kt_$tag.csv
python3 tools/gaps.py $K > gpurun_out/gaps_$tag.txt 2>&1
rm -rf gpurun_out/prof_$tag
head -30 gpurun_out/kt_$tag.csv | cut -c1-150

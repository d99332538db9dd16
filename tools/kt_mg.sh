#!/bin/bash
# kernel trace of the preconditioner probe: bash tools/kt_mg.sh <tag> [n]   (environment knobs pass through)
tag=$1; n=${2:-2049}
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --kernel-trace --stats -d gpurun_out/pk_$tag -- python3 tools/mg_probe.py $n 3 > gpurun_out/pk_$tag.log 2>&1
K=$(find gpurun_out/pk_$tag -name "*.db" | head -1)
python3 - $K <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcol = "grid_x" if "grid_x" in cols else "grid_size_x"
agg = collections.defaultdict(list)
for name, gs, st, en in cur.execute("select name, %s, start, end from kernels order by start" % gcol):
    if "k_mg_" in name or "k_vv_" in name or "k_prec" in name: agg[(name.split("(")[0].replace("void ", "")[:40], gs)].append((en - st) / 1e3)
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-42s grid %8d  n %3d  min %7.2f  med %7.2f us" % (n, g, len(v), min(v), sorted(v)[len(v) // 2]))
PY
rm -rf gpurun_out/pk_$tag

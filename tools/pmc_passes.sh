#!/bin/bash
# Kernel trace + hardware counters of one target, every counter group in a pass of its own (never combined with a trace):
#     bash tools/pmc_passes.sh <tag> <target args of tools/pmc_target.py ...>      e.g.  bash tools/pmc_passes.sh r04_step step 2049 16 3
# -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc.csv (per kernel: calls, average duration, FETCH_SIZE x 2 + WRITE_SIZE per launch,
#    SQ instruction / wait / occupancy counters and what follows from them; tools/pmc_summary.py)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYLAMP_BENCH_NO_4097=1
run() {   # name, counters...
  local name=$1; shift
  rocprofv3 "$@" -d gpurun_out/pp_${tag}_$name -- python3 tools/pmc_target.py "${TARGET[@]}" > gpurun_out/pp_${tag}_$name.log 2>&1
  echo "$name pass done"
}
TARGET=("$@")
run kt --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
if [ -z "$PMC_QUICK" ]; then
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
fi
dbs=""
for p in kt fetch write sq1 sq2; do [ -d gpurun_out/pp_${tag}_$p ] || continue; dbs="$dbs $(find gpurun_out/pp_${tag}_$p -name '*.db' | head -1)"; done
python3 tools/pmc_summary.py $tag $dbs
for p in kt fetch write sq1 sq2; do rm -rf gpurun_out/pp_${tag}_$p; done
head -25 gpurun_out/${tag}_pmc.csv | cut -c1-260

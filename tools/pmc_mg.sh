#!/bin/bash
# SQ counters of the multigrid kernels: bash tools/pmc_mg.sh <tag> [n]
tag=$1; n=${2:-2049}
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d gpurun_out/pg1_$tag -- python3 tools/mg_probe.py $n 3 > gpurun_out/pg1_$tag.log 2>&1
K=$(find gpurun_out/pg1_$tag -name "*.db" | head -1)
python3 tools/pmc_any.py $K k_mg_ g > gpurun_out/pmc_mg_$tag.txt
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/pg2_$tag -- python3 tools/mg_probe.py $n 3 > gpurun_out/pg2_$tag.log 2>&1
K=$(find gpurun_out/pg2_$tag -name "*.db" | head -1)
python3 tools/pmc_any.py $K k_mg_ g >> gpurun_out/pmc_mg_$tag.txt
rm -rf gpurun_out/pg1_$tag gpurun_out/pg2_$tag
grep -E "g=(2974400|2433600|no grid)" gpurun_out/pmc_mg_$tag.txt

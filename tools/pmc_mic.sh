#!/bin/bash
# SQ counters of the marker kernels: bash tools/pmc_mic.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d gpurun_out/pm1_$tag -- python3 tools/mic_probe.py 2049 16 2 > gpurun_out/pm1_$tag.log 2>&1
K=$(find gpurun_out/pm1_$tag -name "*.db" | head -1)
python3 tools/pmc_any.py $K k_ > gpurun_out/pmc_mic_$tag.txt
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/pm2_$tag -- python3 tools/mic_probe.py 2049 16 2 > gpurun_out/pm2_$tag.log 2>&1
K=$(find gpurun_out/pm2_$tag -name "*.db" | head -1)
python3 tools/pmc_any.py $K k_ >> gpurun_out/pmc_mic_$tag.txt
rm -rf gpurun_out/pm1_$tag gpurun_out/pm2_$tag
grep -E "scatter_cells<|k_rk4|k_gather" gpurun_out/pmc_mic_$tag.txt

# HBM-side traffic of the 3-D kernels (FETCH_SIZE and WRITE_SIZE in separate PMC passes, never combined with a trace)
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/p3f -- python3 bench.py --config 3d257 --steps 1 --warmup 0 > gpurun_out/p3f.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/p3w -- python3 bench.py --config 3d257 --steps 1 --warmup 0 > gpurun_out/p3w.log 2>&1
echo "write pass done"
F=$(find gpurun_out/p3f -name "*.db" | head -1); W=$(find gpurun_out/p3w -name "*.db" | head -1)
python3 tools/pmc3d.py $F FETCH_SIZE > gpurun_out/pmc3d_fetch.txt
python3 tools/pmc3d.py $W WRITE_SIZE > gpurun_out/pmc3d_write.txt
rm -rf gpurun_out/p3f gpurun_out/p3w
head -14 gpurun_out/pmc3d_fetch.txt; head -8 gpurun_out/pmc3d_write.txt

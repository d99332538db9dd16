#!/bin/bash
# Summed kernel time of a steady-state time step, one rank against 2 x 4 virtual ranks (rocprofv3 --kernel-trace --stats, no counters):
#     bash tools/rehearse_trace.sh <tag>      -> gpurun_out/<tag>_rehearsal_kernel_time.json, <tag>_rehearsal_{one,blocks}_kernel_stats.csv
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
for mode in one blocks; do
  for st in 3 7; do
    PYLAMP_LOCAL_SERIAL=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${mode}_$st -- python3 tools/rehearse_trace.py $mode $st > gpurun_out/rt_${mode}_$st.json 2> gpurun_out/rt_${mode}_$st.err
    K=$(find gpurun_out/prof_${mode}_$st -name "*.db" | head -1)
    python3 tools/kernel_stats.py $K > gpurun_out/${tag}_rehearsal_${mode}_${st}steps_kernel_stats.csv
    rm -rf gpurun_out/prof_${mode}_$st
    echo "$mode $st done"
  done
done
python3 - <<PY
import csv, json
def total(f, wire=True):
    # wire = False: without the device-to-device copies (on virtual ranks they ARE the messages of the in-process transport)
    return sum(float(r["total_us"]) for r in csv.DictReader(open(f)) if wire or "copyBuffer" not in r["kernel"])
out = {}
for mode in ("one", "blocks"):
    a, b = total("gpurun_out/${tag}_rehearsal_%s_3steps_kernel_stats.csv" % mode), total("gpurun_out/${tag}_rehearsal_%s_7steps_kernel_stats.csv" % mode)
    a2, b2 = total("gpurun_out/${tag}_rehearsal_%s_3steps_kernel_stats.csv" % mode, False), total("gpurun_out/${tag}_rehearsal_%s_7steps_kernel_stats.csv" % mode, False)
    out[mode] = {"kernel_us_3_steps": round(a, 1), "kernel_us_7_steps": round(b, 1), "kernel_ms_per_steady_step": round((b - a) / 4e3, 3),
                 "kernel_ms_per_steady_step_without_copies": round((b2 - a2) / 4e3, 3),
                 "run": json.loads([l for l in open("gpurun_out/rt_%s_7.json" % mode) if l.startswith("{")][-1])}
out["ratio_blocks_over_one"] = round(out["blocks"]["kernel_ms_per_steady_step"] / out["one"]["kernel_ms_per_steady_step"], 3)
out["ratio_blocks_over_one_without_copies"] = round(out["blocks"]["kernel_ms_per_steady_step_without_copies"] / out["one"]["kernel_ms_per_steady_step_without_copies"], 3)
out["note"] = "summed kernel time of ALL ranks of the 2 x 4 layout (PYLAMP_LOCAL_SERIAL=1: one virtual rank at a time on the GPU, so the durations are true) against one rank; per rank = sum / 8"
json.dump(out, open("gpurun_out/${tag}_rehearsal_kernel_time.json", "w"), indent=1)
print(json.dumps(out))
PY

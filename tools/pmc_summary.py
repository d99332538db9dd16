"""Per-kernel summary of the passes of tools/pmc_passes.sh:
    python tools/pmc_summary.py <tag> <kernel-trace db> <FETCH_SIZE db> <WRITE_SIZE db> <SQ pass dbs ...>
-> gpurun_out/<tag>_kernel_stats.csv (calls, total, average: the trace) and gpurun_out/<tag>_pmc.csv:
   kernel, calls, avg_us, fetch_MB (FETCH_SIZE x 2: the gfx950 correction of MI355X_MICROARCH.md), write_MB, traffic_MB, GB/s = traffic / average
   duration, the medians of the SQ counters per launch, and
     valu_busy      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      share of its resident time a wave spends issuing vector ALU instructions
     wait_share     = SQ_WAIT_ANY / SQ_WAVE_CYCLES              ... parked on s_waitcnt / barriers (memory latency)
     stall_share    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES         ... stalled at issue
     waves_per_simd = 4 SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)  achieved occupancy (SQ_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
     simd_valu_busy = valu_busy x waves_per_simd                             share of the SIMDs' time spent issuing vector ALU instructions
     valu_per_wave, vmem_rd_per_wave, lds_per_wave              instructions per wave
One problem size per process: kernels are keyed by name only."""
import csv, re, sqlite3, statistics, sys, collections

tag = sys.argv[1]; kt = sys.argv[2]; dbs = sys.argv[3:]


def key(name):
    return re.sub(r"^void ", "", name).split("(")[0].strip()


stats = {}
cur = sqlite3.connect(kt).cursor()
with open("gpurun_out/%s_kernel_stats.csv" % tag, "w") as f:
    f.write("kernel,calls,total_us,average_us,percent\n")
    for name, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        f.write('"%s",%d,%.1f,%.3f,%.2f\n' % (name, calls, tot, avg, pct))
        k = key(name)
        c0, t0 = stats.get(k, (0, 0.0))
        stats[k] = (c0 + calls, t0 + tot)
med = collections.defaultdict(dict)
for db in dbs:
    c = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counters_collection" in t][0]
    acc = collections.defaultdict(list)
    for name, cname, val in c.execute("select kernel_name, counter_name, value from %s" % view):
        acc[(key(name), cname)].append(float(val))
    for (k, cname), v in acc.items():
        med[k][cname] = statistics.median(v)
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64"]
rows = []
for k, (calls, tot) in stats.items():
    m = med.get(k, {})
    avg = tot / max(calls, 1)
    fe = m.get("FETCH_SIZE"); wr = m.get("WRITE_SIZE")
    fmb = None if fe is None else 2.0 * fe * 1024.0 / 1e6
    wmb = None if wr is None else wr * 1024.0 / 1e6
    tmb = None if fmb is None or wmb is None else fmb + wmb
    wc = m.get("SQ_WAVE_CYCLES") or 0.0
    d = {"kernel": k, "calls": calls, "avg_us": round(avg, 3), "total_ms": round(tot / 1e3, 3), "fetch_MB": fmb, "write_MB": wmb, "traffic_MB": tmb,
         "GBps": None if not tmb else round(tmb / avg * 1e3, 1)}
    for cn in cols:
        d[cn] = m.get(cn)
    d["valu_busy"] = round(m["SQ_ACTIVE_INST_VALU"] / wc, 4) if wc and "SQ_ACTIVE_INST_VALU" in m else None
    d["wait_share"] = round(m["SQ_WAIT_ANY"] / wc, 4) if wc and "SQ_WAIT_ANY" in m else None
    d["stall_share"] = round(m["SQ_WAIT_INST_ANY"] / wc, 4) if wc and "SQ_WAIT_INST_ANY" in m else None
    ga = m.get("GRBM_GUI_ACTIVE")
    d["waves_per_simd"] = round(4.0 * wc / (ga / 8.0 * 1024.0), 3) if wc and ga else None      # (GRBM_GUI_ACTIVE comes summed over the 8 XCDs)
    d["simd_valu_busy"] = round(d["valu_busy"] * d["waves_per_simd"], 3) if d["valu_busy"] is not None and d["waves_per_simd"] is not None else None
    w = m.get("SQ_WAVES")
    for a, b in (("valu_per_wave", "SQ_INSTS_VALU"), ("vmem_rd_per_wave", "SQ_INSTS_VMEM_RD"), ("lds_per_wave", "SQ_INSTS_LDS"), ("fma64_per_wave", "SQ_INSTS_VALU_FMA_F64")):
        d[a] = round(m[b] / w, 1) if w and b in m else None
    rows.append(d)
rows.sort(key=lambda r: -r["total_ms"])
names = list(rows[0].keys())
with open("gpurun_out/%s_pmc.csv" % tag, "w") as f:
    wri = csv.DictWriter(f, fieldnames=names)
    wri.writeheader()
    for r in rows:
        wri.writerow({k: ("" if v is None else (round(v, 3) if isinstance(v, float) else v)) for k, v in r.items()})
print("wrote gpurun_out/%s_kernel_stats.csv and gpurun_out/%s_pmc.csv (%d kernels)" % (tag, tag, len(rows)))

"""profiles/traffic.json + a per-kernel summary CSV from two rocprofv3 PMC passes over tools/pmc_run.py:

    python tools/make_traffic.py <fetch pass: counter_collection.csv or results.db> <write pass: same> <round-tag> [kernel-trace results.db]

(rocprofv3 of ROCm 7.2 writes a rocpd SQLite database by default; `--output-format csv` gives the CSV.  With a fourth
argument the kernel statistics of a `--kernel-trace --stats` run are written to profiles/<tag>_kernel_stats.csv.)

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts a wide coalesced read at half
its bytes (MI355X_MICROARCH.md, HBM section): it is doubled here, and the factor is CHECKED on the stream triad of the
same run (2 GiB read, 1 GiB written per launch).  Kernels are keyed by name and by the grid side derived from the launch
size, bytes per launch = median over the dispatches of that key.  The commit of the build is recorded."""
import csv, json, os, re, subprocess, sys, collections
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
fetch_csv, write_csv, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def rows_of(path, counter):
    if path.endswith(".db"):
        import sqlite3
        cur = sqlite3.connect(path).cursor()
        for name, grid, val in cur.execute("select kernel_name, grid_size, value from counters_collection where counter_name = ?", (counter,)):
            yield name, int(grid), float(val)
    else:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                yield r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])


def load(path, counter):
    d = collections.defaultdict(list)
    for kname, grid, val in rows_of(path, counter):
        name = re.sub(r"^void ", "", kname).split("(")[0].strip()
        d[(name, grid)].append(val * 1024.0)
    return {k: float(np.median(v)) for k, v in d.items()}, {k: len(v) for k, v in d.items()}


F, nF = load(fetch_csv, "FETCH_SIZE")
W, nW = load(write_csv, "WRITE_SIZE")
# calibration on the triad
tri = [k for k in F if k[0].startswith("k_triad")]
calib = None
if tri:
    k = tri[0]
    calib = {"fetch_raw_bytes": F[k], "expected_read_bytes": 2.0 * (1 << 30), "fetch_factor": 2.0 * (1 << 30) / F[k],
             "write_bytes": W.get(k), "expected_write_bytes": 1.0 * (1 << 30)}
factor = 2.0


def side(name, grid):
    """grid side n for the 2-columns-per-lane kernels: launches are ((n+127)/128 * 64) x rows threads"""
    for n in (2049, 4097, 1025, 513):
        gx = (n + 127) // 128 * 64
        if grid % gx == 0 and abs(grid // gx - n) <= 16:
            return n
    return None


out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/pmc_run.py, MI355X, %s; FETCH_SIZE "
                      "doubled per MI355X_MICROARCH.md; bytes per launch (median over dispatches), Infinity-Cache hits included" % tag,
       "_commit": os.environ.get("PYLAMP_COMMIT") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
       "_triad_calibration": calib}
names = {"k_stokes_apply_v2<4, true>": "k_stokes_apply_scaled", "k_stokes_apply_v2<16, true>": "k_stokes_apply_scaled",
         "k_stokes_apply_v2<4, false>": "k_stokes_apply", "k_stokes_apply_v2<16, false>": "k_stokes_apply",
         "k_vv_sweep2<0>": "k_vv_sweep2_cheb", "k_vv_sweep2<1>": "k_vv_sweep2_residual", "k_prec_stage1_v2": "k_prec_stage1_v2",
         "k_vv_first2": "k_vv_first2",
         # names since the multigrid kernels are templates over the level type
         "k_vv_sweep2<0, double, double>": "k_vv_sweep2_cheb", "k_vv_sweep2<1, double, double>": "k_vv_sweep2_residual",
         "k_prec_stage1_v2<double>": "k_prec_stage1_v2", "k_vv_first2<double>": "k_vv_first2"}
rows = []
for (name, grid), fb in sorted(F.items()):
    wb = W.get((name, grid), 0.0)
    rows.append((name, grid, nF[(name, grid)], fb * factor, wb))
    key = names.get(name)
    n = side(name, grid)
    if key and n:
        out.setdefault(key, {})[str(n)] = int(round(fb * factor + wb))
        out.setdefault(key + "_split", {})[str(n)] = {"fetch": int(round(fb * factor)), "write": int(round(wb))}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "%s_pmc_summary.csv" % tag), "w") as f:
    f.write("kernel,grid_size,dispatches,fetch_bytes_x2,write_bytes\n")
    for r in rows:
        f.write("%s,%d,%d,%.0f,%.0f\n" % (r[0].replace(",", ";"), r[1], r[2], r[3], r[4]))
if len(sys.argv) > 4:
    import sqlite3
    cur = sqlite3.connect(sys.argv[4]).cursor()
    with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag), "w") as f:
        f.write("kernel,calls,total_us,average_us,percent\n")
        for name, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            f.write('"%s",%d,%.1f,%.3f,%.2f\n' % (name, calls, tot, avg, pct))
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Export what the judge reads from a rocprofv3 output directory (its *_results.db): the per-kernel totals of a
--kernel-trace --stats run as CSV, or the mean of every --pmc counter per kernel.
   python tools/profile_summary.py stats gpurun_out/<dir> > profiles/<name>.csv
   python tools/profile_summary.py pmc   gpurun_out/<dir> [kernel-name-substring ...] > profiles/<name>.txt"""
import sys, glob, sqlite3, collections

mode, root = sys.argv[1], sys.argv[2]
db = sorted(glob.glob(root + "/**/*_results.db", recursive=True))[0]
c = sqlite3.connect(db)
if mode == "stats":
    print("kernel,calls,total_us,average_us,percent")
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        print('"%s",%d,%.3f,%.3f,%.4f' % (name.replace('"', "'"), calls, total, avg, pct))
else:
    want = sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in c.execute("select kernel_name, counter_name, value, workgroup_size, grid_size, lds_block_size, scratch_size, vgpr_count from counters_collection"):
        if want and not any(w in r[0] for w in want):
            continue
        agg[r[0]][r[1]].append(r[2]); meta[r[0]] = r[3:]
    for k in agg:
        print("%s\n    workgroup %d  grid %d  lds %d B  scratch %d B/lane  vgpr %d" % ((k,) + tuple(meta[k])))
        for cn, v in sorted(agg[k].items()):
            print("    %-22s dispatches=%d mean=%.6g min=%.6g max=%.6g" % (cn, len(v), sum(v) / len(v), min(v), max(v)))

#!/usr/bin/env python3
"""Export what the judge reads from a rocprofv3 output directory (its *_results.db): the per-kernel totals of a
--kernel-trace --stats run as CSV, or the mean of every --pmc counter per kernel.
   python tools/profile_summary.py stats gpurun_out/<dir> > profiles/<name>.csv
   python tools/profile_summary.py pmc   gpurun_out/<dir> [kernel-name-substring ...] > profiles/<name>.txt
   python tools/profile_summary.py idle  gpurun_out/<dir> <kernel-name-substring>   (GPU idle time between launches, per stretch
                                            from one launch of that kernel to the next: host gaps inside an epoch)
   python tools/profile_summary.py timeline gpurun_out/<dir> <kernel-name-substring> <count>   (the last <count> matching launches but 30:
                                            stream, start, duration, grid -- what runs beside what)"""
import sys, glob, sqlite3, collections

mode, root = sys.argv[1], sys.argv[2]
db = sorted(glob.glob(root + "/**/*_results.db", recursive=True))[0]
c = sqlite3.connect(db)
if mode == "stats":
    print("kernel,calls,total_us,average_us,percent")
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        print('"%s",%d,%.3f,%.3f,%.4f' % (name.replace('"', "'"), calls, total, avg, pct))
elif mode == "idle":
    # GPU idle time between consecutive launches (any stream), split at every launch of the kernel named in argv[3] (one segment per
    # epoch when that is k_mcmc): busy = union of the kernel intervals
    sub = sys.argv[3]
    rows = list(c.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if sub in r[0]]
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        t0, t1 = seg[0][1], rows[b][1]
        busy, cur_s, cur_e, gaps = 0, seg[0][1], seg[0][2], []
        for name, st, en in seg[1:]:
            if st > cur_e:
                gaps.append(((st - cur_e) / 1e3, name.split("(")[0][:40]))
                busy += cur_e - cur_s; cur_s, cur_e = st, en
            else:
                cur_e = max(cur_e, en)
        busy += cur_e - cur_s
        tail = (t1 - cur_e) / 1e3
        big = sorted(gaps, reverse=True)[:6]
        print("segment %.3f ms: busy %.3f ms, idle %.3f ms in %d gaps + %.3f ms before the next %s; largest gaps (us, before): %s"
              % ((t1 - t0) / 1e6, busy / 1e6, sum(g for g, _ in gaps) / 1e3, len(gaps), tail / 1e3, sub, ", ".join("%.0f %s" % g for g in big)))
elif mode == "timeline":
    sub, cnt = sys.argv[3], int(sys.argv[4])
    rows = [r for r in c.execute("select name, start, end, stream_id, grid_x, grid_y from kernels order by start") if sub in r[0]]
    sel = rows[-(cnt + 30):-30] if len(rows) > cnt + 30 else rows
    t0 = sel[0][1]
    for r in sel:
        print("%-24s stream %s  start %9.1f us  duration %7.1f us  grid %s x %s" % (r[0].split("(")[0][:24], r[3], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[4], r[5]))
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

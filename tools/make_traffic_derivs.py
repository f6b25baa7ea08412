#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_baseline_derivs.sh (pmc_fetch_size_derivs_n*_B*.txt, pmc_write_size_*) into
profiles/traffic_derivs.json: HBM-side bytes per launch and per walker of the derivative kernels next to their algorithmic bytes
(inputs + outputs of the call).  bench.py attaches these to `shapes.*` / `update_path` of the driver-timed record.
   python tools/make_traffic_derivs.py gpurun_out/<TAG>/derivs profiles/traffic_derivs.json [source note]"""
import glob, json, os, re, sys

root, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.path.normpath(root))
P = 1074                                                       # flow parameters of the (2, 16, 16) depth-2 net


def algorithmic(kind, n):
    N = 2 * n
    if kind == "grad_laplacian":                               # x, state_idx, v in; grad (complex), lap (complex) out
        return 8 * N + 4 * n + 8 * N + 16 * N + 16
    if kind == "grad_laplacian_scores":                        # the fused call: one set of inputs, both sets of outputs
        return 8 * N + 4 * n + 8 * N + 16 * N + 16 + 16 * P
    return 8 * N + 4 * n + 16 * P                              # x, state_idx in; the (P, 2) score row out


def parse(path):
    res, cur = {}, None
    for line in open(path):
        m = re.match(r"void (k_\w+)<([^>]*)>", line)
        if m:
            cur = m.group(1); continue
        m = re.match(r"\s+workgroup (\d+)\s+grid (\d+)", line)
        if m and cur:
            res.setdefault(cur, {})["walkers_per_launch"] = int(m.group(2)) // int(m.group(1)); continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=(\d+) mean=([0-9.e+]+)", line)
        if m and cur:
            res.setdefault(cur, {})[m.group(1)] = float(m.group(3))
    return res


outj = {"note": "FETCH_SIZE / WRITE_SIZE (KB per launch) from separate rocprofv3 --pmc passes of tools/deriv_timing.py (%s), reported uncorrected "
                "(MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request of wide streaming reads, i.e. up to 2x more bytes moved; Infinity-Cache "
                "hits are counted); algorithmic bytes = inputs + outputs of the call per walker" % note, "sizes": {}}
for f in sorted(glob.glob(os.path.join(root, "pmc_fetch_size_derivs_n*_B*.txt"))):
    m = re.search(r"n(\d+)_B(\d+)", f)
    n, B = int(m.group(1)), int(m.group(2))
    fe = parse(f); wr = parse(f.replace("fetch_size", "write_size"))
    row = {}
    for k in fe:
        if k not in wr or "FETCH_SIZE" not in fe[k] or "WRITE_SIZE" not in wr[k]:
            continue
        kind = "grad_laplacian_scores" if ("lap" in k and "scores" in k) else ("grad_laplacian" if ("lap" in k) else "scores")
        w = fe[k]["walkers_per_launch"]
        tot = (fe[k]["FETCH_SIZE"] + wr[k]["WRITE_SIZE"]) * 1024.0
        alg = algorithmic(kind, n)
        row[kind] = {"kernel": k, "walkers_per_launch": w, "FETCH_SIZE_KB_per_launch": fe[k]["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": wr[k]["WRITE_SIZE"],
                     "traffic_bytes_per_walker": tot / w, "algorithmic_bytes_per_walker": alg, "traffic_over_algorithmic": tot / w / alg}
    outj["sizes"]["n%d" % n] = dict(row, batch=B)
json.dump(outj, open(out, "w"), indent=1)
print(json.dumps(outj, indent=1))

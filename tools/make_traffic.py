#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`bench.py` into profiles/traffic_<tag>.json, which bench.py reports as roofline.traffic (HBM bytes per k_mcmc launch).
usage: make_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json> <n> <batch> <mc_steps>"""
import csv, glob, json, sys

def mean_counter(root, name):
    f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and "k_mcmc" in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)

fetch, nf = mean_counter(sys.argv[1], "FETCH_SIZE")
write, nw = mean_counter(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "k_mcmc", "n": int(sys.argv[4]), "batch": int(sys.argv[5]), "mc_steps": int(sys.argv[6]),
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "launches_averaged": [nf, nw],
       "bytes_per_launch": (fetch + write) * 1024.0,
       "note": "FETCH_SIZE/WRITE_SIZE (KB) from separate rocprofv3 --pmc passes, reported uncorrected: the kernel reads x, state_idx and "
               "8.6 KB of flow parameters and writes x, logp once per CHAIN (no register-spill scratch any more), so the bytes are "
               "the walker arrays themselves: (2*8*n*d + 4*n + 8) bytes per walker = 3.9 MB at n=13, B=8192, plus table refills"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)

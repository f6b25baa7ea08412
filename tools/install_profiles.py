#!/usr/bin/env python3
"""Copies what tools/gpu_profiles.sh TAG left in gpurun_out/TAG (merged back by gpurun) into profiles/ as TAG_<name>: the summaries
the docs cite.  Logs and raw rocprofv3 directories stay behind; the per-size epoch breakdowns are merged into one file; the k_mcmc
traffic file bench.py reads (profiles/traffic_n13_B8192.json) is refreshed from the FETCH / WRITE / SQ passes.
   python tools/install_profiles.py r03e"""
import glob, json, os, re, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
keep = ("launch_shape_sweep", "sampler_bound_experiments")
for f in glob.glob(os.path.join(dst, tag + "_*")):
    if not any(k in f for k in keep):
        os.remove(f)
for sub, prefix in (("", tag + "_"), ("derivs", tag + "_derivs_")):
    d = os.path.join(src, sub)
    for f in sorted(os.listdir(d)):
        p = os.path.join(d, f)
        if os.path.isdir(p) or f.endswith((".log", ".err")) or f.startswith("epoch_breakdown_n"):
            continue
        shutil.copy(p, os.path.join(dst, prefix + f))
with open(os.path.join(dst, tag + "_epoch_breakdowns.txt"), "w") as o:
    for n, B in ((13, 8192), (29, 2048), (49, 512), (57, 512)):
        if not os.path.exists(os.path.join(src, "epoch_breakdown_n%d.txt" % n)):
            continue
        o.write("== n=%d B=%d (tools/epoch_breakdown.py %d %d)\n" % (n, B, n, B))
        o.write(open(os.path.join(src, "epoch_breakdown_n%d.txt" % n)).read() + "\n")


def counter(fname, name):
    for line in open(os.path.join(dst, fname)):
        m = re.match(r"\s+%s\s+dispatches=\d+ mean=([0-9.e+]+)" % name, line)
        if m:
            return float(m.group(1))
    raise SystemExit("counter %s not found in %s" % (name, fname))


if os.path.exists(os.path.join(src, "traffic_derivs.json")):          # bench.py reads profiles/traffic_derivs.json (update_path / shapes)
    shutil.copy(os.path.join(src, "traffic_derivs.json"), os.path.join(dst, "traffic_derivs.json"))
tf = os.path.join(dst, "traffic_n13_B8192.json")
t = json.load(open(tf))
t["FETCH_SIZE_KB_per_launch"] = counter(tag + "_pmc_fetch_size_k_mcmc_n13_B8192.txt", "FETCH_SIZE")
t["WRITE_SIZE_KB_per_launch"] = counter(tag + "_pmc_write_size_k_mcmc_n13_B8192.txt", "WRITE_SIZE")
t["bytes_per_launch"] = (t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024.0
t["sq_insts_mfma_per_launch"] = counter(tag + "_pmc_sq_insts_valu_k_mcmc_n13_B8192.txt", "SQ_INSTS_MFMA")
t["sq_insts_valu_per_launch"] = counter(tag + "_pmc_sq_insts_valu_k_mcmc_n13_B8192.txt", "SQ_INSTS_VALU")
t["note"] = re.sub(r"profiles/r\d\d[a-z]_pmc", "profiles/%s_pmc" % tag, t["note"])
json.dump(t, open(tf, "w"), indent=1)
print("installed:", len(glob.glob(os.path.join(dst, tag + "_*"))), "files;", "k_mcmc traffic %.0f B per launch" % t["bytes_per_launch"])

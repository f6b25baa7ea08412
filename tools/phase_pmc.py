#!/usr/bin/env python3
"""Diagnostic: per-dispatch PMC values for the phases of one logp evaluation.  Run under
   rocprofv3 --pmc ... --output-format csv -d OUT -- python3 tools/phase_pmc.py run
then `python3 tools/phase_pmc.py show OUT`.  Dispatch order: flow primal, primal+jacobian, full logp."""
import sys, os, csv, glob, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np
    from bench import synthetic
    from coulombgas_amd.engine import Engine
    from coulombgas_amd._lib import lib, check
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    Emax = {13: 25, 29: 25, 57: 49}[n]
    L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
    eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta); eng.device_mode(True)
    N = n * 2
    d_x = eng.alloc((B, n, 2)).upload(x); d_s = eng.alloc((B, n), np.int32).upload(sidx)
    d_z = eng.alloc((B, n, 2)); d_J = eng.alloc((B, N, N)); d_lp = eng.alloc((B,))
    check(lib().cg_flow_forward(eng._ctx, d_x.ptr, B, d_z.ptr), eng._ctx)
    check(lib().cg_flow_jacobian(eng._ctx, d_x.ptr, B, d_J.ptr), eng._ctx)
    check(lib().cg_logp(eng._ctx, d_x.ptr, d_s.ptr, B, d_lp.ptr), eng._ctx)
    eng.sync()
else:
    files = glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True)
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        if "k_logpsi" not in r["Kernel_Name"] and "k_mcmc" not in r["Kernel_Name"]:
            continue
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    for d, c in rows.items():
        print("dispatch", d, " ".join("%s=%.0f" % (k, v / B) for k, v in sorted(c.items())), "(per walker)")

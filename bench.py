#!/usr/bin/env python3
"""bench.py -- walker-steps/s of the VMC sampling call on MI355X (BASELINE.json metric).

One "step" = one sampling call of main.py:335 without the Transformer sampler (state_indices supplied):
mc_steps Metropolis sweeps over B walkers = mc_steps+1 evaluations of logp (src/MCMC.py:36-37), as ONE launch of
the k_mcmc kernel.  Workload at N=1: BASELINE.json configs[1]: n=13 dim=2 rs=10 Theta=0.15 batch=8192 mc_steps=50.
With --gpus N the walker batch is sharded (8192 walkers PER GPU, weak scaling, independent chains with distinct
Philox streams); the only collective is the scalar accept-rate mean (src/MCMC.py:39) over RCCL.

Launch: `python bench.py --gpus 1 --steps K --warmup W` or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
(any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT will do: bench.py itself does not import
torch.distributed -- the RCCL id travels over one TCP exchange, barrier and MAX over ranks through cg_allreduce_mean).
Rank 0 prints ONE JSON line.
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_WALKER_STEP = {13: 0.558e6, 29: 2.80e6, 49: 8.32e6, 57: 11.4e6}     # SURVEY 8(d) structured count (n = 49: the same formulas, tools/flop_count.py)
BYTES_PER_WALKER_STEP = {13: 484.0, 29: 1060.0, 49: 1780.0, 57: 2068.0}        # SURVEY 8(d) (36 n + 16), RNG in-kernel
HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synthetic(n, dim, B, Emax, rank):
    """SURVEY 8(d) synthetic inputs (coulombgas_amd/synthetic.py)."""
    from coulombgas_amd.synthetic import bench_inputs
    return bench_inputs(n, dim, B, Emax, rank)


def usable_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box gives a 1-GPU job
    a share of the host, e.g. cpu.max = 16 CPUs of 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


DENSE_AD_FLOPS_PER_WALKER_STEP = {13: 2.7e6, 29: 24e6, 57: 170e6}   # what XLA / the C port execute (dense jacfwd)


def cpu_baseline(n, dim, L, sp, theta, sidx, x, mc_steps, stddev, budget_s=24.0, reps=5):
    """Times the CPU restatement (oracle/cg_oracle.c: the algorithm the reference executes -- dense forward-mode Jacobian with
    n*d tangents, SIMD over the tangents, two LU log-dets; OpenMP over walkers) on the host cores: BASELINE.md section 3 --
    config 1's shape (B = 128, the full call) and a bounded sample of the timed workload's shape, each the MEDIAN of `reps`
    repetitions after one warm-up.  `value` is the larger-batch rate (the shape of the metric)."""
    import ctypes as C
    from coulombgas_amd.build import build_oracle
    path = build_oracle()
    if not path or not os.path.exists(path):
        raise RuntimeError("the C oracle (oracle/cg_oracle.c) could not be built")
    lib = C.CDLL(path)
    lib.cgo_mcmc.restype = C.c_double
    lib.cgo_num_threads.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.cgo_set_num_threads(C.c_int(usable_cores()))     # one OpenMP thread per core this process may use
    cores = lib.cgo_num_threads()

    def run(Bs, steps):
        xs = np.ascontiguousarray(x[:Bs]).copy(); ss = np.ascontiguousarray(sidx[:Bs])
        rng = np.random.default_rng(7)
        noise = rng.standard_normal((steps, Bs, n, dim)); unif = rng.uniform(size=(steps, Bs))
        logp = np.empty(Bs)
        t0 = time.perf_counter()
        lib.cgo_mcmc(C.c_int(n), C.c_int(dim), C.c_int(2), C.c_int(16), C.c_int(16), C.c_double(L), p(theta), p(sp), C.c_int(sp.shape[0]),
                     p(ss), p(xs), C.c_int(Bs), C.c_int(steps), C.c_double(stddev), p(noise), p(unif), p(logp))
        return time.perf_counter() - t0

    def median_rate(Bs, steps):
        run(Bs, steps)                                   # warm-up
        ts = sorted(run(Bs, steps) for _ in range(reps))
        return Bs * steps / ts[len(ts) // 2], ts

    Bc = min(4 * cores, x.shape[0])
    run(Bc, 3)                                           # thread team start-up, page faults
    rate = Bc * 3 / run(Bc, 3)                           # calibration
    per_run = budget_s / (2 * (reps + 1))
    # config 1's shape: B = 128 (the whole call if it fits the budget, otherwise fewer Metropolis steps of it)
    B1 = min(128, x.shape[0]); s1 = int(max(1, min(mc_steps, rate * per_run / B1)))
    r1, t1 = median_rate(B1, s1)
    # the timed workload's shape: as many of its walkers as the budget allows (a multiple of the core count), all mc_steps
    B2 = int(min(x.shape[0], max(cores, rate * per_run / mc_steps // cores * cores))); s2 = mc_steps
    r2, t2 = median_rate(B2, s2)
    fl = DENSE_AD_FLOPS_PER_WALKER_STEP.get(n)
    try:                                                 # SURVEY 8(d)'s optional second baseline needs JAX on the box
        import importlib.util
        jax_here = importlib.util.find_spec("jax") is not None
    except Exception:                                    # noqa: BLE001
        jax_here = False
    return {"value": r2, "unit": "walker-steps/s", "cores": int(cores), "kind": "port",
            "jax_on_this_box": jax_here, "jax_note": "the reference's JAX path cannot be timed: JAX / Haiku / Optax are not installed (probed at run time)" if not jax_here else "JAX importable: not used (the reference itself is not on this box)",
            "sample": "median of %d runs: %d walkers x %d mc_steps of the timed workload (n=%d), %.2f s per run" % (reps, B2, s2, n, t2[len(t2) // 2]),
            "config1_shape": {"value": r1, "sample": "median of %d runs: B=%d x %d mc_steps, %.2f s per run" % (reps, B1, s1, t1[len(t1) // 2])},
            "gflops_per_core": (r2 * fl / cores / 1e9) if fl else None,
            "gflops_note": "dense forward-mode count of SURVEY 8(d) (%.3g flop per walker-step at n=%d), the arithmetic this port executes" % (fl or 0, n),
            "runs_s": [round(v, 4) for v in t2]}


def energy_check(eng, n, dim, L, sp, theta, sidx, x, rs=10.0, kappa=10, Gmax=15, n_split=64, n_exact=16):
    """BASELINE.json's second half of the metric: relative error of <E_loc> (src/VMC.py:38-41) of the HIP path vs the CPU
    restatement (oracle/cg_ref.py, torch.func) on identical inputs: walkers the timed chains ended on, in the
    Hutchinson-split mode of BASELINE config 3 (fixed probe, n_split walkers) and with the exact Laplacian (n_exact walkers).
    The oracle is used here as the checker only."""
    import torch
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    t0 = time.perf_counter()
    torch.set_num_threads(usable_cores())                 # torch's default (all host cores) thrashes inside a CPU quota
    G = cg.kpoints(dim, Gmax)
    Vconst = n * rs / L * cg.Madelung(dim, kappa, G)
    eng.set_ewald(kappa, G, rs)
    rflow = R.FermiNet(2, 16, 16, L)
    rparams = R.flow_unravel(R.T(theta), 2, 16, 16, dim)
    r_logpsi = R.make_logpsi(rflow, sp, L)
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(rflow, sp, L)
    out = {"what": "E_loc = -lap - sum grad^2 + Ewald + Vconst, rs=%.1f kappa=%d Gmax=%d; HIP path vs oracle/cg_ref.py" % (rs, kappa, Gmax)}
    v = np.random.default_rng(99).standard_normal(x.shape)
    for name, cnt, mode in (("hutchinson_split", n_split, 2), ("exact", n_exact, 0)):
        if cnt <= 0:
            continue
        xs = np.ascontiguousarray(x[:cnt]); ss = np.ascontiguousarray(sidx[:cnt]); vs = np.ascontiguousarray(v[:cnt])
        g, lap = eng.grad_laplacian(xs, ss, mode, vs if mode else None)
        E = -lap - (g ** 2).sum(axis=(-2, -1)) + eng.ewald(xs) + Vconst
        if mode:
            _, rfn = R.make_logpsi_grad_laplacian(r_logpsi, hutchinson=True, logphi=r_logphi, logjacdet=r_logjacdet)
            gr, lr = rfn(R.T(xs), rparams, torch.as_tensor(ss.astype(np.int64)), R.T(vs))
        else:
            _, rfn = R.make_logpsi_grad_laplacian(r_logpsi)
            gr, lr = rfn(R.T(xs), rparams, torch.as_tensor(ss.astype(np.int64)), None)
        Er = (-lr - (gr ** 2).sum(dim=(-2, -1))).numpy() + R.potential_energy(R.T(xs), kappa, G, L, rs).numpy() + Vconst
        out[name] = {"walkers": int(cnt), "E_mean_gpu": float(E.real.mean()), "E_mean_cpu": float(Er.real.mean()),
                     "rel_err_mean": float(abs(E.real.mean() - Er.real.mean()) / abs(Er.real.mean())),
                     "rel_err_max_per_walker": float(np.abs(E - Er).max() / np.abs(Er).max())}
    out["rel_err_mean"] = max(out[k]["rel_err_mean"] for k in ("hutchinson_split", "exact") if k in out)
    out["seconds"] = time.perf_counter() - t0
    return out


def update_path_extras(eng, n, dim, L, sp, theta, sidx, x, peak_tflops, B_epoch=None, mc_steps=50):
    """OUTSIDE the timed region, extra keys of the JSON line: the update path of BASELINE config 3 (full flow / Slater / log Psi
    path with the Hutchinson-split Laplacian, main.py:344) -- HIP-event time of each derivative kernel on the walkers the timed
    chains ended on (device-resident inputs), their structured flop counts (tools/flop_count.py) against the measured fp64 peak,
    and the wall time of whole SR epochs (sampling + observables + gradient + Fisher matrix + damped solve, coulombgas_amd.train)."""
    import coulombgas_amd as cg
    from coulombgas_amd.engine import DeviceArray
    from coulombgas_amd._lib import lib
    t0 = time.perf_counter()
    B = x.shape[0]
    x_d = DeviceArray.from_numpy(eng, x); s_d = DeviceArray.from_numpy(eng, sidx, np.int32)
    v_d = eng.randn_d("hutch_v", x.shape, 12345)
    P = eng.P
    pack = eng.scratch("fisher_pack", (P * P + 2 * P,))

    def med(fn, reps=3):
        fn(); eng.sync()
        ts = []
        for _ in range(reps):
            eng.timer_start(); fn(); ts.append(eng.timer_stop())
        return sorted(ts)[len(ts) // 2]

    def scores():
        x_d.version += 1                                   # (defeats the engine's score cache: time the kernel, not the look-up)
        eng.scores_compute_d(x_d, s_d)
    k = {"grad_laplacian (k_grad_lap2, Hutchinson-split)": med(lambda: eng.grad_laplacian_d(x_d, s_d, 2, v_d)),
         "grad_laplacian (k_grad_lap2, exact Laplacian: n d basis jet passes, src/logpsi.py:63-106)": med(lambda: eng.grad_laplacian_d(x_d, s_d, 0, None), reps=1),
         "scores (k_scores)": med(scores),
         "quantum Fisher matrix + mean score (k_fisher, reductions)": med(lambda: eng.scores_fisher_d(pack, 0, P * P))}

    def fused():
        x_d.version += 1
        eng.grad_laplacian_d(x_d, s_d, 2, v_d, with_scores=True)
    k["grad_laplacian + scores in one call (cg_grad_laplacian_scores: what an optimisation step runs; the set-up of the two is shared)"] = med(fused)
    eng.set_ewald(10, cg.kpoints(dim, 15), 10.0)
    k["ewald (k_ewald)"] = med(lambda: eng.ewald_d(x_d))
    b = np.random.default_rng(5).standard_normal(P)
    work = eng.scratch("bench_solve_in", (P * P,))
    def solve():
        eng.axpby_d(1.0, pack, 0.0, work, count=P * P)
        eng.scale_d(work, 1.0)                               # (keeps the call sequence of sr._solve_and_clip: copy, then factor + solves)
        eng.spd_solve_d(work, b, damping=1e-3)
    k["damped solve (cg_spd_solve: centring, Cholesky, triangular solves)"] = med(solve)
    out = {"what": "update path, n=%d B=%d (not part of `value`): HIP-event ms per call, device-resident inputs" % (n, B), "kernel_ms": k}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from flop_count import grad_lap_flops, scores_flops
        fl = {"grad_laplacian (k_grad_lap2, Hutchinson-split)": grad_lap_flops(n, dim, mode=2), "scores (k_scores)": scores_flops(n, dim),
              "grad_laplacian (k_grad_lap2, exact Laplacian: n d basis jet passes, src/logpsi.py:63-106)": grad_lap_flops(n, dim, mode=0)}
        out["roofline"] = {kk: {"flop_per_walker": fl[kk], "achieved_tflops": fl[kk] * B / (k[kk] * 1e-3) / 1e12,
                                "frac_of_fp64_peak": fl[kk] * B / (k[kk] * 1e-3) / 1e12 / peak_tflops} for kk in fl}
        out["roofline"]["note"] = "structured flop counts of tools/flop_count.py (FMA = 2 flops, transcendentals not counted), peak = measured fp64 rate %.1f TFLOP/s" % peak_tflops
        tr = derivative_traffic(n)
        if tr:
            out["traffic"] = tr
    except Exception as e:                                   # noqa: BLE001 -- a reporting extra must not lose the metric line
        out["roofline"] = {"error": repr(e)}
    # whole SR epochs through the driver (main.py:316-346 mirror): zero-temperature sampler, hybrid Fisher SR, walkers in HBM
    flow = cg.FermiNet(2, 16, 16, L)
    p0 = flow.unravel(theta, dim)
    samp = cg.GroundStateSampler(n, sp.shape[0])
    marks = [time.perf_counter()]
    rows = []
    def log(row):
        marks.append(time.perf_counter()); rows.append(row)
    cg.train(flow, p0, sp, n, dim, L, rs=10.0, beta=1 / (4 * 0.15), batch=B_epoch or B, epochs=5, sampler=samp, log_prob=samp.log_prob,
             sr=(1e-3, 1e-3), mc_therm=1, mc_steps=mc_steps, seed=3, log=log)
    ep = sorted(np.diff(marks)[2:] * 1e3)                    # (epoch 1 includes thermalisation, epoch 2 first-use allocations)
    out["epoch_ms"] = float(ep[len(ep) // 2])
    out["epoch"] = "median wall time of epochs 3-5 of coulombgas_amd.train: batch %d, mc_steps %d, Hutchinson-split, SR damping 1e-3" % (B_epoch or B, mc_steps)
    out["last_row"] = rows[-1]
    # finite-temperature epochs (main.py:152-164, 277-307 mirror): the autoregressive Transformer density matrix of the shipped runs
    # (2 layers, model size 16, 4 heads, hidden 32; random-initialised) sampled, differentiated and trained on the device beside the flow:
    # sampling + k_van, observables, both gradients, both Fisher matrices (P_van x P_van classical), both damped solves
    try:
        van = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
        pv0 = van.init(5, sp[:n])
        sampler, log_prob = cg.make_autoregressive_sampler(van, sp, n, sp.shape[0])
        marks2 = [time.perf_counter()]
        cg.train(flow, p0, sp, n, dim, L, rs=10.0, beta=1 / (4 * 0.15), batch=B_epoch or B, epochs=5, sampler=sampler, log_prob=log_prob,
                 params_van=pv0, sr=(1e-3, 1e-3), mc_therm=1, mc_steps=mc_steps, seed=4, log=lambda row: marks2.append(time.perf_counter()))
        ep2 = sorted(np.diff(marks2)[2:] * 1e3)
        out["hybrid_epoch_ms"] = float(ep2[len(ep2) // 2])
        out["hybrid_epoch"] = ("median wall time of epochs 3-5 with the density-matrix Transformer trained too: P_van = %d parameters, "
                               "classical Fisher matrix + second damped solve on the device" % lib().cg_van_num_params(sp.shape[0], 2, 16, 4, 32, dim))
    except Exception as e:                                   # noqa: BLE001 -- a reporting extra must not lose the metric line
        out["hybrid_epoch_ms"] = None; out["hybrid_epoch"] = "failed: %r" % (e,)
    out["seconds"] = time.perf_counter() - t0
    return out


def derivative_traffic(n):
    """profiles/traffic_derivs.json (tools/make_traffic_derivs.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the
    committed tree): HBM-side bytes per walker of the two derivative kernels next to the algorithmic bytes of the call."""
    path = os.path.join(ROOT, "profiles", "traffic_derivs.json")
    try:
        tj = json.load(open(path))["sizes"].get("n%d" % n)
    except (OSError, ValueError, KeyError):
        return None
    if not tj:
        return None
    out = {}
    for kind in ("grad_laplacian", "scores", "grad_laplacian_scores"):
        if kind in tj:
            out[kind] = {k: tj[kind][k] for k in ("kernel", "traffic_bytes_per_walker", "algorithmic_bytes_per_walker", "traffic_over_algorithmic")}
    out["source"] = "profiles/traffic_derivs.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, per launch / walkers per launch)"
    return out


def production_shapes(peak_tflops, mc_steps=50, stddev=0.1):
    """OUTSIDE the timed region, extra key `shapes`: BASELINE configs 4 and 5 and the third size the reference published runs for
    (data/n_29_*, n_49_*, n_57_*) at their per-GPU batch: one sampling call (k_mcmc, HIP events), the two derivative kernels and whole SR
    epochs of coulombgas_amd.train, so that the driver-timed record carries them too (profiles/ holds the same numbers per round)."""
    import coulombgas_amd as cg
    from coulombgas_amd.engine import Engine, DeviceArray
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from flop_count import grad_lap_flops, scores_flops
    out = {}
    for key, n, B, Emax, B_exact in (("n29", 29, 2048, 25, 256), ("n49", 49, 512, 36, 0), ("n57", 57, 512, 49, 0)):
        t0 = time.perf_counter()
        L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
        eng = Engine(n, 2, 2, 16, 16, L, sp)
        try:
            eng.set_params(theta); eng.device_mode(True)
            d_x = eng.alloc((B, n, 2)).upload(x); d_s = eng.alloc((B, n), np.int32).upload(sidx); d_lp = eng.alloc((B,))
            ts = []
            for it in range(5):                                # two thermalisation rounds, three timed calls
                eng.timer_start()
                eng.mcmc_dev(d_x, d_s, B, mc_steps, stddev, seed=77 + it, walker_offset=0, logp_buf=d_lp)
                ts.append(eng.timer_stop()); eng.wrap_dev(d_x, B)
            k_ms = sorted(ts[2:])[1]
            fl = FLOPS_PER_WALKER_STEP.get(n)                  # (SURVEY 8(d) counts n = 13 / 29 / 57)
            where = {29: "per-GPU shape of BASELINE config 4", 57: "per-GPU shape of BASELINE config 5"}.get(n, "per-GPU shape of the reference's data/n_%d_* runs" % n)
            r = {"workload": "n=%d dim=2 Emax=%d batch=%d mc_steps=%d (%s)" % (n, Emax, B, mc_steps, where),
                 "kernel_ms": k_ms, "walker_steps_per_s": B * mc_steps / (k_ms * 1e-3),
                 "frac": fl * B * mc_steps / (k_ms * 1e-3) / 1e12 / peak_tflops if fl else None, "finite": bool(np.isfinite(d_lp.download()).all())}
            eng.device_mode(False)
            x_d = DeviceArray.from_numpy(eng, d_x.download()); s_d = DeviceArray.from_numpy(eng, sidx, np.int32)
            v_d = eng.randn_d("hutch_v", x.shape, 4321)

            def med(fn, reps=3):
                fn(); eng.sync(); tt = []
                for _ in range(reps):
                    eng.timer_start(); fn(); tt.append(eng.timer_stop())
                return sorted(tt)[len(tt) // 2]

            def scores():
                x_d.version += 1
                eng.scores_compute_d(x_d, s_d)
            gl = med(lambda: eng.grad_laplacian_d(x_d, s_d, 2, v_d)); sc = med(scores)
            r["grad_laplacian_ms"] = gl; r["scores_ms"] = sc

            def fused():                                       # the two in one call, as an optimisation step runs them (shared set-up)
                x_d.version += 1
                eng.grad_laplacian_d(x_d, s_d, 2, v_d, with_scores=True)
            r["grad_laplacian_scores_ms"] = med(fused)
            r["grad_laplacian_frac"] = grad_lap_flops(n, 2, mode=2) * B / (gl * 1e-3) / 1e12 / peak_tflops
            r["scores_frac"] = scores_flops(n, 2) * B / (sc * 1e-3) / 1e12 / peak_tflops
            tr = derivative_traffic(n)                         # HBM-side bytes per walker from the committed --pmc passes (not measured in this run)
            if tr:
                r["traffic"] = tr
            if B_exact:                                        # the reference's default (exact) Laplacian, on a slice of the batch
                xe = DeviceArray.from_numpy(eng, x_d.numpy()[:B_exact]); se = DeviceArray.from_numpy(eng, sidx[:B_exact], np.int32)
                ge = med(lambda: eng.grad_laplacian_d(xe, se, 0, None), reps=1)
                r["grad_laplacian_exact"] = {"walkers": B_exact, "ms": ge,
                                             "frac": grad_lap_flops(n, 2, mode=0) * B_exact / (ge * 1e-3) / 1e12 / peak_tflops}
        finally:
            eng.close()
        flow = cg.FermiNet(2, 16, 16, L)
        marks = [time.perf_counter()]
        cg.train(flow, flow.unravel(theta, 2), sp, n, 2, L, rs=10.0, beta=1 / (4 * 0.15), batch=B, epochs=5, sampler=cg.GroundStateSampler(n, sp.shape[0]),
                 log_prob=cg.GroundStateSampler(n, sp.shape[0]).log_prob, sr=(1e-3, 1e-3), mc_therm=1, mc_steps=mc_steps, seed=3,
                 log=lambda row: marks.append(time.perf_counter()))
        ep = sorted(np.diff(marks)[2:] * 1e3)
        r["epoch_ms"] = float(ep[len(ep) // 2])
        r["seconds"] = time.perf_counter() - t0
        out[key] = r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 100 x 11 ms: long enough for an outside GPU-activity sampler to see the timed region
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=13)
    ap.add_argument("--batch", type=int, default=8192, help="walkers per GPU")
    ap.add_argument("--mc_steps", type=int, default=50)
    ap.add_argument("--mc_stddev", type=float, default=0.1)
    ap.add_argument("--Emax", type=int, default=25)
    ap.add_argument("--threads", type=int, default=0, help="threads per walker workgroup (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-energy-check", action="store_true")
    ap.add_argument("--no-update-extras", action="store_true", help="skip the (untimed) SR-epoch / derivative-kernel report")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch one process per GPU (e.g. torch.distributed.run)" % (args.gpus, world), file=sys.stderr)
            sys.exit(2)
    force_dist = os.environ.get("CG_FORCE_DIST") == "1"        # exercise the N > 1 code path on a 1-GPU box

    from coulombgas_amd import utils
    from coulombgas_amd.engine import Engine
    from coulombgas_amd.comm import RcclComm, NullComm
    utils.set_device(local)
    n, dim, B = args.n, 2, args.batch
    L, sp, theta, sidx, x = synthetic(n, dim, B, args.Emax, rank)
    eng = Engine(n, dim, 2, 16, 16, L, sp, device=local)
    eng.set_params(theta)
    if args.threads:
        eng.set_block_threads(args.threads)
    comm_kind = "none"
    if world > 1 or force_dist:
        # the library's own RCCL path (cg_comm_* / cg_allreduce_mean) or nothing: a broken communicator must fail the run
        comm = RcclComm(eng, rank, world); comm_kind = "rccl via cg_allreduce_mean"
    else:
        comm = NullComm()

    eng.device_mode(True)
    d_x = eng.alloc((B, n, dim)).upload(x)
    d_s = eng.alloc((B, n), np.int32).upload(sidx)
    d_lp = eng.alloc((B,))
    d_acc = eng.alloc((16,))

    def barrier():                                             # every rank here, every stream drained (a one-element all-reduce)
        eng.sync()
        comm.barrier()

    def sampling_call(it):
        eng.mcmc_dev(d_x, d_s, B, args.mc_steps, args.mc_stddev, seed=42 + it, walker_offset=rank * B, logp_buf=d_lp)
        eng.wrap_dev(d_x, B)                                   # src/VMC.py:24
        if world > 1 or force_dist:                            # src/MCMC.py:39: pmean of the accept rate, formed and reduced on the device
            return comm.accept_rate(eng, args.mc_steps * B)
        return None

    for it in range(args.warmup):                              # thermalisation rounds (main.py:241-246)
        sampling_call(it)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for it in range(args.steps):
        eng.timer_start()
        eng.mcmc_dev(d_x, d_s, B, args.mc_steps, args.mc_stddev, seed=4242 + it, walker_offset=rank * B, logp_buf=d_lp)
        kernel_ms += eng.timer_stop()                          # HIP events on the kernel's own stream
        eng.wrap_dev(d_x, B)
        if world > 1 or force_dist:
            comm.accept_rate(eng, args.mc_steps * B)
    barrier()
    elapsed = time.perf_counter() - t0
    accept = eng.mcmc_accepts() / float(args.mc_steps * B)
    elapsed = comm.pmax(elapsed)                               # MAX over the ranks

    # sanity of the timed state: finite log-probabilities (a NaN chain would still "run fast")
    lp = d_lp.download()
    ok = bool(np.isfinite(lp).all())
    x_final = d_x.download()

    if rank == 0:
        walker_steps = float(B) * args.mc_steps * args.steps * world
        value = walker_steps / elapsed
        k_avg_s = kernel_ms / args.steps * 1e-3
        fl = FLOPS_PER_WALKER_STEP.get(n)
        by = BYTES_PER_WALKER_STEP.get(n)
        eng.device_mode(False)
        peak_fma = eng.microbench_fp64(0)
        peak_mfma = eng.microbench_fp64(1)
        peak = max(peak_fma, peak_mfma)
        ach = (fl * B * args.mc_steps / k_avg_s / 1e12) if fl else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_n%d_B%d.json" % (n, B))
        if os.path.exists(tpath):               # HBM bytes per launch from the committed rocprofv3 --pmc passes
            tj = json.load(open(tpath))
            if tj.get("mc_steps") == args.mc_steps:
                traffic = tj["bytes_per_launch"]
        mfma_frac = None
        if traffic is not None and tj.get("sq_insts_mfma_per_launch"):      # f64 MFMA share, from the committed SQ counter pass
            mfma_frac = tj["sq_insts_mfma_per_launch"] * 2048.0 / k_avg_s / 1e12 / peak_mfma
        roofline = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": (ach / peak) if ach else None, "traffic": traffic,
                    "binds": "fp64 VALU / transcendental issue (compute label of the schema: the matrix cores carry only the dense layers, see mfma_frac; HBM: see hbm)",
                    "mfma_frac": mfma_frac,
                    "kernel": "k_mcmc", "kernel_avg_ms": k_avg_s * 1e3,
                    "note": "fp64 kernel: peak = measured v_fma_f64 / v_mfma_f64_16x16x4 rate on this GPU (%.1f / %.1f TFLOP/s); "
                            "achieved = SURVEY 8(d) algorithmic %.3g flop/walker-step x %d walker-steps per launch / HIP-event kernel time"
                            % (peak_fma, peak_mfma, fl or 0, B * args.mc_steps),
                    "hbm": {"achieved": (by * B * args.mc_steps / k_avg_s / 1e9) if by else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": (by * B * args.mc_steps / k_avg_s / 1e9 / HBM_PEAK_GBS) if by else None}}
        cpu = None
        energy = None
        update = None
        shapes = None
        side_errors = []
        if world == 1 and not args.no_update_extras:
            try:                               # the metric line must survive a failing extra (reported, exit code 1 afterwards)
                update = update_path_extras(eng, n, dim, L, sp, theta, sidx, x_final, peak, mc_steps=args.mc_steps)
            except Exception as e:             # noqa: BLE001
                update = {"error": repr(e)}; side_errors.append("update_path")
        if world == 1 and not args.no_update_extras and n == 13:
            try:
                shapes = production_shapes(peak, mc_steps=args.mc_steps, stddev=args.mc_stddev)
            except Exception as e:             # noqa: BLE001
                shapes = {"error": repr(e)}; side_errors.append("shapes")
        if world == 1 and not args.no_cpu_baseline:
            try:
                _, _, _, sidx0, x0 = synthetic(n, dim, B, args.Emax, 0)
                cpu = cpu_baseline(n, dim, L, sp, theta, sidx0, x0, args.mc_steps, args.mc_stddev)
            except Exception as e:             # noqa: BLE001 -- checker / baseline trouble (gcc, oracle build): keep the measured line
                cpu = {"error": repr(e)}; side_errors.append("cpu_baseline")
            if not args.no_energy_check:
                try:
                    big = n > 16               # the torch oracle is minutes per walker beyond n = 13: fewer walkers there
                    energy = energy_check(eng, n, dim, L, sp, theta, sidx, x_final, n_split=4 if big else 64, n_exact=0 if big else 8)
                except Exception as e:         # noqa: BLE001
                    energy = {"error": repr(e)}; side_errors.append("energy")
        out = {"metric": "walker-steps/sec (batch x mcsteps/s), n=%d 2D batch %d" % (n, B), "value": value,
               "unit": "walker-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "n=%d dim=2 rs=10.0 Theta=0.15 Emax=%d batch=%d/GPU mc_steps=%d mc_stddev=%.2f: sampling call "
                                      "(MCMC chain incl. flow+Jacobian+Slater logp), in-kernel Philox RNG" % (n, args.Emax, B, args.mc_steps, args.mc_stddev),
                          "walkers_per_gpu": B, "mc_steps": args.mc_steps, "threads_per_walker": eng.launch_info()["threads"],
                          "lds_bytes_per_walker": eng.launch_info()["lds_bytes"]},
               "accept_rate": accept, "finite": ok, "comm": comm_kind, "roofline": roofline, "cpu_baseline": cpu,
               "energy": energy, "update_path": update, "shapes": shapes}
        print(json.dumps(out), flush=True)
        if side_errors:
            print("bench.py: %s failed (see the JSON line)" % ", ".join(side_errors), file=sys.stderr)
            ok = False
    comm.close()
    eng.close()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()

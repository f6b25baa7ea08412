"""The training loop of main.py:216-384 for one process per GPU, on top of the accelerated hot path.

`update` mirrors main.py:263-310 (jacrev of the two loss closures, pmean, accumulation over acc_steps, the final-step
algebra grad - <F> score / grad - <E> score, optimizer step); `train` mirrors the epoch loop (:316-384) including the
thermalisation rounds (:241-246) and the data.txt row format (:367-372).

The variational density matrix is passed as `sampler`, `log_prob` (+ `log_prob_vjp`, the vector-Jacobian product
jax.jacrev(classical_lossfn) needs, and `classical_score_fn` if params_van is to be trained).  With
coulombgas_amd.make_autoregressive_sampler (the reference's Transformer, run on the GPU: csrc/cg_van.hpp) all four come from
one object.
`GroundStateSampler` is the trivial stand-in (zero temperature: every walker in the n lowest orbitals, log_prob = 0)."""
import os
import warnings
import numpy as np
from . import sr as _sr
from .comm import get_comm, allgather
from .checkpoint import ckpt_filename, load_data, save_data, adam_state_from_ckpt
from .vmc import sample_stateindices_and_x, make_loss
from .logpsi import make_logpsi, make_logphi_logjacdet, make_logp, make_logpsi_grad_laplacian, make_quantum_score
from .potential import kpoints, Madelung

DATA_KEYS = ("F_mean", "F2_mean", "E_mean", "E2_mean", "K_mean", "K2_mean", "V_mean", "V2_mean", "S_mean", "S2_mean")


class GroundStateSampler:
    """Zero-temperature stand-in for the autoregressive sampler: the n lowest orbitals of the (reversed, main.py:88-90)
    table, i.e. its last n rows, for every walker; log-probability 0."""
    def __init__(self, n, num_orbitals):
        self.idx = np.arange(num_orbitals - n, num_orbitals, dtype=np.int32)

    def __call__(self, params_van, key, batch):
        return np.tile(self.idx, (batch, 1))

    def log_prob(self, params_van, state_indices):
        return np.zeros(np.shape(state_indices)[0])


def adam(lr, b1=0.9, b2=0.999, eps=1e-8):
    """optax.adam(lr) (main.py:186) for nested dicts / tuples of arrays; None leaves pass through."""
    def tmap(f, *trees):
        t0 = trees[0]
        if t0 is None:
            return None
        if isinstance(t0, dict):
            return {k: tmap(f, *[t[k] for t in trees]) for k in t0}
        if isinstance(t0, (tuple, list)):
            return type(t0)(tmap(f, *ts) for ts in zip(*trees))
        return f(*[np.asarray(t) for t in trees])

    def init_fn(params):
        z = tmap(np.zeros_like, params)
        return {"count": 0, "mu": z, "nu": tmap(np.zeros_like, params)}

    def update_fn(grads, state, params=None):
        c = state["count"] + 1
        mu = tmap(lambda m, g: b1 * m + (1 - b1) * g, state["mu"], grads)
        nu = tmap(lambda v, g: b2 * v + (1 - b2) * g * g, state["nu"], grads)
        upd = tmap(lambda m, v: -lr * (m / (1 - b1 ** c)) / (np.sqrt(v / (1 - b2 ** c)) + eps), mu, nu)
        return upd, {"count": c, "mu": mu, "nu": nu}

    return _sr.GradientTransformation(init_fn, update_fn)


def _tree(f, *trees):
    t0 = trees[0]
    if t0 is None:
        return None
    if isinstance(t0, dict):
        return {k: _tree(f, *[t[k] for t in trees]) for k in t0}
    if isinstance(t0, (tuple, list)):
        return type(t0)(_tree(f, *ts) for ts in zip(*trees))
    return f(*trees)


def _is_device(a):
    return not isinstance(a, np.ndarray) and hasattr(a, "ptr")


def make_update(observable_and_lossfn, optimizer, acc_steps, fishers_fn=None, log_prob_vjp=None, comm=None):
    """main.py:263-310.  Returns update(params_van, params_flow, opt_state, state_indices, x, key, acc, final_step) ->
    (params_van, params_flow, opt_state, acc) with acc the dict of accumulators the reference threads through pmap.
    Fisher matrices that arrive as DeviceArrays are accumulated in HBM (Engine.axpby_d), never on the host."""
    def new_acc():
        return {"data": {k: 0.0 for k in DATA_KEYS}, "grads": None, "scores": None, "fishers": None}

    def acc_fisher(acc_f, f):
        """classical_fisher_acc += ..., quantum_fisher_acc += ..., quantum_score_mean_acc += ... (main.py:285-289); the first
        step COPIES (the engine's output buffers are overwritten by the next accumulation step)."""
        out = []
        for k, b in enumerate(f):
            a = None if acc_f is None else acc_f[k]
            if b is None:
                out.append(None)
            elif _is_device(b):
                if a is None:
                    a = b.eng.scratch("fisher_acc_%d" % k, b.shape)
                    b.eng.axpby_d(1.0, b, 0.0, a)
                else:
                    b.eng.axpby_d(1.0, b, 1.0, a)
                out.append(a)
            else:
                out.append(np.array(b, copy=True) if a is None else a + b)
        return tuple(out)

    def update(params_van, params_flow, opt_state, state_indices, x, key, acc, final_step):
        cm = comm or get_comm()
        acc = acc or new_acc()
        data, classical_lossfn, quantum_lossfn = observable_and_lossfn(params_van, params_flow, state_indices, x, key)
        grad_flow, score_flow = quantum_lossfn.grad(params_flow, reduce=True)          # :278 + the pmean of :280 on the device
        grad_van = score_van = None
        if log_prob_vjp is not None and params_van is not None:                        # :277
            pair = getattr(getattr(log_prob_vjp, "__self__", None), "vjp_pair_d", None) or getattr(log_prob_vjp, "pair_d", None)
            if pair is not None and _is_device(x):
                # device density matrix: clip weights, both weighted score sums and their pmean (:280) without leaving HBM
                classical_lossfn(params_van, values=False)
                buf, unflat = pair(params_van, state_indices, classical_lossfn.weights, classical_lossfn.score_weights)
                cm.pmean_d(buf)
                h = np.asarray(buf); Pv = h.size // 2
                grad_van, score_van = unflat(h[:Pv]), unflat(h[Pv:])
            else:
                classical_lossfn(params_van)
                grad_van = log_prob_vjp(params_van, state_indices, classical_lossfn.weights)
                score_van = log_prob_vjp(params_van, state_indices, classical_lossfn.score_weights)
                flat, unravel = _sr.ravel_pytree({"g": grad_van, "s": score_van})
                tree = unravel(cm.pmean(flat))                                         # :280 (classical part, host model)
                grad_van, score_van = tree["g"], tree["s"]
        grads = grad_flow if grad_van is None else {"v": grad_van, "f": grad_flow}
        scores = score_flow if score_van is None else {"v": score_van, "f": score_flow}
        acc["data"] = {k: acc["data"][k] + data[k] for k in DATA_KEYS}                 # :281-283
        acc["grads"] = grads if acc["grads"] is None else _tree(lambda a, b: a + b, acc["grads"], grads)
        acc["scores"] = scores if acc["scores"] is None else _tree(lambda a, b: a + b, acc["scores"], scores)
        # acc_steps > 1: the ranks' Fisher matrices are accumulated locally and averaged over the ranks ONCE, in the final step (the
        # pmean is linear; one all-reduce of the two matrices per update instead of one per accumulation step)
        defer = acc_steps > 1 and hasattr(fishers_fn, "reduce_accumulated")
        if fishers_fn is not None:                                                     # :285-289
            f = fishers_fn(params_van, params_flow, state_indices, x, reduce=False) if defer else fishers_fn(params_van, params_flow, state_indices, x)
            acc["fishers"] = acc_fisher(acc["fishers"], f)
        if final_step:                                                                 # :291-307
            d = {k: v / acc_steps for k, v in acc["data"].items()}
            g = _tree(lambda a: a / acc_steps, acc["grads"]); s = _tree(lambda a: a / acc_steps, acc["scores"])
            if grad_van is None:
                g_flow = _tree(lambda a, b: a - d["E_mean"] * b, g, s); g_van = None
            else:
                g_van = _tree(lambda a, b: a - d["F_mean"] * b, g["v"], s["v"])
                g_flow = _tree(lambda a, b: a - d["E_mean"] * b, g["f"], s["f"])
            fish = None
            if acc["fishers"] is not None:
                fish = tuple(None if a is None else (a.eng.scale_d(a, 1.0 / acc_steps) if _is_device(a) else a / acc_steps)
                             for a in acc["fishers"])
                if defer:
                    fish = fishers_fn.reduce_accumulated(fish)
            updates, opt_state = optimizer.update((g_van, g_flow), opt_state, params=fish)
            if updates[0] is not None:
                params_van = _sr.apply_updates(params_van, updates[0])
            params_flow = _sr.apply_updates(params_flow, updates[1])
            acc = dict(acc, data=d, grads=(g_van, g_flow))
        return params_van, params_flow, opt_state, acc

    update.new_acc = new_acc
    return update


def format_row(i, data, rs, batch_total, acc_steps, accept_rate):
    """The data.txt row of main.py:352-372 (energies in Ry / rs^2)."""
    out = []
    for k in ("F", "E", "K", "V", "S"):
        m, m2 = data[k + "_mean"], data[k + "2_mean"]
        std = np.sqrt(max(m2 - m * m, 0.0) / (batch_total * acc_steps))
        sc = 1.0 if k == "S" else 1.0 / rs ** 2
        out += [m * sc, std * sc]
    return ("%6d" + "  %.6f" * 10 + "  %.4f") % ((i,) + tuple(out) + (accept_rate,))


def train(flow, params_flow, sp_indices, n, dim, L, rs, beta, batch, epochs, sampler, log_prob, params_van=None,
          optimizer=None, sr=None, kappa=10, Gmax=15, mc_therm=10, mc_steps=50, mc_stddev=0.1, acc_steps=1,
          hutchinson=True, seed=42, log=None, log_prob_vjp=None, classical_score_fn=None, comm=None, device_resident=True,
          ckpt_path=None, ckpt_every=100, epoch_finished=0):
    """main.py:216-384 on one rank.  sr = (damping, max_norm) selects hybrid_fisher_sr (main.py:179-184), otherwise
    `optimizer` (default adam(1e-3)).  Returns (params_van, params_flow, rows) with rows the data.txt lines.
    ckpt_path / ckpt_every / epoch_finished: the checkpoints of main.py:374-381 ({"keys", "x", "params_van", "params_flow",
    "opt_state"}, x with the reference's leading device axis) and the resume of main.py:217-223 (a shipped epoch_*.pkl
    resumes too).  device_resident: walkers, local energies, scores and Fisher matrices stay in HBM for the whole run."""
    cm = comm or get_comm()
    if log_prob_vjp is None and hasattr(log_prob, "vjp"):          # make_autoregressive_sampler's log_prob carries its own
        log_prob_vjp = log_prob.vjp                                 # reverse pass: the density matrix is trained as well
        if hasattr(log_prob, "vjp_pair_d"):
            log_prob_vjp.pair_d = log_prob.vjp_pair_d
        if classical_score_fn is None and hasattr(log_prob, "grad"):
            classical_score_fn = log_prob.grad
    G = kpoints(dim, Gmax)
    Vconst = n * rs / L * Madelung(dim, kappa, G)                                      # :170-171
    logpsi_novmap = make_logpsi(flow, sp_indices, L)
    logphi, logjacdet = make_logphi_logjacdet(flow, sp_indices, L)
    logp = make_logp(logpsi_novmap)
    fishers_fn = None
    if sr is not None:
        fishers_fn, optimizer = _sr.hybrid_fisher_sr(classical_score_fn, make_quantum_score(logpsi_novmap), sr[0], sr[1], comm=cm)
    elif optimizer is None:
        optimizer = adam(1e-3)
    opt_state = optimizer.init((params_van, params_flow))
    ss = np.random.SeedSequence(seed).spawn(cm.world)[cm.rank]                        # :237, one key per device
    eng = flow.engine(n, dim, sp_indices)
    first_epoch = epoch_finished + 1                                                   # main.py:316 in both branches
    load_name = ckpt_filename(epoch_finished, ckpt_path) if ckpt_path is not None else None
    if epoch_finished > 0 and (load_name is None or not os.path.isfile(load_name)):
        # the reference (main.py:217-223) starts from scratch at epoch_finished + 1 in this case; say so, it is rarely meant
        warnings.warn("train: epoch_finished=%d but no checkpoint %s: starting from fresh walkers and parameters at epoch %d"
                      % (epoch_finished, load_name, first_epoch))
    if load_name is not None and epoch_finished > 0 and os.path.isfile(load_name):     # :217-223 resume
        # (ckpt_path must be the same on every rank: the checkpoint all-gather below is a collective)
        ck = load_data(load_name)
        xs = np.asarray(ck["x"], dtype=np.float64)
        if xs.ndim == 4:                                                               # leading device axis of the reference
            if xs.shape[0] != cm.world:
                raise ValueError("checkpoint %s holds walkers of %d devices, this run has %d ranks" % (load_name, xs.shape[0], cm.world))
            x = xs[cm.rank]
        else:
            x = xs
        params_van, params_flow = ck["params_van"], ck["params_flow"]
        st = adam_state_from_ckpt(ck.get("opt_state"))
        if st is not None and isinstance(opt_state, dict):
            if params_van is None and isinstance(st.get("mu"), (tuple, list)) and st["mu"] and st["mu"][0] is not None:
                st = dict(st, mu=(None,) + tuple(st["mu"][1:]), nu=(None,) + tuple(st["nu"][1:]))   # moments of a density matrix this run does not train
            opt_state = st
        elif ck.get("opt_state") is not None and isinstance(opt_state, dict):
            warnings.warn("train: the optimizer state of %s could not be adopted; Adam moments restart at zero" % load_name)
        ks = np.asarray(ck["keys"])
        key = np.random.SeedSequence([int(v) for v in np.atleast_2d(ks)[cm.rank % np.atleast_2d(ks).shape[0]].ravel()])
        thermalise = False
    else:
        rng = np.random.default_rng(ss)
        x = rng.uniform(0.0, L, (batch, n, dim))                                       # :236
        key = ss
        thermalise = True
    from .engine import Engine, DeviceArray
    if device_resident and isinstance(eng, Engine):
        for f in (sampler, log_prob):          # the density-matrix Transformer samples / evaluates on the same GPU
            if hasattr(f, "attach"):
                f.attach(eng)
        x = DeviceArray.from_numpy(eng, x)            # the walkers live in HBM from here on (main.py:239: replicate / shard)
    if thermalise:
        for _ in range(mc_therm):                                                      # :241-246
            key, _, x, _ = sample_stateindices_and_x(key, sampler, params_van, logp, x, params_flow, mc_steps, mc_stddev, L, comm=cm)
    logpsi, lgl = make_logpsi_grad_laplacian(logpsi_novmap, hutchinson=hutchinson, logphi=logphi, logjacdet=logjacdet)   # :254-256
    observable_and_lossfn = make_loss(log_prob, logpsi, lgl, kappa, G, L, rs, Vconst, beta, comm=cm)               # :258-259
    update = make_update(observable_and_lossfn, optimizer, acc_steps, fishers_fn, log_prob_vjp, comm=cm)
    rows = []
    for i in range(first_epoch, epochs + 1):                                           # :316-372
        host_solves = _sr.device_solve_fallbacks()
        acc, accept_acc = update.new_acc(), 0.0
        for a in range(acc_steps):
            key, state_indices, x, accept_rate = sample_stateindices_and_x(key, sampler, params_van, logp, x, params_flow,
                                                                           mc_steps, mc_stddev, L, comm=cm)
            accept_acc += accept_rate
            params_van, params_flow, opt_state, acc = update(params_van, params_flow, opt_state, state_indices, x,
                                                             key.spawn(1)[0], acc, a == acc_steps - 1)
        row = format_row(i, acc["data"], rs, batch * cm.world, acc_steps, accept_acc / acc_steps)
        rows.append(row)
        if log is not None and cm.rank == 0:
            log(row)
        host_solves = _sr.device_solve_fallbacks() - host_solves
        if host_solves and cm.rank == 0:         # (not part of the data.txt row, whose columns are the reference's: main.py:367-372)
            warnings.warn("train: epoch %d: %d damped solve(s) were not positive definite on the device and went to the host's "
                          "symmetric-indefinite solver (matrix download + LAPACK; %d so far)" % (i, host_solves, _sr.device_solve_fallbacks()))
        if ckpt_path is not None and i % ckpt_every == 0:                              # :374-381
            xs = allgather(cm, np.asarray(x))                                          # (world, batch, n, dim) like the reference
            ks = allgather(cm, key.generate_state(2, dtype=np.uint32).astype(np.float64)).astype(np.uint32)
            if cm.rank == 0:
                os.makedirs(ckpt_path, exist_ok=True)
                save_data({"keys": ks, "x": xs, "params_van": params_van, "params_flow": params_flow, "opt_state": opt_state},
                          ckpt_filename(i, ckpt_path))
            # the run goes on from the key it has just written (two words per rank, the reference's key format, cannot hold a
            # SeedSequence's spawn history): a run resumed from this file then retraces this one draw for draw (main.py:217-223)
            key = np.random.SeedSequence([int(v) for v in np.atleast_2d(ks)[cm.rank].ravel()])
    return params_van, params_flow, rows

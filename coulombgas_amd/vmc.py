"""Host-side mirror of src/VMC.py (+ the jacrev / final-step algebra of main.py:277-298).

Arrays: every closure accepts numpy arrays or `DeviceArray`s (engine.py).  The arithmetic of an optimisation step -- local
energies, their moments, clip widths, loss weights, theta-gradients -- runs in kernels on device-resident arrays
(Engine.*_d); the host sees the ten observables and O(P) vectors only."""
import numpy as np
from .comm import get_comm
from .mcmc import mcmc

OBSERVABLES = ("K_mean", "K2_mean", "V_mean", "V2_mean", "E_mean", "E2_mean", "F_mean", "F2_mean", "S_mean", "S2_mean")
_I_E, _I_F = 4, 6          # positions of <E> and <F> in the moments vector (src/VMC.py:46-53 order)


def sample_stateindices_and_x(key, sampler, params_van, logp, x, params_flow, mc_steps, mc_stddev, L, comm=None):
    """src/VMC.py:8-25 for ONE device (one process per GPU replaces the pmap).
    key: numpy SeedSequence (or int).  sampler(params_van, key_state, batch) -> (batch, n) int state indices: a numpy array
    (GroundStateSampler) or a DeviceArray (the autoregressive Transformer samples on the GPU, cg_van_sample).
    x: numpy array, or a DeviceArray that is advanced in place (the reference donates x, src/VMC.py:11).
    Returns key, state_indices, x (wrapped into [0,L)), accept_rate (pmean'd)."""
    ss = key if isinstance(key, np.random.SeedSequence) else np.random.SeedSequence(int(key))
    key, key_state, key_MCMC = ss.spawn(3)                                    # jax.random.split(key, 3), :20
    batch = np.shape(x)[0]
    state_indices = sampler(params_van, key_state, batch)          # a device-side sampler hands back a DeviceArray
    if not hasattr(state_indices, "ptr"):
        state_indices = np.asarray(state_indices, dtype=np.int32)
    comm = comm or get_comm()
    x, accept_rate = mcmc(logp.bind(params_flow, state_indices), x, key_MCMC, mc_steps, mc_stddev,
                          walker_offset=comm.rank * batch, comm=comm, wrap_L=L)     # :23-24 (wrap fused into the call)
    return key, state_indices, x, accept_rate


def complex_clip(a, lo, hi):
    """jnp.clip on complex E_loc (src/VMC.py:73) = minimum(maximum(a, lo), hi) with the lexicographic
    complex order of the JAX generation the reference targets.  Host restatement of k_clip_weights,
    used by the value path of quantum_lossfn only."""
    a = np.asarray(a, dtype=np.complex128)
    lt = lambda p, q: (p.real < q.real) | ((p.real == q.real) & (p.imag < q.imag))
    lo_c, hi_c = np.complex128(lo), np.complex128(hi)
    m = np.where(lt(a, lo_c), lo_c, a)
    return np.where(lt(hi_c, m), hi_c, m)


def make_loss(log_prob, logpsi, logpsi_grad_laplacian, kappa, G, L, rs, Vconst, beta, comm=None, fuse_scores=True):
    """src/VMC.py:31-80.  `logpsi` is the vmapped closure returned by make_logpsi_grad_laplacian
    (main.py:254-259).  The two loss closures return the reference's (value, score) pair and carry
    `.grad()`, which returns what jax.jacrev(lossfn) returns in main.py:277-278.
    fuse_scores: the grad / Laplacian call also leaves the per-sample scores of the same walkers on the device (one fused kernel at the
    production sizes: the set-up of the two is the same), where quantum_lossfn.grad finds them; make_observable turns it off."""
    wf = logpsi.wf
    fuse = fuse_scores and getattr(logpsi_grad_laplacian, "takes_with_scores", False)

    def observable_and_lossfn(params_van, params_flow, state_indices, x, key):
        cm = comm or get_comm()
        eng = wf.engine(x, params_flow)
        eng.set_ewald(kappa, G, rs)
        x_d = eng.asdevice(x, "x")
        s_d = eng.asdevice(state_indices, "sidx", np.int32)
        lps = log_prob(params_van, state_indices)                                            # :34 (device Transformer: a DeviceArray)
        if hasattr(lps, "ptr"):
            lps_d = lps
        else:
            lps = np.asarray(lps, dtype=np.float64)
            lps_d = eng.asdevice(lps, "logp_states") if lps.any() else None
        grad, laplacian = (logpsi_grad_laplacian(x_d, params_flow, s_d, key, with_scores=True) if fuse
                           else logpsi_grad_laplacian(x_d, params_flow, s_d, key))           # :35, stays on the device
        V = eng.ewald_d(x_d)                                                                 # :40
        Eloc, Floc, mom = eng.local_energy_d(grad, laplacian, V, lps_d, Vconst, beta)       # :39-42 + local means of :46-53
        cm.pmean_d(mom)                                                                      # :44-53, one 10-double all-reduce
        vals = eng.to_host(mom)
        observable = {k: float(v) for k, v in zip(OBSERVABLES, vals)}
        F_mean, E_mean = observable["F_mean"], observable["E_mean"]
        B = int(np.shape(x)[0])
        cache = {}

        def tv_E():
            """pmean(mean |E_loc - <E>|) of :72, computed once per call of observable_and_lossfn, on the device"""
            if "tvE" not in cache:
                cache["tvE"] = cm.pmean_d(eng.abs_dev_d(Eloc, (mom, _I_E), "tv_E"))
            return cache["tvE"]

        def classical_lossfn(params_van, values=True):
            """src/VMC.py:60-67.  The clip width, the clipped F_loc and the weights d gradF_phi / d logp_states[b] = F_clip[b] / B are
            formed on the device (cg_abs_dev / cg_clip_weights in their real-valued mode, the pmean of :63 in place on the device
            scalar): F_loc never visits the host.  values=False (the driver: only .weights / .score_weights are used) skips the
            two O(B) read-backs the returned pair needs."""
            tvF = cm.pmean_d(eng.abs_dev_d(Floc, (mom, _I_F), "tv_F"))                        # :63
            w, _ = eng.clip_weights_d(Floc, (mom, _I_F), tvF, 1.0 / B, "wF")                  # clip(F_loc, <F> -+ 5 tv) / B
            classical_lossfn.weights = w
            classical_lossfn.score_weights = eng.asdevice(np.full(B, 1.0 / B), "w_uniform")
            if not values:
                return None, None
            lps = np.asarray(log_prob(params_van, state_indices), dtype=np.float64)
            return float(lps @ eng.to_host(w)), float(lps.mean())

        def quantum_lossfn(params_flow):
            """(gradF_theta, quantum_score) values of :69-76 (diagnostic path: downloads log Psi and E_loc)"""
            logpsix = np.asarray(logpsi(x_d, params_flow, s_d))
            tv = float(eng.to_host(tv_E())[0])
            Eloc_clipped = complex_clip(eng.to_host(Eloc), E_mean - 5.0 * tv, E_mean + 5.0 * tv)
            quantum_lossfn.Eloc_clipped = Eloc_clipped
            return float(2 * (logpsix * Eloc_clipped.conj()).real.mean()), float(2 * logpsix.real.mean())

        def quantum_grad(params_flow, as_pytree=True, reduce=False):
            """(d gradF_theta / d theta, d quantum_score / d theta) = jax.jacrev(quantum_lossfn)(params_flow), main.py:278.
            2 mean Re(logPsi conj(Ec)) -> weights (2/B) Re Ec on Re logPsi and (2/B) Im Ec on Im logPsi; the score gradient is
            2 Re mean_b S_b.  reduce=True also applies the pmean of main.py:280 (one 3P-double all-reduce on the device)."""
            e = wf.engine(x, params_flow)
            P = e.P
            w_re, w_im = e.clip_weights_d(Eloc, (mom, _I_E), tv_E(), 2.0 / B, "w")          # :73 + weights
            e.scores_compute_d(x_d, s_d)
            gs = e.scratch("grad_score", (3 * P,))
            e.scores_vjp_d(w_re, w_im, gs, 0)
            e.scores_mean_d(gs, P)
            if reduce:
                cm.pmean_d(gs)
            h = e.to_host(gs)
            g, s = h[:P].copy(), 2.0 * h[P::2]
            if not as_pytree:
                return g, s
            dim = np.shape(x)[-1]
            return wf.flow.unravel(g, dim), wf.flow.unravel(s, dim)

        quantum_lossfn.grad = quantum_grad
        observable_and_lossfn.Eloc, observable_and_lossfn.Floc = Eloc, Floc
        return observable, classical_lossfn, quantum_lossfn

    return observable_and_lossfn


def make_observable(log_prob, logpsi, logpsi_grad_laplacian, kappa, G, L, rs, Vconst, beta, comm=None):
    """Thin alias named by BASELINE.json's north star: returns only the observable dict of make_loss."""
    f = make_loss(log_prob, logpsi, logpsi_grad_laplacian, kappa, G, L, rs, Vconst, beta, comm, fuse_scores=False)
    return lambda *a: f(*a)[0]

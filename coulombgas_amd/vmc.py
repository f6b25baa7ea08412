"""Host-side mirror of src/VMC.py (+ the jacrev / final-step algebra of main.py:277-298)."""
import numpy as np
from .comm import get_comm
from .mcmc import mcmc
from .potential import potential_energy


def sample_stateindices_and_x(key, sampler, params_van, logp, x, params_flow, mc_steps, mc_stddev, L, comm=None):
    """src/VMC.py:8-25 for ONE device (one process per GPU replaces the pmap).
    key: numpy SeedSequence (or int).  sampler(params_van, key_state, batch) -> (batch, n) int state indices
    (the autoregressive sampler is outside the accelerated path and stays host code).
    Returns key, state_indices, x (wrapped into [0,L)), accept_rate (pmean'd)."""
    ss = key if isinstance(key, np.random.SeedSequence) else np.random.SeedSequence(int(key))
    key, key_state, key_MCMC = ss.spawn(3)                                    # jax.random.split(key, 3), :20
    batch = np.shape(x)[0]
    state_indices = np.asarray(sampler(params_van, key_state, batch), dtype=np.int32)
    comm = comm or get_comm()
    x, accept_rate = mcmc(logp.bind(params_flow, state_indices), x, key_MCMC, mc_steps, mc_stddev,
                          walker_offset=comm.rank * batch, comm=comm)
    x = x - L * np.floor(x / L)                                                # :24
    return key, state_indices, x, accept_rate


def complex_clip(a, lo, hi):
    """jnp.clip on complex E_loc (src/VMC.py:73) = minimum(maximum(a, lo), hi) with the lexicographic
    complex order of the JAX generation the reference targets."""
    a = np.asarray(a, dtype=np.complex128)
    lt = lambda p, q: (p.real < q.real) | ((p.real == q.real) & (p.imag < q.imag))
    lo_c, hi_c = np.complex128(lo), np.complex128(hi)
    m = np.where(lt(a, lo_c), lo_c, a)
    return np.where(lt(hi_c, m), hi_c, m)


def make_loss(log_prob, logpsi, logpsi_grad_laplacian, kappa, G, L, rs, Vconst, beta, comm=None):
    """src/VMC.py:31-80.  `logpsi` is the vmapped closure returned by make_logpsi_grad_laplacian
    (main.py:254-259).  The two loss closures return the reference's (value, score) pair and carry
    `.grad()`, which returns what jax.jacrev(lossfn) returns in main.py:277-278."""
    wf = logpsi.wf

    def observable_and_lossfn(params_van, params_flow, state_indices, x, key):
        cm = comm or get_comm()
        pmean = cm.pmean
        logp_states = np.asarray(log_prob(params_van, state_indices), dtype=np.float64)
        grad, laplacian = logpsi_grad_laplacian(x, params_flow, state_indices, key)
        kinetic = -laplacian - (grad ** 2).sum(axis=(-2, -1))                  # :39
        eng = wf.engine(x, params_flow)
        potential = potential_energy(x, kappa, G, L, rs, engine=eng) + Vconst  # :40
        Eloc = kinetic + potential
        Floc = logp_states / beta + Eloc.real

        vals = np.array([kinetic.real.mean(), (kinetic.real ** 2).mean(),
                         potential.mean(), (potential ** 2).mean(),
                         Eloc.real.mean(), (Eloc.real ** 2).mean(),
                         Floc.mean(), (Floc ** 2).mean(),
                         -logp_states.mean(), (logp_states ** 2).mean()])
        vals = pmean(vals)                                                      # :44-53 (one packed all-reduce)
        names = ["K_mean", "K2_mean", "V_mean", "V2_mean", "E_mean", "E2_mean", "F_mean", "F2_mean", "S_mean", "S2_mean"]
        observable = {k: float(v) for k, v in zip(names, vals)}
        F_mean, E_mean = observable["F_mean"], observable["E_mean"]
        B = Eloc.shape[0]

        def classical_lossfn(params_van):
            lps = np.asarray(log_prob(params_van, state_indices), dtype=np.float64)
            tv = pmean(float(np.abs(Floc - F_mean).mean()))                    # :63
            Floc_clipped = np.clip(Floc, F_mean - 5.0 * tv, F_mean + 5.0 * tv)
            classical_lossfn.weights = Floc_clipped / B      # d gradF_phi / d logp_states[b]
            classical_lossfn.score_weights = np.full(B, 1.0 / B)
            return float((lps * Floc_clipped).mean()), float(lps.mean())

        def quantum_lossfn(params_flow):
            logpsix = logpsi(x, params_flow, state_indices)
            tv = pmean(float(np.abs(Eloc - E_mean).mean()))                    # :72
            Eloc_clipped = complex_clip(Eloc, E_mean - 5.0 * tv, E_mean + 5.0 * tv)
            quantum_lossfn.Eloc_clipped = Eloc_clipped
            return float(2 * (logpsix * Eloc_clipped.conj()).real.mean()), float(2 * logpsix.real.mean())

        def quantum_grad(params_flow, as_pytree=True):
            """(d gradF_theta / d theta, d quantum_score / d theta) = jax.jacrev(quantum_lossfn)(params_flow), main.py:278.
            2 mean Re(logPsi conj(Ec)) -> weights (2/B) Re Ec on Re logPsi and (2/B) Im Ec on Im logPsi."""
            tv = pmean(float(np.abs(Eloc - E_mean).mean()))
            Ec = complex_clip(Eloc, E_mean - 5.0 * tv, E_mean + 5.0 * tv)
            e = wf.engine(x, params_flow)
            g = e.param_vjp(x, state_indices, 2.0 * Ec.real / B, 2.0 * Ec.imag / B)
            s = e.param_vjp(x, state_indices, np.full(B, 2.0 / B), np.zeros(B))
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
    f = make_loss(log_prob, logpsi, logpsi_grad_laplacian, kappa, G, L, rs, Vconst, beta, comm)
    return lambda *a: f(*a)[0]

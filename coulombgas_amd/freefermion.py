"""Stage-1 training of the reference (src/freefermion/pretraining.py): the autoregressive density matrix alone, on
non-interacting fermions, by minimising F = <log p / beta + E> with the classical natural gradient (fisher_sr) or adam.

`exact_free_energy` replaces the reference's mpmath evaluation (src/freefermion/analytic.py:38-80, 1200-digit
alternating recursion) by the positive-term canonical recursion over orbitals, which is stable in double precision:
    Z_m(k) = Z_{m-1}(k) + e^{-beta E_m} Z_{m-1}(k-1),   E = -d ln Z / d beta  (carried along),  F = -ln Z / beta, S = beta (E - F)."""
import numpy as np
from . import sr as _sr
from .autoregressive import make_autoregressive_sampler, make_classical_score


def exact_free_energy(Es, n, beta):
    """Canonical ensemble of n fermions in the single-particle levels Es.  Returns F, E, S."""
    Es = np.asarray(Es, dtype=np.float64)
    e0 = Es.min()
    w = np.exp(-beta * (Es - e0))                       # shifted weights in (0, 1]
    Z = np.zeros(n + 1); dZ = np.zeros(n + 1)           # Z(k), dZ(k) = dZ/dbeta of the shifted problem
    Z[0] = 1.0
    logscale = 0.0
    for m in range(Es.size):
        for k in range(min(m + 1, n), 0, -1):
            dZ[k] += w[m] * (dZ[k - 1] - (Es[m] - e0) * Z[k - 1])
            Z[k] += w[m] * Z[k - 1]
        s = Z.max()
        Z /= s; dZ /= s; logscale += np.log(s)
    lnZ = np.log(Z[n]) + logscale - beta * n * e0
    E = -dZ[n] / Z[n] + n * e0
    F = -lnZ / beta
    return F, E, beta * (E - F)


def make_loss(log_prob, Es, beta):
    """src/freefermion/pretraining.py:9-32.  loss_fn(params, state_indices) -> (gradF, aux); loss_fn.grad(...) is
    jax.grad of its first output: mean_b (F_b - <F>) d log p_b / d params (F under stop_gradient)."""
    Es = np.asarray(Es, dtype=np.float64)

    def stats(params, state_indices):
        logp = np.asarray(log_prob(params, state_indices))          # (device Transformer: one O(batch) read-back per epoch)
        E = Es[np.asarray(state_indices)].sum(axis=-1)
        F = logp / beta + E
        aux = {"E_mean": E.mean(), "E_std": E.std(), "F_mean": F.mean(), "F_std": F.std(),
               "S_mean": -logp.mean(), "S_std": (-logp).std()}
        return logp, F, aux

    def loss_fn(params, state_indices):
        logp, F, aux = stats(params, state_indices)
        return float((logp * (F - F.mean())).mean()), aux

    def grad(params, state_indices):
        logp, F, aux = stats(params, state_indices)
        return log_prob.vjp(params, state_indices, (F - F.mean()) / F.shape[0]), aux

    loss_fn.grad = grad
    return loss_fn


def pretrain(van, params_van, n, dim, Theta, sp_indices_twist, key, lr=1e-3, sr=True, damping=1e-3, max_norm=1e-3,
             batch=8192, epoch=5000, log=None, engine=None, density_matrix=None):
    """src/freefermion/pretraining.py:34-108.  sp_indices_twist: the reversed twisted orbital table (main.py:79-90).
    engine: a GPU Engine for (n, dim) -- created here when None: the sampler, the log-probabilities, the reverse pass (per-sample
    scores, weighted VJP) and the classical Fisher matrix + damped solve run on the device.
    density_matrix: a (sampler, log_prob) pair to use instead (the CPU tests of this loop pass their numpy checker).
    Returns the trained params_van and the data.txt rows (epoch, F, F_std, E, E_std, S, S_std)."""
    if dim == 3:
        L = (4 / 3 * np.pi * n) ** (1 / 3); beta = 1 / ((4.5 * np.pi) ** (2 / 3) * Theta)
    else:
        L = np.sqrt(np.pi * n); beta = 1 / (4 * Theta)
    sp = np.asarray(sp_indices_twist, dtype=np.float64)
    Es = (2 * np.pi / L) ** 2 * (sp ** 2).sum(axis=-1)
    if density_matrix is not None:
        sampler, log_prob = density_matrix
    else:
        if engine is None:                                  # the density matrix needs no flow: any architecture, unit box
            from .engine import Engine
            engine = Engine(n, dim, 2, 16, 16, 1.0, sp)
        sampler, log_prob = make_autoregressive_sampler(van, sp, n, sp.shape[0], engine=engine)
    loss_fn = make_loss(log_prob, Es, beta)
    if sr:
        optimizer = _sr.fisher_sr(make_classical_score(log_prob), damping, max_norm, engine)
    else:
        from .driver import adam
        optimizer = adam(lr)
    opt_state = optimizer.init(params_van)
    ss = key if isinstance(key, np.random.SeedSequence) else np.random.SeedSequence(int(key))
    rows = []
    for i in range(1, epoch + 1):
        ss, sub = ss.spawn(2)
        state_indices = sampler(params_van, np.random.default_rng(sub), batch)
        grads, aux = loss_fn.grad(params_van, state_indices)
        updates, opt_state = optimizer.update(grads, opt_state, params=(params_van, state_indices) if sr else None)
        params_van = _sr.apply_updates(params_van, updates)
        row = ("%6d" + "  %.6f" * 6) % (i, aux["F_mean"], aux["F_std"] / np.sqrt(batch), aux["E_mean"], aux["E_std"] / np.sqrt(batch),
                                        aux["S_mean"], aux["S_std"] / np.sqrt(batch))
        rows.append(row)
        if log is not None:
            log(row)
    return params_van, rows

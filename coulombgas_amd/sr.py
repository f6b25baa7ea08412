"""Host-side mirror of src/sr.py (stochastic reconfiguration), API in parallel with the reference's optax-style
`GradientTransformation(init, update)`.

The Fisher matrices are where the work is: the quantum one, Re(S^H S)/B over the per-sample scores
S = d log Psi / d theta (src/logpsi.py:183-203), is formed on the GPU (cg_quantum_fisher: reverse passes for the
scores, f64 MFMA SYRK, all on the device), and the centred, damped solve runs there too (cg_spd_solve: blocked Cholesky +
triangular solves); the norm clip (src/sr.py:102-117) is O(P) host arithmetic.  The classical score function (the autoregressive
Transformer; with coulombgas_amd.make_autoregressive_sampler its per-sample scores are formed and kept on the GPU as a
DeviceScores handle, cg_van_scores_*) is supplied by the caller and returns that handle, a (B, P_van) array, or a pytree of
arrays with a leading batch axis, ravelled in sorted-key order like jax's ravel_pytree."""
from collections import namedtuple
import numpy as np
from .comm import get_comm

GradientTransformation = namedtuple("GradientTransformation", ["init", "update"])
EmptyState = namedtuple("EmptyState", [])


def ravel_pytree(tree):
    """jax.flatten_util.ravel_pytree for nested dicts of arrays (leaves in sorted-key order).  Returns flat, unravel."""
    if isinstance(tree, np.ndarray) or np.isscalar(tree):
        a = np.asarray(tree)
        return a.reshape(-1), (lambda f, shp=a.shape: np.asarray(f).reshape(shp))
    keys = sorted(tree)
    parts = [ravel_pytree(tree[k]) for k in keys]
    sizes = [p[0].size for p in parts]
    flat = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0)

    def unravel(f):
        out, off = {}, 0
        for k, (_, un), sz in zip(keys, parts, sizes):
            out[k] = un(f[off:off + sz]); off += sz
        return out
    return flat, unravel


def _ravel_batched(score):
    """jax.vmap(lambda pytree: ravel_pytree(pytree)[0])(score): (B, P)."""
    if isinstance(score, np.ndarray):
        return score.reshape(score.shape[0], -1)
    keys = sorted(score)
    return np.concatenate([_ravel_batched(score[k]) for k in keys], axis=1)


_warned_indefinite = [False]
_host_solves = [0]          # damped solves that left the device ("not positive definite" there): see device_solve_fallbacks()


def device_solve_fallbacks():
    """How many damped solves of this process were rerouted from the GPU to the host's symmetric-indefinite solver because the
    shifted Fisher matrix was not positive definite on the device.  Each one downloads the matrix (279 MB at P = 5907) and costs
    seconds of LAPACK inside a training epoch; `train` reports the count of an epoch next to its data.txt row."""
    return _host_solves[0]


def _solve_and_clip(fisher, grads_raveled, damping, max_norm, engine=None, center=None):
    """src/sr.py:38-45 / 102-117: (F + damping I)^-1 g, scaled by -min(sqrt(max_norm / g.F^-1 g), 1).  With an engine the
    damped solve runs on the GPU at every size (cg_spd_solve: centring, shift, blocked Cholesky, both triangular solves); host
    LAPACK only for host matrices handed in WITHOUT an engine (a score function of the caller's own).  The ONLY device outcome that is handled here is CG_ERR_STATE = "the shifted matrix is not
    positive definite" (round-off on a nearly singular Fisher matrix): the reference's jax.scipy.linalg.solve is a general
    LU solve and would still return a result, so that case goes to the host's symmetric-indefinite solver, once with a
    warning.  Every other device error propagates.  center: complex score mean m; the matrix solved is
    F - Re(conj(m) m^T) (src/sr.py:88)."""
    from scipy.linalg import solve, LinAlgError
    from ._lib import CoulombGasError, CG_ERR_STATE
    upd = None
    on_device = not isinstance(fisher, np.ndarray) and hasattr(fisher, "ptr")
    if on_device or (engine is not None and hasattr(engine, "spd_solve")):
        try:
            if on_device:          # the matrix never leaves HBM (the solver factors a device-side copy)
                upd = fisher.eng.spd_solve_d(fisher.base, grads_raveled, damping, center, index=fisher.index, P=fisher.shape[0])
            else:
                upd = engine.spd_solve(fisher, grads_raveled, damping, center)
        except CoulombGasError as e:
            if e.code != CG_ERR_STATE:
                raise
            _host_solves[0] += 1
            if not _warned_indefinite[0]:
                import warnings
                warnings.warn("SR: damped Fisher matrix not positive definite on the device (%s); symmetric-indefinite host solve" % e)
                _warned_indefinite[0] = True
            if on_device:
                fisher = np.asarray(fisher)
    if upd is None:
        if center is not None:                     # src/sr.py:88
            fisher = fisher - np.outer(center.real, center.real) - np.outer(center.imag, center.imag)
        fisher = fisher + damping * np.eye(fisher.shape[0])
        try:                                       # Fisher + damping I is symmetric positive definite: Cholesky
            upd = solve(fisher, grads_raveled, assume_a="pos")
        except LinAlgError:                        # (round-off made it indefinite: the symmetric solver)
            upd = solve(fisher, grads_raveled, assume_a="sym")
    gnorm = float(np.sum(grads_raveled * upd))
    scale = min(np.sqrt(max_norm / gnorm), 1.0) if gnorm > 0 else 1.0
    return -scale * upd


def fisher_sr(score_fn, damping, max_norm, engine=None):
    """src/sr.py:13-52: natural gradient for a purely classical model.  update(grads, state, (params, state_indices)).
    engine (optional): a coulombgas_amd Engine whose GPU forms the Fisher matrix."""
    def init_fn(params):
        return EmptyState()

    def update_fn(grads, state, params):
        params, state_indices = params
        g, unravel = ravel_pytree(grads)
        score = score_fn(params, state_indices)
        if hasattr(score, "fisher_d"):                        # device Transformer: scores and Fisher matrix stay on the GPU
            fisher = score.fisher_d()
        else:
            score = _ravel_batched(score)
            fisher = engine.fisher_real(score) if engine is not None and hasattr(engine, "fisher_real") else score.T.dot(score) / score.shape[0]
        return unravel(_solve_and_clip(fisher, g, damping, max_norm, engine)), state

    return GradientTransformation(init_fn, update_fn)


def hybrid_fisher_sr(classical_score_fn, quantum_score_fn, damping, max_norm, comm=None):
    """src/sr.py:56-122.  Returns (fishers_fn, optimizer) like the reference.
    fishers_fn(params_van, params_flow, state_indices, x) -> (classical_fisher, quantum_fisher, quantum_score_mean), each
    averaged over the ranks (the reference's pmean, :70-76).  `quantum_score_fn` is make_quantum_score(logpsi): its wave
    function's engine computes Re(S^H S)/B and mean(S) on the GPU without moving the (B, P) score matrix to the host."""
    wf = quantum_score_fn.wf
    last_engine = [None]

    def fishers_fn(params_van, params_flow, state_indices, x, reduce=True):
        """src/sr.py:65-84.  The matrices are formed, all-reduced and returned in HBM (DeviceArrays on a GPU engine): the
        (P x P) quantum Fisher matrix and the mean score share one buffer and one all-reduce; nothing of size O(B P) or
        O(P^2) visits the host.
        reduce=False: this rank's matrices, NOT averaged over the ranks -- make_update accumulates them over the acc_steps of an
        update and averages the sums once (fishers_fn.reduce below): the mean over the ranks is linear, so acc_steps all-reduces of
        P_van^2 + P^2 doubles (279 MB + 9 MB at the shipped sizes) become one."""
        cm = comm or get_comm()
        eng = wf.engine(x, params_flow)
        last_engine[0] = eng
        red = (lambda a: cm.pmean_d(a)) if reduce else (lambda a: a)
        classical_fisher = None
        if classical_score_fn is not None:
            cs = classical_score_fn(params_van, state_indices)
            if hasattr(cs, "fisher_d"):                       # device Transformer: the scores never leave the GPU
                classical_fisher = red(cs.fisher_d())
            else:
                classical_fisher = red(eng.fisher_real_d(_ravel_batched(cs)))            # :77-79
        x_d = eng.asdevice(x, "x")
        s_d = eng.asdevice(state_indices, "sidx", np.int32)
        eng.scores_compute_d(x_d, s_d)                                                   # :69-71 (shared with the theta-VJP)
        P = eng.P
        pack = eng.scratch("fisher_pack", (P * P + 2 * P,))
        eng.scores_fisher_d(pack, 0, P * P)
        red(pack)                                                                        # :73, :80-82 in one all-reduce
        qf = eng.view(pack, 0, (P, P))
        sm = eng.to_host_slice(pack, P * P, 2 * P)
        return classical_fisher, qf, sm[0::2] + 1j * sm[1::2]

    def reduce_accumulated(fishers):
        """the pmean of src/sr.py:70-76 applied to the SUMS over the accumulation steps (what fishers_fn(reduce=False) returned, added
        up): device matrices in place, the complex score mean through the host"""
        cm = comm or get_comm()
        cf, qf, qm = fishers
        if cf is not None:
            cf = cm.pmean_d(cf) if hasattr(cf, "ptr") else cm.pmean(cf)
        qf = cm.pmean_d(qf) if hasattr(qf, "ptr") else cm.pmean(qf)
        qm = np.asarray(qm)
        both = np.asarray(cm.pmean(np.concatenate([qm.real, qm.imag])))
        return cf, qf, both[:qm.size] + 1j * both[qm.size:]
    fishers_fn.reduce_accumulated = reduce_accumulated

    def init_fn(params):
        return EmptyState()

    def update_fn(grads, state, params):
        grad_params_van, grad_params_flow = grads
        classical_fisher, quantum_fisher, quantum_score_mean = params
        update_van = None
        if grad_params_van is not None and classical_fisher is not None:
            gv, unravel_van = ravel_pytree(grad_params_van)
            update_van = unravel_van(_solve_and_clip(classical_fisher, gv, damping, max_norm, last_engine[0]))
        gf, unravel_flow = ravel_pytree(grad_params_flow)
        # the centring of :88, F - Re(conj(m) m^T), is applied inside the solve (on the device when there is one)
        update_flow = unravel_flow(_solve_and_clip(quantum_fisher, gf, damping, max_norm, last_engine[0],
                                                   center=np.asarray(quantum_score_mean)))
        return (update_van, update_flow), state

    return fishers_fn, GradientTransformation(init_fn, update_fn)


def apply_updates(params, updates):
    """optax.apply_updates for nested dicts of arrays (None entries pass through)."""
    if updates is None:
        return params
    if isinstance(params, dict):
        return {k: apply_updates(params[k], updates[k]) for k in params}
    if isinstance(params, (tuple, list)):
        return type(params)(apply_updates(p, u) for p, u in zip(params, updates))
    return np.asarray(params) + np.asarray(updates)

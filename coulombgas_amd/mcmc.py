"""Host-side mirror of src/MCMC.py.  The whole chain (mc_steps proposals, logp evaluations,
accept/select) is one kernel launch (cg_mcmc)."""
import numpy as np
from .comm import get_comm


def _seed_of(key):
    if isinstance(key, np.random.SeedSequence):
        return int(key.generate_state(1, dtype=np.uint64)[0])
    if isinstance(key, np.random.Generator):
        return int(key.integers(0, 2 ** 63))
    return int(key)


def mcmc(logp_fn, x_init, key, mc_steps, mc_stddev=0.02, noise=None, unif=None, walker_offset=0, comm=None):
    """src/MCMC.py:7-40.  `logp_fn` must be `logp.bind(params_flow, state_indices)` (make_logp): the
    reference passes `lambda x: logp(x, params_flow, state_indices)` (src/VMC.py:23); an opaque Python
    callable cannot run inside the GPU chain, and there is no CPU fallback.
    `key`: int / SeedSequence / Generator seeding the in-kernel Philox stream; or pass `noise`
    (steps,B,n,dim) and `unif` (steps,B) to replace jax.random.normal/uniform (src/MCMC.py:26,29).
    Returns x, accept_rate (mean over ranks, src/MCMC.py:39)."""
    if not hasattr(logp_fn, "wf"):
        raise TypeError("mcmc needs logp.bind(params_flow, state_indices) from make_logp, got %r" % (logp_fn,))
    eng = logp_fn.wf.engine(x_init, logp_fn.params)
    x, _, nacc = eng.mcmc(x_init, logp_fn.state_indices, mc_steps, mc_stddev, seed=_seed_of(key),
                          walker_offset=walker_offset, noise=noise, unif=unif)
    batch = int(np.prod(np.shape(x_init)[:-2]))
    accept_rate = nacc / (mc_steps * batch) if mc_steps * batch else 0.0
    accept_rate = (comm or get_comm()).pmean(accept_rate)
    return x, accept_rate

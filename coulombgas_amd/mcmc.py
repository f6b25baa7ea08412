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


def mcmc(logp_fn, x_init, key, mc_steps, mc_stddev=0.02, noise=None, unif=None, walker_offset=0, comm=None, wrap_L=None):
    """src/MCMC.py:7-40.  `logp_fn` must be `logp.bind(params_flow, state_indices)` (make_logp): the
    reference passes `lambda x: logp(x, params_flow, state_indices)` (src/VMC.py:23); an opaque Python
    callable cannot run inside the GPU chain, and there is no CPU fallback.
    `key`: int / SeedSequence / Generator seeding the in-kernel Philox stream; or pass `noise`
    (steps,B,n,dim) and `unif` (steps,B) to replace jax.random.normal/uniform (src/MCMC.py:26,29).
    x_init: numpy array (a new array is returned) or DeviceArray (advanced in place and returned).
    wrap_L: also apply x -= L floor(x / L) (src/VMC.py:24) before anything leaves the device.
    Returns x, accept_rate (mean over ranks, src/MCMC.py:39)."""
    if not hasattr(logp_fn, "wf"):
        raise TypeError("mcmc needs logp.bind(params_flow, state_indices) from make_logp, got %r" % (logp_fn,))
    eng = logp_fn.wf.engine(x_init, logp_fn.params)
    lead = np.shape(x_init)[:-2]
    batch = int(np.prod(lead))
    on_device = not isinstance(x_init, np.ndarray) and hasattr(x_init, "ptr")
    if on_device:
        x_d = x_init
    else:
        x_d = eng.asdevice(np.asarray(x_init, dtype=np.float64).reshape((batch,) + tuple(np.shape(x_init)[-2:])), "x_chain")
    si = logp_fn.state_indices
    s_d = si if hasattr(si, "ptr") else eng.asdevice(np.asarray(si).reshape(batch, -1), "sidx", np.int32)
    cm = comm or get_comm()
    dev_rate = hasattr(cm, "accept_rate") and hasattr(eng, "mcmc_accept_rate")
    nacc = eng.mcmc_d(x_d, s_d, mc_steps, mc_stddev, seed=_seed_of(key), walker_offset=walker_offset, noise=noise, unif=unif,
                      **({"count": False} if dev_rate else {}))
    if wrap_L is not None:
        eng.wrap_d(x_d)
    if dev_rate:      # src/MCMC.py:37-39: the rate is formed from the device counter and averaged over the ranks there
        accept_rate = cm.accept_rate(eng, mc_steps * batch)
    else:
        accept_rate = cm.pmean(nacc / (mc_steps * batch) if mc_steps * batch else 0.0)
    x = x_d if on_device else eng.to_host(x_d).reshape(lead + tuple(np.shape(x_init)[-2:]))
    return x, accept_rate

"""The autoregressive Transformer density matrix of src/autoregressive.py / src/sampler.py on the GPU: this module is the
host-side mirror of the reference's interface (Transformer, make_autoregressive_sampler, make_classical_score) on top of the
device kernels of csrc/cg_van.hpp (cg_van_sample / cg_van_log_prob / cg_van_scores_*: sampler, log-probability, per-sample
scores, weighted VJP, classical Fisher matrix).  Parameters keep Haiku's names and shapes, so shipped `params_van`
(checkpoints, pretrained models) load unchanged.  There is no host fallback: the closures raise without a GPU engine (a numpy
restatement of the same functions lives under tests/ as a checker)."""
import numpy as np


class Transformer:
    """src/autoregressive.py:50-96: architecture, parameter tree (Haiku names / shapes) and initialiser.  The forward and
    reverse passes run on the device (csrc/cg_van.hpp)."""

    def __init__(self, output_size, num_layers, model_size, num_heads, hidden_size, name="transformer"):
        if model_size % num_heads != 0:
            raise ValueError("Model_size of the transformer must be divisible by the number of heads. "
                             "Got model_size=%d and num_heads=%d." % (model_size, num_heads))
        self.output_size, self.num_layers, self.model_size, self.num_heads = output_size, num_layers, model_size, num_heads
        self.key_size = model_size // num_heads
        self.hidden_size, self.name = hidden_size, name

    # -- parameters ------------------------------------------------------------------------------------------
    def param_shapes(self, dim):
        nm, ms, hs = self.name, self.model_size, self.hidden_size
        shapes = {nm: {"x1hat": (self.output_size,)}, nm + "/embedding_mlp": {"b": (ms,), "w": (dim, ms)},
                  nm + "/output_mlp": {"b": (self.output_size,), "w": (ms, self.output_size)}}
        for i in range(self.num_layers):
            for part in ("query", "key", "value", "linear"):
                shapes["%s/layer%d_attn/%s" % (nm, i, part)] = {"b": (ms,), "w": (ms, ms)}
            shapes["%s/layer%d_mlp/linear" % (nm, i)] = {"b": (hs,), "w": (ms, hs)}
            shapes["%s/layer%d_mlp/linear_1" % (nm, i)] = {"b": (ms,), "w": (hs, ms)}
        return shapes

    def init(self, key, x):
        """hk.transform(...).init: VarianceScaling(0.02 / num_layers) weights (fan_in, truncated normal), zero biases,
        x1hat ~ TruncatedNormal(sqrt(init_scale / output_size)) (src/autoregressive.py:72-93)."""
        rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
        dim = np.shape(x)[-1]
        scale = 0.02 / self.num_layers

        def trunc(shape, std):
            v = rng.standard_normal(shape)
            bad = np.abs(v) > 2.0
            while bad.any():
                v[bad] = rng.standard_normal(int(bad.sum())); bad = np.abs(v) > 2.0
            return v * std
        out = {}
        for mod, leaves in self.param_shapes(dim).items():
            out[mod] = {}
            for leaf, shp in leaves.items():
                if leaf == "b":
                    out[mod][leaf] = np.zeros(shp)
                elif leaf == "x1hat":
                    out[mod][leaf] = trunc(shp, np.sqrt(scale / self.output_size))
                else:
                    fan = shp[1] if mod.endswith("embedding_mlp") else shp[0]      # "fan_out" for the embedding (:75)
                    out[mod][leaf] = trunc(shp, np.sqrt(scale / fan) / 0.87962566103423978)
        return out


def flat_params(network, params, dim):
    """Transformer parameters in the flat order of include/coulombgas.h (cg_van_set_params)."""
    nm = network.name
    parts = [params[nm]["x1hat"]]
    for m in _flat_modules(network):
        parts += [params[m]["b"], params[m]["w"]]
    return np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in parts])


def _flat_modules(network):
    nm = network.name
    mods = [nm + "/embedding_mlp"]
    for i in range(network.num_layers):
        mods += ["%s/layer%d_attn/%s" % (nm, i, part) for part in ("query", "key", "value", "linear")]
        mods += ["%s/layer%d_mlp/linear" % (nm, i), "%s/layer%d_mlp/linear_1" % (nm, i)]
    return mods + [nm + "/output_mlp"]


def unflat_params(network, flat, dim):
    """inverse of flat_params; flat may carry leading axes (per-sample gradients)"""
    flat = np.asarray(flat)
    lead, shapes, off = flat.shape[:-1], network.param_shapes(dim), 0
    out = {}

    def take(shp):
        nonlocal off
        sz = int(np.prod(shp)); a = flat[..., off:off + sz].reshape(lead + tuple(shp)); off += sz
        return a
    out[network.name] = {"x1hat": take(shapes[network.name]["x1hat"])}
    for m in _flat_modules(network):
        b = take(shapes[m]["b"]); out[m] = {"b": b, "w": take(shapes[m]["w"])}
    assert off == flat.shape[-1]
    return out


class DeviceScores:
    """Per-sample gradients of log p held on the GPU (cg_van_scores_compute).  hybrid_fisher_sr takes the Fisher matrix from
    .fisher_d() without the (B, P) matrix visiting the host; np.asarray(.) / .tree() download it (tests, diagnostics).  All
    host-visible results are in ravel_pytree order (sorted keys, like jax.flatten_util) -- perm maps it to the device's."""

    _perms = {}          # (network architecture, dim) -> (parameter count, ravel order -> flat order): formed once, not per call

    @classmethod
    def layout(cls, network, dim):
        key = (network.name, network.output_size, network.num_layers, network.model_size, network.num_heads, network.hidden_size, dim)
        got = cls._perms.get(key)
        if got is None:
            from .sr import ravel_pytree
            n = sum(int(np.prod(shp)) for leaves in network.param_shapes(dim).values() for shp in leaves.values())
            got = cls._perms[key] = (n, ravel_pytree(unflat_params(network, np.arange(n), dim))[0].astype(np.int32))
        return got

    def __init__(self, engine, network, dim, B):
        self.engine, self.network, self.dim, self.B = engine, network, dim, B
        n, self.perm = self.layout(network, dim)
        self.shape = (B, n)

    def fisher_d(self):
        return self.engine.van_scores_fisher_d(self.perm)

    def flat(self):
        return self.engine.van_scores_get()                     # (B, P) in the device's flat order

    def tree(self):
        return unflat_params(self.network, self.flat(), self.dim)

    def __array__(self, dtype=None, copy=None):
        a = self.flat()[:, self.perm]
        return a if dtype is None else a.astype(dtype)


def make_autoregressive_sampler(network, sp_indices, n, num_states, mask_fn=False, engine=None):
    """src/sampler.py:4-50 with a leading batch axis built in (the reference vmaps).  sampler(params, key, batch) ->
    (batch, n) int32 sorted state indices; log_prob(params, state_indices (batch, n)) -> (batch,).
    engine (or sampler.attach(engine) later; coulombgas_amd.train does it): a GPU Engine of the same (n, dim) -- the sampler
    and the log-probability run on the device (cg_van_sample / cg_van_log_prob: one wave per sample, key / value cache
    in LDS) and hand DeviceArrays to the hot path, and the gradients (log_prob.grad -> DeviceScores, log_prob.vjp) come from
    the device's reverse pass (cg_van_scores_*).  Calling the closures with no engine attached raises."""
    sp_indices = np.asarray(sp_indices, dtype=np.float64)
    base = np.tril(np.ones((n, num_states), dtype=bool), k=num_states - n)
    dev = {"engine": engine}

    bound = {"leaves": None, "eng": None}          # the parameter arrays the engine holds (by identity: a step calls in five times with one pytree)

    def _dev_engine(params):
        eng = dev["engine"]
        if eng is None:
            raise RuntimeError("autoregressive density matrix: no GPU engine attached (pass engine= / call .attach(engine); "
                               "coulombgas_amd.train attaches its own)")
        leaves = [params[network.name]["x1hat"]] + [params[m][l] for m in _flat_modules(network) for l in ("b", "w")]
        old = bound["leaves"]
        if bound["eng"] is eng and old is not None and len(old) == len(leaves) and all(a is b_ for a, b_ in zip(old, leaves)) \
                and getattr(eng, "_van_key", None) is not None and eng._van_key[3] is bound:
            return eng                                   # the very arrays of the last call (nobody else has set parameters since)
        eng.van_set_params((num_states, network.num_layers, network.model_size, network.num_heads, network.hidden_size),
                           sp_indices, flat_params(network, params, sp_indices.shape[1]), owner=bound)
        bound["leaves"], bound["eng"] = leaves, eng
        return eng

    def _mask(state_idx):
        """src/sampler.py:72-91 (host logic: which orbitals each electron may still take)"""
        state_idx = np.asarray(state_idx)
        idx_lb = np.concatenate([np.full(state_idx.shape[:-1] + (1,), -1), state_idx[..., :-1]], axis=-1)
        return base & (np.arange(num_states) > idx_lb[..., None])

    def sampler(params, key, batch, unif=None):
        eng = _dev_engine(params)
        from .mcmc import _seed_of
        return eng.van_sample_d(batch, 0 if unif is not None else _seed_of(key), 0, unif)[0]

    def log_prob(params, state_idx):
        eng = _dev_engine(params)
        if hasattr(state_idx, "ptr"):
            return eng.van_log_prob_d(state_idx)
        return eng.van_log_prob(state_idx)

    def _dev_scores(params, state_idx):
        eng = _dev_engine(params)
        s_d = state_idx if hasattr(state_idx, "ptr") else eng.asdevice(np.asarray(state_idx, dtype=np.int32), "van_sidx_in", np.int32)
        eng.van_scores_compute_d(s_d)
        return eng

    def grad(params, state_idx):
        """jax.vmap(jax.grad(log_prob), (None, 0), 0): per-sample gradients as a DeviceScores handle (np.asarray / .tree()
        download them)."""
        eng = _dev_scores(params, state_idx)
        return DeviceScores(eng, network, sp_indices.shape[1], int(np.shape(state_idx)[0]))

    def vjp(params, state_idx, w):
        """sum_b w[b] * d log_prob_b / d params  (what jax.jacrev of a weighted sum of log-probabilities returns)."""
        eng = _dev_scores(params, state_idx)
        w_d = w if hasattr(w, "ptr") else eng.asdevice(np.asarray(w, dtype=np.float64), "van_w")
        g = eng.van_scores_vjp_d(w_d, eng.scratch("van_vjp", (DeviceScores.layout(network, sp_indices.shape[1])[0],)))
        return unflat_params(network, eng.to_host(g), sp_indices.shape[1])

    def vjp_pair_d(params, state_idx, w1, w2):
        """both weighted sums jax.jacrev(classical_lossfn) needs (main.py:277) in ONE device buffer [sum_b w1[b] S_b | sum_b w2[b] S_b]
        (flat parameter order): the caller all-reduces it in place (main.py:280) before anything is read back."""
        eng = _dev_scores(params, state_idx)
        Pv = DeviceScores.layout(network, sp_indices.shape[1])[0]
        out = eng.scratch("van_vjp_pair", (2 * Pv,))
        for k, w in enumerate((w1, w2)):
            w_d = w if hasattr(w, "ptr") else eng.asdevice(np.asarray(w, dtype=np.float64), "van_w%d" % k)
            eng.van_scores_vjp_d(w_d, eng.view(out, k * Pv, (Pv,)))
        return out, (lambda flat: unflat_params(network, flat, sp_indices.shape[1]))

    log_prob.grad, log_prob.vjp, log_prob.vjp_pair_d = grad, vjp, vjp_pair_d

    def attach(engine):
        dev["engine"] = engine
    sampler.attach = log_prob.attach = attach
    if mask_fn:
        return _mask, sampler, log_prob
    return sampler, log_prob


def make_classical_score(log_prob):
    """src/sampler.py:52-65: params, (batch, n) samples -> pytree of per-sample scores d log p / d params."""
    return lambda params, state_indices: log_prob.grad(params, state_indices)

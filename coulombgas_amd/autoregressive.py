"""Host-side (numpy) forward pass of the autoregressive Transformer of src/autoregressive.py and the sampler /
log-probability of src/sampler.py.  It produces the integer `state_indices` the accelerated hot path consumes and the
entropy term log p of the loss; it is n tiny batched matmuls per sampling call and stays on the host.

Gradients of log p w.r.t. the parameters (jax.grad(log_prob), src/sampler.py:65: the classical score of the SR optimizer
and the VJP jax.jacrev(classical_lossfn) needs) come from a hand-written reverse pass of the same forward code
(`log_prob.grad` / `log_prob.vjp` / `make_classical_score`).  Parameters keep Haiku's names and shapes, so shipped
`params_van` (checkpoints, pretrained models) load unchanged."""
import contextlib
import numpy as np


def _blas_limit():
    """The model is tiny (16-wide): thousands of small batched matmuls.  On many-core hosts an unrestricted BLAS thread pool
    turns each of them into a synchronisation storm (measured: minutes instead of milliseconds on a 256-thread box)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=4, user_api="blas")
    except Exception:
        return contextlib.nullcontext()


def _linear(p, x):
    return x @ p["w"] + p["b"]


class Transformer:
    """src/autoregressive.py:50-96.  apply(params, None, x): x (..., n, dim) -> logits (..., n, output_size)."""

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

    # -- forward ---------------------------------------------------------------------------------------------
    def _attention(self, params, i, x):
        nm = "%s/layer%d_attn/" % (self.name, i)
        T = x.shape[-2]
        H, K = self.num_heads, self.key_size
        # heads to the front: (..., H, T, K); batched matmuls (numpy's einsum is ~5x slower on these shapes)
        split = lambda y: np.swapaxes(y.reshape(y.shape[:-1] + (H, K)), -2, -3)
        q, k, v = (split(_linear(params[nm + part], x)) for part in ("query", "key", "value"))
        logits = (q @ np.swapaxes(k, -1, -2)) / np.sqrt(K)
        mask = np.tril(np.ones((T, T), dtype=bool))                                # CausalSelfAttention, :26-27
        logits = np.where(mask, logits, -1e30)
        logits = logits - logits.max(axis=-1, keepdims=True)
        w = np.exp(logits); w /= w.sum(axis=-1, keepdims=True)
        attn = np.swapaxes(w @ v, -2, -3)                                          # (..., T, H, K)
        return _linear(params[nm + "linear"], attn.reshape(attn.shape[:-2] + (H * K,)))

    def forward_cache(self, params, x):
        """apply() keeping what the reverse pass needs.  x (B, T, dim)."""
        nm, H, K = self.name, self.num_heads, self.key_size
        x0 = np.asarray(x, dtype=np.float64)
        T = x0.shape[-2]
        h = np.tanh(_linear(params[nm + "/embedding_mlp"], x0))
        cache = {"x0": x0, "h0": h, "layers": []}
        mask = np.tril(np.ones((T, T), dtype=bool))
        for i in range(self.num_layers):
            an = "%s/layer%d_attn/" % (nm, i)
            split = lambda y: np.swapaxes(y.reshape(y.shape[:-1] + (H, K)), -2, -3)       # (B, H, T, K)
            q, k, v = (split(_linear(params[an + part], h)) for part in ("query", "key", "value"))
            lg = np.where(mask, (q @ np.swapaxes(k, -1, -2)) / np.sqrt(K), -1e30)
            lg = lg - lg.max(axis=-1, keepdims=True)
            A = np.exp(lg); A /= A.sum(axis=-1, keepdims=True)
            o = np.swapaxes(A @ v, -2, -3).reshape(h.shape[:-1] + (H * K,))
            h1 = h + _linear(params[an + "linear"], o)
            m = np.tanh(_linear(params["%s/layer%d_mlp/linear" % (nm, i)], h1))
            h2 = h1 + _linear(params["%s/layer%d_mlp/linear_1" % (nm, i)], m)
            cache["layers"].append({"hin": h, "q": q, "k": k, "v": v, "A": A, "o": o, "h1": h1, "m": m})
            h = h2
        th = np.tanh(h)
        y = _linear(params[nm + "/output_mlp"], th)
        cache["th"] = th
        x1hat = np.broadcast_to(params[nm]["x1hat"], y.shape[:-2] + (1, self.output_size))
        return np.concatenate([x1hat, y[..., :-1, :]], axis=-2), cache

    def backward(self, params, cache, dlogits, per_sample):
        """Reverse pass: dlogits (B, T, output_size) -> parameter gradients, with a leading batch axis on every leaf when
        per_sample, otherwise summed over the batch."""
        nm, H, K = self.name, self.num_heads, self.key_size
        B, T = dlogits.shape[0], dlogits.shape[1]
        wsum = ((lambda a, d: np.swapaxes(a, 1, 2) @ d) if per_sample else
                (lambda a, d: a.reshape(-1, a.shape[-1]).T @ d.reshape(-1, d.shape[-1])))
        bsum = (lambda d: d.sum(axis=1)) if per_sample else (lambda d: d.sum(axis=(0, 1)))
        g = {nm: {"x1hat": dlogits[:, 0, :] if per_sample else dlogits[:, 0, :].sum(axis=0)}}
        dy = np.concatenate([dlogits[:, 1:, :], np.zeros((B, 1, self.output_size))], axis=1)
        po = params[nm + "/output_mlp"]
        g[nm + "/output_mlp"] = {"w": wsum(cache["th"], dy), "b": bsum(dy)}
        dh = (dy @ po["w"].T) * (1.0 - cache["th"] ** 2)
        for i in reversed(range(self.num_layers)):
            c = cache["layers"][i]
            an = "%s/layer%d_attn/" % (nm, i)
            p1, p2 = params["%s/layer%d_mlp/linear" % (nm, i)], params["%s/layer%d_mlp/linear_1" % (nm, i)]
            g["%s/layer%d_mlp/linear_1" % (nm, i)] = {"w": wsum(c["m"], dh), "b": bsum(dh)}
            dpre = (dh @ p2["w"].T) * (1.0 - c["m"] ** 2)
            g["%s/layer%d_mlp/linear" % (nm, i)] = {"w": wsum(c["h1"], dpre), "b": bsum(dpre)}
            dh1 = dh + dpre @ p1["w"].T
            pl = params[an + "linear"]
            g[an + "linear"] = {"w": wsum(c["o"], dh1), "b": bsum(dh1)}
            do = np.swapaxes((dh1 @ pl["w"].T).reshape(B, T, H, K), 1, 2)                  # (B, H, T, K)
            dA = do @ np.swapaxes(c["v"], -1, -2)
            dv = np.swapaxes(c["A"], -1, -2) @ do
            dS = c["A"] * (dA - (c["A"] * dA).sum(axis=-1, keepdims=True)) / np.sqrt(K)
            merge = lambda y: np.swapaxes(y, 1, 2).reshape(B, T, H * K)
            dq = merge(dS @ c["k"])
            dk = merge(np.swapaxes(dS, -1, -2) @ c["q"])
            dv = merge(dv)
            dhin = dh1
            for part, d in (("query", dq), ("key", dk), ("value", dv)):
                g[an + part] = {"w": wsum(c["hin"], d), "b": bsum(d)}
                dhin = dhin + d @ params[an + part]["w"].T
            dh = dhin
        dpre0 = dh * (1.0 - cache["h0"] ** 2)
        g[nm + "/embedding_mlp"] = {"w": wsum(cache["x0"], dpre0), "b": bsum(dpre0)}
        return g

    def apply(self, params, rng, x):
        nm = self.name
        x = np.tanh(_linear(params[nm + "/embedding_mlp"], np.asarray(x, dtype=np.float64)))
        for i in range(self.num_layers):
            x = x + self._attention(params, i, x)
            h = np.tanh(_linear(params["%s/layer%d_mlp/linear" % (nm, i)], x))
            x = x + _linear(params["%s/layer%d_mlp/linear_1" % (nm, i)], h)
        x = _linear(params[nm + "/output_mlp"], np.tanh(x))
        x1hat = np.broadcast_to(params[nm]["x1hat"], x.shape[:-2] + (1, self.output_size))
        return np.concatenate([x1hat, x[..., :-1, :]], axis=-2)                    # :93


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

    def __init__(self, engine, network, dim, B):
        from .sr import ravel_pytree
        self.engine, self.network, self.dim, self.B = engine, network, dim, B
        n = sum(int(np.prod(shp)) for leaves in network.param_shapes(dim).values() for shp in leaves.values())
        self.perm = ravel_pytree(unflat_params(network, np.arange(n), dim))[0].astype(np.int32)
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


def make_autoregressive_sampler(network, sp_indices, n, num_states, mask_fn=False, engine=None, host=False):
    """src/sampler.py:4-50 with a leading batch axis built in (the reference vmaps).  sampler(params, key, batch) ->
    (batch, n) int32 sorted state indices; log_prob(params, state_indices (batch, n)) -> (batch,).
    engine (or sampler.attach(engine) later; coulombgas_amd.train does it): a GPU Engine of the same (n, dim) -- the sampler
    and the log-probability then run on the device (cg_van_sample / cg_van_log_prob: one wave per sample, key / value cache
    in LDS) and hand DeviceArrays to the hot path, and the gradients (log_prob.grad -> DeviceScores, log_prob.vjp) come from
    the device's reverse pass (cg_van_scores_*).  Calling the closures with no engine attached raises: the numpy
    restatement below is the checker of the device kernels (tests) and runs only when asked for with host=True."""
    sp_indices = np.asarray(sp_indices, dtype=np.float64)
    base = np.tril(np.ones((n, num_states), dtype=bool), k=num_states - n)
    dev = {"engine": engine}

    def _dev_engine(params):
        eng = dev["engine"]
        if eng is None and not host:
            raise RuntimeError("autoregressive density matrix: no GPU engine attached (pass engine= / call .attach(engine); "
                               "coulombgas_amd.train attaches its own).  host=True selects the numpy restatement used by the tests.")
        if eng is not None:
            eng.van_set_params((num_states, network.num_layers, network.model_size, network.num_heads, network.hidden_size),
                               sp_indices, flat_params(network, params, sp_indices.shape[1]))
        return eng

    def _mask(state_idx):
        state_idx = np.asarray(state_idx)
        idx_lb = np.concatenate([np.full(state_idx.shape[:-1] + (1,), -1), state_idx[..., :-1]], axis=-1)
        return base & (np.arange(num_states) > idx_lb[..., None])

    def _logits(params, state_idx):
        logits = network.apply(params, None, sp_indices[state_idx])
        return np.where(_mask(state_idx), logits, -1e50)

    def sampler(params, key, batch, unif=None):
        eng = _dev_engine(params)
        if eng is not None:
            from .mcmc import _seed_of
            return eng.van_sample_d(batch, 0 if unif is not None else _seed_of(key), 0, unif)[0]
        with _blas_limit():
            return _sampler(params, key, batch, unif)

    def _sampler(params, key, batch, unif=None):
        rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
        state_indices = np.zeros((batch, n), dtype=np.int32)
        for i in range(n):
            # the conditional of electron i needs positions <= i only (causal attention): run the prefix, not all n
            logits = network.apply(params, None, sp_indices[state_indices[:, :i + 1]])[:, i, :]
            logits = np.where(_mask(state_indices)[:, i, :], logits, -1e50)
            u = rng.uniform(size=logits.shape) if unif is None else np.asarray(unif)[:, i, :]
            g = -np.log(-np.log(u))                                                # Gumbel-max = jax.random.categorical
            state_indices[:, i] = np.argmax(logits + g, axis=-1)
        return state_indices

    def log_prob(params, state_idx):
        eng = _dev_engine(params)
        if eng is not None:
            if hasattr(state_idx, "ptr"):
                return eng.van_log_prob_d(state_idx)
            return eng.van_log_prob(state_idx)
        state_idx = np.asarray(state_idx)
        with _blas_limit():
            logits = _logits(params, state_idx)
        m = logits.max(axis=-1, keepdims=True)
        logp = logits - m - np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
        return np.take_along_axis(logp, state_idx[..., None], axis=-1)[..., 0].sum(axis=-1)

    def _dlogits(params, state_idx):
        state_idx = np.asarray(state_idx)                  # (a DeviceArray downloads here: the gradients are host numpy)
        logits, cache = network.forward_cache(params, sp_indices[state_idx])
        logits = np.where(_mask(state_idx), logits, -1e50)
        m = logits.max(axis=-1, keepdims=True)
        p = np.exp(logits - m); p /= p.sum(axis=-1, keepdims=True)
        d = -p
        np.put_along_axis(d, state_idx[..., None], np.take_along_axis(d, state_idx[..., None], axis=-1) + 1.0, axis=-1)
        return d, cache                                       # d log p / d logits = onehot - softmax (0 on masked entries)

    def _dev_scores(params, state_idx):
        eng = _dev_engine(params)
        if eng is None:
            return None
        s_d = state_idx if hasattr(state_idx, "ptr") else eng.asdevice(np.asarray(state_idx, dtype=np.int32), "van_sidx_in", np.int32)
        eng.van_scores_compute_d(s_d)
        return eng

    def grad(params, state_idx):
        """jax.vmap(jax.grad(log_prob), (None, 0), 0): per-sample gradients, every leaf with a leading batch axis (host), or
        a DeviceScores handle with an engine attached."""
        eng = _dev_scores(params, state_idx)
        if eng is not None:
            return DeviceScores(eng, network, sp_indices.shape[1], int(np.shape(state_idx)[0]))
        with _blas_limit():
            d, cache = _dlogits(params, state_idx)
            return network.backward(params, cache, d, per_sample=True)

    def vjp(params, state_idx, w):
        """sum_b w[b] * d log_prob_b / d params  (what jax.jacrev of a weighted sum of log-probabilities returns)."""
        eng = _dev_scores(params, state_idx)
        if eng is not None:
            w_d = w if hasattr(w, "ptr") else eng.asdevice(np.asarray(w, dtype=np.float64), "van_w")
            g = eng.van_scores_vjp_d(w_d, eng.scratch("van_vjp", (DeviceScores(eng, network, sp_indices.shape[1], 0).shape[1],)))
            return unflat_params(network, eng.to_host(g), sp_indices.shape[1])
        with _blas_limit():
            d, cache = _dlogits(params, state_idx)
            return network.backward(params, cache, d * np.asarray(w, dtype=np.float64)[:, None, None], per_sample=False)

    log_prob.grad, log_prob.vjp = grad, vjp

    def attach(engine):
        dev["engine"] = engine
    sampler.attach = log_prob.attach = attach
    if mask_fn:
        return _mask, sampler, log_prob
    return sampler, log_prob


def make_classical_score(log_prob):
    """src/sampler.py:52-65: params, (batch, n) samples -> pytree of per-sample scores d log p / d params."""
    return lambda params, state_indices: log_prob.grad(params, state_indices)

"""Test infrastructure: numpy restatement of the autoregressive Transformer density matrix (src/autoregressive.py:50-96,
src/sampler.py:4-65) -- forward pass, sampler, log-probability and a hand-written reverse pass.  It is the second checker of
the device kernels (csrc/cg_van.hpp) next to oracle/cg_ref.py's torch restatement, and what the CPU-only tests of the host
logic (driver, SR, pre-training loop) run the density matrix on.  The product package never imports it."""
import contextlib
import numpy as np
from coulombgas_amd.autoregressive import Transformer


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


class HostTransformer(Transformer):
    """apply(params, None, x): x (..., n, dim) -> logits (..., n, output_size), plus forward_cache / backward."""

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


def make_host_sampler(network, sp_indices, n, num_states, mask_fn=False):
    """src/sampler.py:4-50 on the host (numpy), with a leading batch axis built in.  sampler(params, key, batch) -> (batch, n)
    int32 sorted state indices; log_prob(params, state_indices) -> (batch,); log_prob.grad / log_prob.vjp from the reverse pass."""
    if not isinstance(network, HostTransformer):          # a product Transformer carries the architecture only
        network = HostTransformer(network.output_size, network.num_layers, network.model_size, network.num_heads,
                                  network.hidden_size, network.name)
    sp_indices = np.asarray(sp_indices, dtype=np.float64)
    base = np.tril(np.ones((n, num_states), dtype=bool), k=num_states - n)

    def _mask(state_idx):
        state_idx = np.asarray(state_idx)
        idx_lb = np.concatenate([np.full(state_idx.shape[:-1] + (1,), -1), state_idx[..., :-1]], axis=-1)
        return base & (np.arange(num_states) > idx_lb[..., None])

    def _logits(params, state_idx):
        logits = network.apply(params, None, sp_indices[state_idx])
        return np.where(_mask(state_idx), logits, -1e50)

    def sampler(params, key, batch, unif=None):
        with _blas_limit():
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
        state_idx = np.asarray(state_idx)
        with _blas_limit():
            logits = _logits(params, state_idx)
        m = logits.max(axis=-1, keepdims=True)
        logp = logits - m - np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
        return np.take_along_axis(logp, state_idx[..., None], axis=-1)[..., 0].sum(axis=-1)

    def _dlogits(params, state_idx):
        state_idx = np.asarray(state_idx)
        logits, cache = network.forward_cache(params, sp_indices[state_idx])
        logits = np.where(_mask(state_idx), logits, -1e50)
        m = logits.max(axis=-1, keepdims=True)
        p = np.exp(logits - m); p /= p.sum(axis=-1, keepdims=True)
        d = -p
        np.put_along_axis(d, state_idx[..., None], np.take_along_axis(d, state_idx[..., None], axis=-1) + 1.0, axis=-1)
        return d, cache                                       # d log p / d logits = onehot - softmax (0 on masked entries)

    def grad(params, state_idx):
        """jax.vmap(jax.grad(log_prob), (None, 0), 0): per-sample gradients, every leaf with a leading batch axis"""
        with _blas_limit():
            d, cache = _dlogits(params, state_idx)
            return network.backward(params, cache, d, per_sample=True)

    def vjp(params, state_idx, w):
        """sum_b w[b] * d log_prob_b / d params"""
        with _blas_limit():
            d, cache = _dlogits(params, state_idx)
            return network.backward(params, cache, d * np.asarray(w, dtype=np.float64)[:, None, None], per_sample=False)

    log_prob.grad, log_prob.vjp = grad, vjp
    if mask_fn:
        return _mask, sampler, log_prob
    return sampler, log_prob

"""
oracle/cg_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by coulombgas_amd/).

CPU restatement, statement by statement, of the reference's VMC hot path
(fermiflow/CoulombGas @ v1) in numpy + torch.func (fp64 / complex128).  The
reference is JAX/Haiku and cannot be imported in this image (no jax wheel, no
network), so every function below follows the cited reference lines and keeps
the reference's *autodiff structure* (jacfwd inside logpsi, jacrev + jvp for
the Laplacian, jacrev for the theta-gradients) with torch.func standing in for
jax transforms.  That makes it an independent check on the hand-derived
derivatives inside the HIP kernels.

PARITY PINNING.  The oracle is pinned (tests/test_oracle_kat.py) by
  * the reference's own analytic known-answer tests
    (tests/test_slater.py:112, tests/test_logpsi.py:106: plane-wave kinetic energy),
  * the reference's symmetry tests (tests/test_flow.py:25,32,38,
    tests/test_slater.py:30,36-37, tests/test_logpsi.py:45,54,72,77,123-124,151),
  * the textbook 2-D Madelung constant and the published potential energies of
    the shipped walkers (data/n_*/epoch_*.pkl vs data.txt, fixtures in tests/golden/).
It is NOT pinned against outputs of the JAX reference itself (not runnable
here) for: theta-gradients of the losses, the complex clip, jax.random streams.
Those rows are "parity unpinned" (DESIGN.md section 3).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.
"""
import math
import numpy as np
import torch
from torch.func import jacfwd, jacrev, jvp, vmap, grad

torch.set_default_dtype(torch.float64)
PI = math.pi


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


# --------------------------------------------------------------------------- #
# flow parameters: Haiku tree <-> flat vector (jax.flatten_util.ravel_pytree order)
# --------------------------------------------------------------------------- #
def flow_layer_names(depth):
    """Haiku names (src/flow.py:11-14,54): splayers are created first in
    __init__ -> '~/linear', '~/linear_1', ...; then tplayers; `final` is created
    in __call__ -> 'fermi_net/linear'."""
    def nm(i):
        return "fermi_net/~/linear" + ("" if i == 0 else "_%d" % i)
    sp = [nm(i) for i in range(depth)]
    tp = [nm(depth + i) for i in range(depth - 1)]
    return sp, tp, "fermi_net/linear"


def flow_param_shapes(depth, spsize, tpsize, dim):
    sp, tp, fin = flow_layer_names(depth)
    shapes = {}
    # layer 0 input: f = [s0 (dim), mean s0 (dim), mean_j t0 (2dim+1)]
    shapes[sp[0]] = (4 * dim + 1, spsize)
    for i in range(1, depth):
        shapes[sp[i]] = (2 * spsize + tpsize, spsize)
    shapes[tp[0]] = (2 * dim + 1, tpsize)
    for i in range(1, depth - 1):
        shapes[tp[i]] = (tpsize, tpsize)
    shapes[fin] = (spsize, dim)
    return shapes


def flow_ravel_order(depth, spsize, tpsize, dim):
    """[(module, leaf, shape)] in ravel_pytree order: sorted module names, 'b' before 'w'."""
    shapes = flow_param_shapes(depth, spsize, tpsize, dim)
    out = []
    for name in sorted(shapes):
        fin, fout = shapes[name]
        out.append((name, "b", (fout,)))
        out.append((name, "w", (fin, fout)))
    return out


def flow_unravel(theta, depth, spsize, tpsize, dim):
    params, off = {}, 0
    for name, leaf, shp in flow_ravel_order(depth, spsize, tpsize, dim):
        sz = int(np.prod(shp))
        params.setdefault(name, {})[leaf] = theta[off:off + sz].reshape(shp)
        off += sz
    assert off == theta.shape[0]
    return params


def flow_ravel(params, depth, spsize, tpsize, dim):
    return torch.cat([T(params[n][l]).reshape(-1)
                      for n, l, _ in flow_ravel_order(depth, spsize, tpsize, dim)])


def flow_init(rng, depth, spsize, tpsize, dim, init_stddev=0.01):
    """N(0, init_stddev^2) weights, zero biases (src/flow.py:6-14,54).  numpy Generator, not jax.random."""
    th = []
    for _, leaf, shp in flow_ravel_order(depth, spsize, tpsize, dim):
        th.append(np.zeros(shp).ravel() if leaf == "b" else init_stddev * rng.standard_normal(shp).ravel())
    return np.concatenate(th)


def softplus(u):
    # jax.nn.softplus(x) = logaddexp(x, 0): no threshold
    return torch.logaddexp(u, torch.zeros_like(u))


class FermiNet:
    """src/flow.py:5-55"""

    def __init__(self, depth, spsize, tpsize, L):
        self.depth, self.spsize, self.tpsize, self.L = depth, spsize, tpsize, L
        self.sp_names, self.tp_names, self.final_name = flow_layer_names(depth)

    def _tpstream0(self, x):                       # src/flow.py:20-26
        n, _ = x.shape
        rij = x[:, None, :] - x
        cos_rij, sin_rij = torch.cos(2 * PI / self.L * rij), torch.sin(2 * PI / self.L * rij)
        eye = torch.eye(n, dtype=x.dtype)
        dij = torch.linalg.norm(torch.sin(PI / self.L * rij) + eye[..., None], dim=-1) * (1.0 - eye)
        return torch.cat((cos_rij, sin_rij, dij[..., None]), dim=-1)

    def _f(self, sp, tp):                           # src/flow.py:28-37
        n, _ = sp.shape
        return torch.cat((sp, sp.mean(dim=0, keepdim=True).expand(n, -1), tp.mean(dim=1)), dim=-1)

    def apply(self, params, x):                     # src/flow.py:39-55
        lin = lambda name, a: a @ params[name]["w"] + params[name]["b"]
        sp, tp = torch.zeros_like(x), self._tpstream0(x)
        for i in range(self.depth - 1):
            f = self._f(sp, tp)
            if i == 0:
                sp = softplus(lin(self.sp_names[i], f))
                tp = softplus(lin(self.tp_names[i], tp))
            else:
                sp = sp + softplus(lin(self.sp_names[i], f))
                tp = tp + softplus(lin(self.tp_names[i], tp))
        f = self._f(sp, tp)
        sp = sp + softplus(lin(self.sp_names[-1], f))
        return x + lin(self.final_name, sp)


class IdentityFlow:
    """hk.transform(lambda x: x) of tests/test_logpsi.py:88"""
    def apply(self, params, x):
        return x


# --------------------------------------------------------------------------- #
# src/slater.py
# --------------------------------------------------------------------------- #
def logslaterdet(indices, x, L):
    """src/slater.py:4-19 (complex log det; the custom JVP at :23-44 is mathematically the
    derivative of this expression -- checked by the reference's tests/test_slater.py:59-79)."""
    k = 2 * PI / L * indices
    k_dot_x = (k * x[:, None, :]).sum(dim=-1)
    _, dim = x.shape
    D = 1 / L ** (dim / 2) * torch.exp(1j * k_dot_x)
    phase, logabsdet = torch.linalg.slogdet(D)
    return logabsdet + torch.log(phase)


# --------------------------------------------------------------------------- #
# src/logpsi.py
# --------------------------------------------------------------------------- #
def make_logpsi(flow, sp_indices, L):                 # src/logpsi.py:7-33
    sp_indices = T(sp_indices)

    def logpsi(x, params, state_idx):
        z = flow.apply(params, x)
        log_phi = logslaterdet(sp_indices[state_idx], z, L)
        n, dim = x.shape
        x_flatten = x.reshape(-1)
        flow_flatten = lambda xf: flow.apply(params, xf.reshape(n, dim)).reshape(-1)
        jac = jacfwd(flow_flatten)(x_flatten)
        _, logjacdet = torch.linalg.slogdet(jac)
        return torch.stack([log_phi.real + 0.5 * logjacdet, log_phi.imag])

    return logpsi


def make_logphi_logjacdet(flow, sp_indices, L):       # src/logpsi.py:35-53
    sp_indices = T(sp_indices)

    def logphi(x, params, state_idx):
        z = flow.apply(params, x)
        log_phi = logslaterdet(sp_indices[state_idx], z, L)
        return torch.stack([log_phi.real, log_phi.imag])

    def logjacdet(x, params):
        n, dim = x.shape
        x_flatten = x.reshape(-1)
        flow_flatten = lambda xf: flow.apply(params, xf.reshape(n, dim)).reshape(-1)
        jac = jacfwd(flow_flatten)(x_flatten)
        _, ld = torch.linalg.slogdet(jac)
        return 0.5 * ld

    return logphi, logjacdet


def make_logpsi_grad_laplacian(logpsi, forloop=True, hutchinson=False, logphi=None, logjacdet=None):
    """src/logpsi.py:55-172.  The Hutchinson vector `v` is an explicit input here
    (the reference draws it with jax.random.normal(key, x.shape), :110)."""

    def logpsi_vmapped(x, params, state_idx):           # :58-61
        out = torch.stack([logpsi(x[b], params, state_idx[b]) for b in range(x.shape[0])])
        return torch.complex(out[:, 0], out[:, 1])

    def _one_exact(x, params, state_idx):               # :63-106
        g = jacrev(logpsi)(x, params, state_idx)
        g = torch.complex(g[0], g[1])
        n, dim = x.shape
        x_flatten = x.reshape(-1)
        grad_logpsi = jacrev(lambda xf: logpsi(xf.reshape(n, dim), params, state_idx))
        eye = torch.eye(x_flatten.shape[0])
        if forloop:                                      # :86-92
            lap = torch.zeros((), dtype=torch.complex128)
            for i in range(x_flatten.shape[0]):
                _, tangent = jvp(grad_logpsi, (x_flatten,), (eye[i],))
                lap = lap + torch.complex(tangent[0, i], tangent[1, i])
        else:                                            # :93-100
            def body_fun(xf, basevec):
                _, tangent = jvp(grad_logpsi, (xf,), (basevec,))
                return (tangent * basevec).sum(dim=-1)
            lap2 = vmap(body_fun, (None, 1), 1)(x_flatten, eye).sum(dim=-1)
            lap = torch.complex(lap2[0], lap2[1])
        return g, lap

    def _one_hutch_full(x, params, state_idx, v):       # :112-132
        g, hvp = jvp(jacrev(lambda xx: logpsi(xx, params, state_idx)), (x,), (v,))
        g = torch.complex(g[0], g[1])
        rl = (hvp * v).sum(dim=(-2, -1))
        return g, torch.complex(rl[0], rl[1])

    def _one_hutch_split(x, params, state_idx, v):      # :134-164
        g_phi = jacrev(logphi)(x, params, state_idx)
        g_phi = torch.complex(g_phi[0], g_phi[1])
        g_jac, hvp = jvp(grad(lambda xx: logjacdet(xx, params)), (x,), (v,))
        g = g_phi + g_jac
        n, dim = x.shape
        x_flatten = x.reshape(-1)
        grad_logphi = jacrev(lambda xf: logphi(xf.reshape(n, dim), params, state_idx))
        eye = torch.eye(x_flatten.shape[0])
        lap = torch.zeros((), dtype=torch.complex128)
        for i in range(x_flatten.shape[0]):
            _, tangent = jvp(grad_logphi, (x_flatten,), (eye[i],))
            lap = lap + torch.complex(tangent[0, i], tangent[1, i])
        random_logjacdet = (hvp * v).sum(dim=(-2, -1))
        return g, lap + random_logjacdet

    def fn(x, params, state_indices, v=None):
        B = x.shape[0]
        gs, ls = [], []
        for b in range(B):
            if not hutchinson:
                g, l = _one_exact(x[b], params, state_indices[b])
            elif logphi is None and logjacdet is None:
                g, l = _one_hutch_full(x[b], params, state_indices[b], v[b])
            else:
                g, l = _one_hutch_split(x[b], params, state_indices[b], v[b])
            gs.append(g); ls.append(l)
        return torch.stack(gs), torch.stack(ls)

    return logpsi_vmapped, fn


def make_logp(logpsi):                                  # src/logpsi.py:174-181
    def logp(x, params, state_idx):
        return torch.stack([2 * logpsi(x[b], params, state_idx[b])[0] for b in range(x.shape[0])])
    return logp


def make_quantum_score(logpsi_theta):                   # src/logpsi.py:183-203
    """logpsi_theta(x, theta_flat, state_idx) -> (2,).  Returns (B, P) complex per-sample scores."""
    def quantum_score_fn(x, theta, state_idx):
        out = []
        for b in range(x.shape[0]):
            j = jacrev(lambda th: logpsi_theta(x[b], th, state_idx[b]))(theta)
            out.append(torch.complex(j[0], j[1]))
        return torch.stack(out)
    return quantum_score_fn


# --------------------------------------------------------------------------- #
# src/MCMC.py
# --------------------------------------------------------------------------- #
def mcmc(logp_fn, x_init, noise, unif, mc_steps, mc_stddev=0.02):
    """src/MCMC.py:22-39 with the normal / uniform draws supplied as arrays
    noise (steps,B,n,dim), unif (steps,B) instead of jax.random.  Returns x, logp, accept_rate
    (the single-device value; the pmean at :39 is the caller's)."""
    x = x_init
    logp = logp_fn(x_init)
    num_accepts = 0.0
    for i in range(mc_steps):
        x_proposal = x + mc_stddev * noise[i]
        logp_proposal = logp_fn(x_proposal)
        ratio = torch.exp(logp_proposal - logp)
        accept = unif[i] < ratio
        x = torch.where(accept[:, None, None], x_proposal, x)
        logp = torch.where(accept, logp_proposal, logp)
        num_accepts += float(accept.sum())
    batch = x.shape[0]
    return x, logp, num_accepts / (mc_steps * batch)


def wrap(x, L):                                         # src/VMC.py:24
    return x - L * torch.floor(x / L)


# --------------------------------------------------------------------------- #
# src/potential.py
# --------------------------------------------------------------------------- #
def kpoints(dim, Gmax):                                 # src/potential.py:7-17
    n = np.arange(-Gmax, Gmax + 1)
    nis = np.meshgrid(*([n] * dim))
    G = np.array([ni.flatten() for ni in nis]).T
    G2 = (G ** 2).sum(axis=-1)
    G = G[(G2 <= Gmax ** 2) * (G2 > 0)]
    return G


def _gk_g0(dim, kappa, G):
    Gnorm = torch.linalg.norm(T(G), dim=-1)
    if dim == 3:                                        # :54-56
        g_k = torch.exp(-PI ** 2 * Gnorm ** 2 / kappa ** 2) / (PI * Gnorm ** 2)
        g_0 = -PI / kappa ** 2
    elif dim == 2:                                      # :57-59
        g_k = torch.erfc(PI * Gnorm / kappa) / Gnorm
        g_0 = -2 * math.sqrt(PI) / kappa
    return g_k, g_0


def Madelung(dim, kappa, G):                            # src/potential.py:19-34
    g_k, g_0 = _gk_g0(dim, kappa, G)
    return float(g_k.sum() + g_0 - 2 * kappa / math.sqrt(PI))


def psi(x, kappa, G):                                   # src/potential.py:36-65
    n, dim = x.shape
    i, j = np.triu_indices(n, k=1)
    rij = (x[:, None, :] - x)[i, j]
    rij = rij - torch.round(rij)                        # jnp.rint = round-half-even = torch.round
    dij = torch.linalg.norm(rij, dim=-1)
    V_shortrange = (torch.erfc(kappa * dij) / dij).sum()
    g_k, g_0 = _gk_g0(dim, kappa, G)
    V_longrange = (g_k * torch.cos(2 * PI * T(G) @ rij.T).sum(dim=-1)).sum() + g_0 * rij.shape[0]
    return V_shortrange + V_longrange


def potential_energy(x, kappa, G, L, rs):               # src/potential.py:69-77
    return torch.stack([2 * rs / L * psi(x[b] / L, kappa, G) for b in range(x.shape[0])])


# --------------------------------------------------------------------------- #
# src/VMC.py:31-80 + main.py:277-298
# --------------------------------------------------------------------------- #
def complex_clip(a, lo, hi):
    """jnp.clip(a, lo, hi) = minimum(maximum(a, lo), hi) with the lexicographic complex
    ordering of the JAX generation the reference targets."""
    lo_c = torch.complex(T(lo), torch.zeros(()))
    hi_c = torch.complex(T(hi), torch.zeros(()))
    def lex_lt(p, q):
        return (p.real < q.real) | ((p.real == q.real) & (p.imag < q.imag))
    m = torch.where(lex_lt(a, lo_c), lo_c, a)
    return torch.where(lex_lt(hi_c, m), hi_c, m)


def observables_and_weights(logp_states, grad_x, laplacian, potential, Vconst, beta):
    """src/VMC.py:39-58,63-64,72-73 on one device (pmean over a single shard = identity)."""
    kinetic = -laplacian - (grad_x ** 2).sum(dim=(-2, -1))
    pot = potential + Vconst
    Eloc = kinetic + pot
    Floc = logp_states / beta + Eloc.real
    obs = {"K_mean": kinetic.real.mean(), "K2_mean": (kinetic.real ** 2).mean(),
           "V_mean": pot.mean(), "V2_mean": (pot ** 2).mean(),
           "E_mean": Eloc.real.mean(), "E2_mean": (Eloc.real ** 2).mean(),
           "F_mean": Floc.mean(), "F2_mean": (Floc ** 2).mean(),
           "S_mean": -logp_states.mean(), "S2_mean": (logp_states ** 2).mean()}
    tvF = (Floc - obs["F_mean"]).abs().mean()
    Floc_clipped = torch.clamp(Floc, obs["F_mean"] - 5.0 * tvF, obs["F_mean"] + 5.0 * tvF)
    tvE = (Eloc - obs["E_mean"]).abs().mean()
    Eloc_clipped = complex_clip(Eloc, obs["E_mean"] - 5.0 * tvE, obs["E_mean"] + 5.0 * tvE)
    return obs, Eloc, Floc, Floc_clipped, Eloc_clipped


def quantum_loss_and_grads(logpsi_theta, theta, x, state_indices, Eloc_clipped):
    """src/VMC.py:69-76 + jax.jacrev(quantum_lossfn) of main.py:278.
    Returns (gradF_theta, quantum_score) values and their theta-gradients (P,), (P,)."""
    def lossfn(th):
        out = torch.stack([logpsi_theta(x[b], th, state_indices[b]) for b in range(x.shape[0])])
        logpsix = torch.complex(out[:, 0], out[:, 1])
        gradF_theta = 2 * (logpsix * Eloc_clipped.conj()).real.mean()
        quantum_score = 2 * logpsix.real.mean()
        return torch.stack([gradF_theta, quantum_score])
    vals = lossfn(theta)
    J = jacrev(lossfn)(theta)
    return vals[0], vals[1], J[0], J[1]


# --------------------------------------------------------------------------- #
# src/sr.py
# --------------------------------------------------------------------------- #
def sr_solve_and_clip(fisher, grads_raveled, damping, max_norm):
    """src/sr.py:38-45 and :102-117 (same statements for the classical and the quantum block), numpy."""
    fisher = fisher + damping * np.eye(fisher.shape[0])
    updates_raveled = np.linalg.solve(fisher, grads_raveled)
    gnorm = np.sum(grads_raveled * updates_raveled)
    scale = np.minimum(np.sqrt(max_norm / gnorm), 1)
    return -scale * updates_raveled


def hybrid_fisher_sr_update(classical_score, quantum_score, grad_van, grad_flow, damping, max_norm):
    """src/sr.py:62-122 on one device (the pmeans are identities): classical_score (B,Pv) real, quantum_score (B,Pf)
    complex, raveled gradients.  Returns (classical_fisher, quantum_fisher, quantum_score_mean, update_van, update_flow)."""
    B = classical_score.shape[0]
    quantum_score_mean = quantum_score.mean(axis=0)                                   # :70
    classical_fisher = classical_score.T.dot(classical_score) / B                     # :74
    quantum_fisher = quantum_score.conj().T.dot(quantum_score).real / B               # :76
    qf = quantum_fisher - (quantum_score_mean.conj()[:, None] * quantum_score_mean).real      # :88
    return (classical_fisher, quantum_fisher, quantum_score_mean,
            sr_solve_and_clip(classical_fisher, grad_van, damping, max_norm),
            sr_solve_and_clip(qf, grad_flow, damping, max_norm))


# --------------------------------------------------------------------------- #
# src/autoregressive.py + src/sampler.py (torch restatement: autograd gives the checker's gradients)
# --------------------------------------------------------------------------- #
def transformer_apply(params, x, num_layers, num_heads):
    """src/autoregressive.py:70-96 for one sample x (n, dim); params: dict module -> leaf -> torch tensor."""
    nm = "transformer"
    lin = lambda p, y: y @ p["w"] + p["b"]
    x = torch.tanh(lin(params[nm + "/embedding_mlp"], x))
    T = x.shape[0]
    mask = torch.tril(torch.ones((T, T)))[None]
    for i in range(num_layers):
        an = "%s/layer%d_attn/" % (nm, i)
        ms = x.shape[-1]; K = ms // num_heads
        q, k, v = (lin(params[an + part], x).reshape(T, num_heads, K) for part in ("query", "key", "value"))
        lg = torch.einsum("thd,Thd->htT", q, k) / np.sqrt(K)
        lg = torch.where(mask > 0, lg, torch.tensor(-1e30, dtype=lg.dtype))
        w = torch.softmax(lg, dim=-1)
        o = torch.einsum("htT,Thd->thd", w, v).reshape(T, ms)
        x = x + lin(params[an + "linear"], o)
        h = torch.tanh(lin(params["%s/layer%d_mlp/linear" % (nm, i)], x))
        x = x + lin(params["%s/layer%d_mlp/linear_1" % (nm, i)], h)
    x = lin(params[nm + "/output_mlp"], torch.tanh(x))
    return torch.vstack((params[nm]["x1hat"][None], x[:-1]))


def autoregressive_log_prob(params, state_idx, sp_indices, num_layers, num_heads):
    """src/sampler.py:6-46 for one sample (state_idx (n,) int64)."""
    n, num_states = state_idx.shape[0], sp_indices.shape[0]
    logits = transformer_apply(params, sp_indices[state_idx], num_layers, num_heads)
    mask = torch.tril(torch.ones((n, num_states)), diagonal=num_states - n)
    idx_lb = torch.cat((torch.tensor([-1]), state_idx[:-1]))
    mask = torch.where(torch.arange(num_states) > idx_lb[:, None], mask, torch.zeros(()))
    logits = torch.where(mask > 0, logits, torch.tensor(-1e50, dtype=logits.dtype))
    logp = torch.log_softmax(logits, dim=-1)
    return logp[torch.arange(n), state_idx].sum()

#!/usr/bin/env python3
"""
Generates the DATA fixtures that come from the reference tree (run once, in the build
container where /root/reference is mounted; the outputs are committed, this script never
runs on the GPU box):

  coulombgas_amd/data/orbitals_dim2_Emax{25,36,49}.npy   (shipped with the package) twisted, sorted and reversed single-particle table
                                     = `sp_indices_twist` of main.py:79-90 (twist 1/4,1/4)
  orbitals_dim3_Emax60.npy           untwisted `sp_orbitals(3)[0]` used by the reference tests
                                     (tests/test_slater.py:17, tests/test_logpsi.py:33)
  shipped_n29_rs10.npz / shipped_n29_rs1.npz / shipped_n57_rs10.npz
                                     a slice of the shipped walkers (`x` of epoch_*.pkl), the trained
                                     flow parameters and the matching published data.txt row.

Orbital ordering: `sp_orbitals` sorts degenerate integer energies with
numpy's *unstable* default argsort (src/orbitals.py:42).  The shipped models only reproduce
their published free energies with the legacy (non-SIMD) quicksort tie order followed by a
*stable* sort for the twisted energies (in main.py `twist` is a jnp array, so
src/orbitals.py:53 runs JAX's stable argsort).  The legacy order is obtained by importing the
reference's own src/orbitals.py in a child process with numpy's SIMD sort kernels disabled.
"""
import os, sys, subprocess, pickle, glob, json
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(os.path.dirname(HERE)), "coulombgas_amd", "data")     # orbital tables ship with the package

CHILD = r"""
import importlib.util, numpy as np, sys, json
spec = importlib.util.spec_from_file_location("ref_orbitals", "%s/src/orbitals.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
dim, Emax = int(sys.argv[1]), int(sys.argv[2])
idx, Es = m.sp_orbitals(dim, Emax)
np.save(sys.argv[3], idx)
""" % REF


def legacy_sp_orbitals(dim, Emax):
    env = dict(os.environ)
    env["NPY_DISABLE_CPU_FEATURES"] = ("AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL "
                                       "AVX512_ICL AVX512_SPR AVX2 FMA3")
    tmp = "/tmp/_orb_%d_%d.npy" % (dim, Emax)
    subprocess.check_call([sys.executable, "-c", CHILD, str(dim), str(Emax), tmp], env=env)
    return np.load(tmp)


def twisted_table(dim, Emax, twist):
    idx = legacy_sp_orbitals(dim, Emax)
    it = idx + np.asarray(twist)                       # src/orbitals.py:51
    Es = (it ** 2).sum(axis=-1)                        # :52
    order = np.argsort(Es, kind="stable")              # :53 under jnp (stable)
    return it[order][::-1].copy()                      # main.py:90 reversal


class _Stub:
    def __init__(self, *a, **k):
        pass


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        if mod.startswith("optax") or mod.startswith("jax"):
            if name == "_reconstruct_array":
                def f(fun, args, arr_state, aval_state):
                    a = fun(*args); a.__setstate__(arr_state); return a
                return f
            return _Stub
        return super().find_class(mod, name)


def shipped(n, rs, epoch, nwalk, out):
    d = glob.glob("%s/data/n_%d_dim_2_rs_%s_*" % (REF, n, rs))[0]
    ck = _Unpickler(open(os.path.join(d, "epoch_%06d.pkl" % epoch), "rb")).load()
    x = np.asarray(ck["x"]).reshape(-1, n, 2)
    row = np.loadtxt(os.path.join(d, "data.txt"))[epoch - 1]
    first = np.loadtxt(os.path.join(d, "data.txt"))[0]
    pf = ck["params_flow"]
    # ravel_pytree order: sorted module names, b before w
    theta = np.concatenate([np.asarray(pf[k][l]).ravel() for k in sorted(pf) for l in ("b", "w")])
    np.savez_compressed(os.path.join(HERE, out), x=x[:nwalk], theta=theta,
                        data_row=row, data_row_epoch1=first, n=n, rs=float(rs), epoch=epoch,
                        n_walkers_total=x.shape[0])
    print(out, x[:nwalk].shape, theta.shape, row)


def shipped_run(n, rs, nwalk):
    """One production run of data/n_*: a slice of the final walkers, the trained flow and Transformer parameters and the last
    published data.txt row, in ONE npz under tests/golden/shipped_runs/ (all 18 runs: round-3 widening of the reference-held pin)."""
    d = glob.glob("%s/data/n_%d_dim_2_rs_%s_*" % (REF, n, rs))[0]
    pk = sorted(glob.glob(os.path.join(d, "epoch_*.pkl")))[-1]
    epoch = int(os.path.basename(pk)[6:12])
    ck = _Unpickler(open(pk, "rb")).load()
    x = np.asarray(ck["x"]).reshape(-1, n, 2)
    rows = np.loadtxt(os.path.join(d, "data.txt"))
    pf, pv = ck["params_flow"], ck["params_van"]
    theta = np.concatenate([np.asarray(pf[k][l]).ravel() for k in sorted(pf) for l in ("b", "w")])
    flat = {"van|%s|%s" % (m, l): np.asarray(v) for m in pv for l, v in pv[m].items()}
    Emax = int(os.path.basename(d).split("_Emax_")[1].split("_")[0])
    os.makedirs(os.path.join(HERE, "shipped_runs"), exist_ok=True)
    out = os.path.join(HERE, "shipped_runs", "n%d_rs%s.npz" % (n, rs))
    np.savez_compressed(out, x=x[:nwalk], theta=theta, data_row=rows[epoch - 1], n=n, rs=float(rs), Emax=Emax, epoch=epoch,
                        n_walkers_total=x.shape[0], **flat)
    print(out, x[:nwalk].shape, os.path.getsize(out) // 1024, "KB", rows[epoch - 1][:4])


def pretrained_free_energies():
    """the pretrained free-fermion Transformers of n = 49 / 57 (data/freefermion/pretraining) with their last data.txt row"""
    for n in (49, 57):
        d = glob.glob("%s/data/freefermion/pretraining/n_%d_*/*" % (REF, n))[0]
        van_fixture(os.path.join(d, "params_van.pkl"), None, "pretrained_van_n%d.npz" % n,
                    {"data_row_last": np.loadtxt(os.path.join(d, "data.txt"))[-1]})


def van_fixture(src_pkl, key, out, extra):
    """params_van of a shipped model (pretrained free-fermion model or a checkpoint) as flat npz: 'module|leaf' -> array."""
    ck = _Unpickler(open(src_pkl, "rb")).load()
    pv = ck if key is None else ck[key]
    flat = {"%s|%s" % (m, l): np.asarray(v) for m in pv for l, v in pv[m].items()}
    np.savez_compressed(os.path.join(HERE, out), **flat, **extra)
    print(out, len(flat), "leaves")


if __name__ == "__main__":
    if "--runs" in sys.argv:                   # round 3: all 18 production runs + the two larger pretrained models
        for n, nw in ((29, 192), (49, 128), (57, 128)):
            for rs in ("0.25", "0.5", "1.0", "3.0", "5.0", "10.0"):
                shipped_run(n, rs, nw)
        pretrained_free_energies()
        sys.exit(0)
    d = glob.glob("%s/data/freefermion/pretraining/n_13_*/*" % REF)[0]
    van_fixture(os.path.join(d, "params_van.pkl"), None, "pretrained_van_n13.npz",
                {"data_row_last": np.loadtxt(os.path.join(d, "data.txt"))[-1]})
    d = glob.glob("%s/data/n_29_dim_2_rs_10.0_*" % REF)[0]
    van_fixture(os.path.join(d, "epoch_003000.pkl"), "params_van", "shipped_n29_rs10_van.npz", {})
    # Transformers of the other two statistical KATs (BASELINE configs 4 and 5) and the pretrained n = 29 model the
    # production runs start from (epoch-1 rows of data.txt: flow ~ identity)
    d = glob.glob("%s/data/n_29_dim_2_rs_1.0_*" % REF)[0]
    van_fixture(os.path.join(d, "epoch_003000.pkl"), "params_van", "shipped_n29_rs1_van.npz", {})
    d = glob.glob("%s/data/n_57_dim_2_rs_10.0_*" % REF)[0]
    van_fixture(os.path.join(d, "epoch_005000.pkl"), "params_van", "shipped_n57_rs10_van.npz", {})
    d = glob.glob("%s/data/freefermion/pretraining/n_29_*/*" % REF)[0]
    van_fixture(os.path.join(d, "params_van.pkl"), None, "pretrained_van_n29.npz",
                {"data_row_last": np.loadtxt(os.path.join(d, "data.txt"))[-1]})
    for Emax in (25, 36, 49):
        t = twisted_table(2, Emax, (0.25, 0.25))
        np.save(os.path.join(DATA, "orbitals_dim2_Emax%d.npy" % Emax), t)
        print("Emax", Emax, t.shape)
    np.save(os.path.join(DATA, "orbitals_dim3_Emax60.npy"), legacy_sp_orbitals(3, 60).astype(np.float64))
    shipped(29, "10.0", 3000, 512, "shipped_n29_rs10.npz")
    shipped(29, "1.0", 3000, 512, "shipped_n29_rs1.npz")
    shipped(57, "10.0", 5000, 256, "shipped_n57_rs10.npz")

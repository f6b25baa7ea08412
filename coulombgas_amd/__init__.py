"""coulombgas_amd -- MI355X-native VMC hot path of fermiflow/CoulombGas.

Exports the names main.py takes from the reference's `src` package for this path
(src/__init__.py:1-13) plus the SR optimizer (src/sr.py), the checkpoint format, the autoregressive Transformer density
matrix (src/autoregressive.py, src/sampler.py: sampler, log-probability and gradients on the GPU) and the free-fermion
pre-training (src/freefermion/pretraining.py)."""
from .orbitals import sp_orbitals, twist_sort
from .flow import FermiNet
from .potential import kpoints, Madelung, potential_energy
from .logpsi import (make_logpsi, make_logphi_logjacdet, make_logpsi_grad_laplacian, make_logp,
                     make_quantum_score)
from .mcmc import mcmc
from .vmc import sample_stateindices_and_x, make_loss, make_observable
from .sr import fisher_sr, hybrid_fisher_sr, apply_updates
from .driver import train, make_update, adam, GroundStateSampler
from .checkpoint import ckpt_filename, load_data, save_data, pretrained_model_filename
from .autoregressive import Transformer, make_autoregressive_sampler, make_classical_score
from .freefermion import pretrain, exact_free_energy
from .utils import shard, replicate
from .engine import Engine

__all__ = ["sp_orbitals", "twist_sort", "FermiNet", "kpoints", "Madelung", "potential_energy", "make_logpsi", "make_logphi_logjacdet",
           "make_logpsi_grad_laplacian", "make_logp", "make_quantum_score", "mcmc",
           "sample_stateindices_and_x", "make_loss", "make_observable", "fisher_sr", "hybrid_fisher_sr", "apply_updates", "train", "make_update", "adam", "GroundStateSampler",
           "ckpt_filename", "load_data", "save_data", "pretrained_model_filename", "Transformer", "make_autoregressive_sampler", "make_classical_score", "pretrain", "exact_free_energy",
           "shard", "replicate", "Engine"]

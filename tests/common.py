"""Shared helpers for the parity tests: seeded synthetic inputs -- the package's own generators."""
import os
from coulombgas_amd.synthetic import orbitals, box_length, flow_theta, state_indices, walkers  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

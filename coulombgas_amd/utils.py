"""src/utils.py counterparts.  The reference places arrays with an identity pmap (`shard`) and
broadcasts pytrees (`replicate`) inside ONE process driving 8 devices.  Here every GPU has its own
process (RANK/LOCAL_RANK), so the leading device axis disappears: `shard` returns this rank's slice,
`replicate` returns the pytree unchanged."""
import os
import numpy as np

_DEVICE = None


def current_device():
    global _DEVICE
    if _DEVICE is None:
        _DEVICE = int(os.environ.get("LOCAL_RANK", "0"))
    return _DEVICE


def set_device(d):
    global _DEVICE
    _DEVICE = int(d)


def shard(x, rank=None, world=None):
    """x has the reference's leading device axis (world, ...): return this rank's block (src/utils.py:4)."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    x = np.asarray(x)
    if x.shape[0] != world:
        raise ValueError("leading axis %d != number of devices %d (main.py:214-215)" % (x.shape[0], world))
    return x[rank]


def replicate(pytree, num_devices=None):
    """src/utils.py:6-8: every rank already holds the full pytree."""
    return pytree

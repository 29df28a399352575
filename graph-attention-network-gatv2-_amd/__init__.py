"""gatv2_amd — MI355X-native GATv2 edge-centric hot path behind a C ABI.

The directory name is fixed by the build contract (and is not an importable identifier); load it
with ``__graft_entry__.load_package()``, which registers it as ``gatv2_amd``.

Contents: ``csrc/`` HIP kernels + C ABI (libgatv2_hip.so), ``host/`` the C++ ``train_edge``
drop-in, ``abi.py`` ctypes binding, ``synth.py`` synthetic datasets, ``shard.py`` destination-range
sharding over torch.distributed.  No CPU fallback exists here by design.
"""
from . import abi, shard, synth  # noqa: F401
from .abi import GatContext, GatError, GatLibraryError  # noqa: F401

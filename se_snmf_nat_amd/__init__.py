"""se_snmf_nat_amd -- MI355X-native sparse-NMF multiplicative-update engine.

Drop-in for the hot path of lordet01/SE_SNMF_NAT (src/sparse_nmf.m, src/sparse_nmf_GPU.m and
the basis-training loop of run_basis_DNMF.m) behind the same function signature.  The compute
lives in libsnmf_hip.so (hand-written HIP for gfx950, C ABI in include/snmf.h); this package is
the host-side mirror of the reference interface.  There is no CPU fallback.
"""
from .api import (Context, Plan, SnmfError, default_context, dnmf_adapt, run_basis_dnmf, snmf_mdi, snmf_mdi_Sm,  # noqa: F401
                  sparse_nmf, sparse_nmf_GPU)



def release_device_lists():
    """Give back what the multi-device entries keep per device list for the life of the process (contexts, pinned bounce
    buffers, gather buffers: snmf_multi_release_cache in include/snmf.h).  Returns the number of idle teams destroyed."""
    from . import _lib
    return int(_lib.load().snmf_multi_release_cache())


__version__ = "0.1.0"

"""hxv: MI355X-native sector Hamiltonian x vector engine behind CDMFT-LANC-ED's spHtimesV_p.

Layout: csrc/ (HIP kernels + C-ABI, built into lib/libhxv.so), fortran/ (ISO_C_BINDING glue for
the reference's own host code), hxv/ (this Python mirror of the reference interface).
"""
from .engine import HxvError, HxvSector, LIB_PATH, load_library, EXPORTS, pool_stats, pool_trim  # noqa: F401
from .hamiltonian import EDContext  # noqa: F401
from . import models  # noqa: F401
from .distributed import ShardedHxv, ShardedLanczos, TransposedHxv, dw_split, sharded_eigh_lowest, start_vector_slab  # noqa: F401

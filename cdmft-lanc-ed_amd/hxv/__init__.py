"""hxv: MI355X-native sector Hamiltonian x vector engine behind CDMFT-LANC-ED's spHtimesV_p.

Layout: csrc/ (HIP kernels + C-ABI, built into lib/libhxv.so), fortran/ (ISO_C_BINDING glue for
the reference's own host code), hxv/ (this Python mirror of the reference interface).
"""
from .engine import (HxvError, HxvSector, LIB_PATH, LocalGroup, RcclGroup, halo_plan_from_csr, load_library, EXPORTS, pool_stats, pool_trim, run_ranks,  # noqa: F401
                     sector_cache_clear, sector_cache_stats, set_exchange_default, comm_cache_stats, comm_cache_clear, live_handles, host_register, host_unregister)
from .hamiltonian import EDContext  # noqa: F401
from . import models  # noqa: F401
from .distributed import (HaloHxv, ShardedHxv, ShardedLanczos, TransposedHxv, dw_split, exchange_ingest_bytes, halo_plan,  # noqa: F401
                          sharded_eigh_lowest, start_vector_slab)

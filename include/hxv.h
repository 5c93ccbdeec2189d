/*
 * hxv.h -- C-ABI of the MI355X-native sector Hamiltonian x vector engine.
 *
 * This is the drop-in boundary for ONE path of QcmPlab/CDMFT-LANC-ED: the procedure
 * pointer  spHtimesV_p  (abstract interface cc_sparse_HxV, ED_VARS_GLOBAL.f90:72-78,146)
 * and the routines that open/close the sector it acts on (ED_HAMILTONIAN.f90:39-221).
 * Plain pointers and sizes only; no C++/torch types.  Every entry returns 0 on success
 * and a non-zero hxv_status otherwise (the reference's convention is `stop "..."`,
 * e.g. ED_HAMILTONIAN_SPARSE_HxV.f90:57,247; the Fortran glue turns non-zero into stop).
 *
 * Vector layout (identical to the reference): element (iup,idw), 0-based, lives at
 *   i = iup + idw*DimUp            (ED_HAMILTONIAN/sparse/H_local.f90:2-3)
 * as complex(8) = two doubles (re,im).  A rank of an nranks-way split owns the contiguous
 * columns idw in [dw0, dw0+qdw), qdw = DimDw/nranks (+1 for rank < mod(DimDw,nranks))
 * (ED_HAMILTONIAN.f90:93-105), i.e. qdw*DimUp contiguous elements = vecDim_Hv_sector.
 */
#ifndef HXV_H
#define HXV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hxv_handle hxv_handle;

enum hxv_status {
  HXV_OK = 0,
  HXV_ERR_ARG = 1,        /* bad argument / inconsistent sizes                          */
  HXV_ERR_HIP = 2,        /* a HIP runtime call failed (no device, OOM, launch failure) */
  HXV_ERR_STATE = 3,      /* handle not in the right state                              */
  HXV_ERR_UNSUPPORTED = 4 /* valid input the engine does not implement                  */
};

/* Operator inputs = the reference's module globals visible at bind time.
 * Complex arrays are interleaved (re,im) doubles in the reference's Fortran array order, so
 * the Fortran glue passes c_loc() of the reference arrays themselves.                      */
typedef struct {
  int32_t nlat, norb, nspin, nbath; /* ED_INPUT_VARS.f90:13-16                            */
  int32_t hfmode;                   /* ED_INPUT_VARS.f90:164 (logical -> 0/1)             */
  int32_t reserved;
  double uloc[5];                   /* ED_INPUT_VARS.f90:129                              */
  double ust, jh, jx, jp, xmu;      /* ED_INPUT_VARS.f90:130-135                          */
  const double *imphloc; /* complex (Nlat,Nlat,Nspin,Nspin,Norb,Norb)        ED_VARS_GLOBAL.f90:119 */
  const double *hbath;   /* complex (Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath)  = Hbath_build(lambda_ib),
                            ED_HAMILTONIAN_SPARSE_HxV.f90:65; may be NULL if nbath==0      */
  const double *vbath;   /* real (Nlat,Nspin,Norb,Nbath) = diag_hybr, ED_HAMILTONIAN_SPARSE_HxV.f90:70 */
} hxv_model;

/* ---- open / close a sector -------------------------------------------------------------
 * Replaces build_Hv_sector(isector) (ED_HAMILTONIAN.f90:39-143) + ed_buildh_main
 * (ED_HAMILTONIAN_SPARSE_HxV.f90:40-110): builds the sector maps, the one-spin hopping
 * tables and the separable diagonal on the host, uploads them to HIP device `device`.
 * (nup,ndw) = get_Nup/get_Ndw(isector) (ED_SETUP.f90:477-500). rank/nranks = the DimDw
 * split of ED_HAMILTONIAN.f90:93-105 (nranks=1: serial, MpiStatus=F).  With Norb>1 and Jx or
 * Jp != 0 (Jhflag, ED_SETUP.f90:200-201) the spin-exchange / pair-hopping block spH0nd
 * (sparse/H_non_local.f90:4-100) is applied on the fly, inside the product's second pass (option "fold_nd" = 0: as a third
 * pass over Hv).                                                                         */
int hxv_create_from_model(const hxv_model *model, int32_t nup, int32_t ndw, int32_t rank, int32_t nranks, int32_t device,
                          hxv_handle **out);

/* Same, from the reference's own stored matrices (spH0ups(1), spH0dws(1), spH0d of
 * ED_VARS_GLOBAL.f90:142-144) flattened to CSR: rowptr[n+1] (0-based offsets), cols (1-based,
 * as stored by sp_insert_element ED_SPARSE_MATRIX.f90:254-322), vals interleaved complex.
 * diag = the local rows of spH0d (qdw*DimUp complex values, one per row).  A split sector (nranks > 1) takes the exchange chosen
 * by hxv_set_exchange_default like a sector opened from a model.                         */
int hxv_create_from_csr(int32_t dimup, int32_t dimdw, const int64_t *up_rowptr, const int32_t *up_cols, const double *up_vals,
                        const int64_t *dw_rowptr, const int32_t *dw_cols, const double *dw_vals, const double *diag,
                        int32_t rank, int32_t nranks, int32_t device, hxv_handle **out);

/* spH0nd of a from_csr handle (ED_VARS_GLOBAL.f90:145; built by sparse/H_non_local.f90:23-98 when Jhflag, added to Hv at
 * ED_HAMILTONIAN_SPARSE_HxV.f90:217-225 serial / :298-312 MPI): the LOCAL rows (vecDim of them) with GLOBAL 1-based column indices
 * i = iup + (idw-1)*DimUp, rowptr[vecDim+1] 0-based offsets, vals interleaved complex.  Applied as its own pass over Hv after the
 * product, from the gathered vector (all-gather exchange only).  Once per handle; not on handles opened from a model with
 * Jx / Jp (those build the block themselves).  Dim < 2^31 (the reference's column type).                                  */
int hxv_set_nonlocal_csr(hxv_handle *h, const int64_t *rowptr, const int32_t *cols, const double *vals);

/* delete_Hv_sector (ED_HAMILTONIAN.f90:149-190). NULL is a no-op. */
int hxv_destroy(hxv_handle *h);

/* vecDim_Hv_sector (ED_HAMILTONIAN.f90:197-221): local vector length DimUp*qdw. */
int64_t hxv_vecdim(const hxv_handle *h);
/* DimUp, DimDw, Dim, mpiQdw, mpiIshift (ED_HAMILTONIAN.f90:58-60,93-105); any out may be NULL */
int hxv_dims(const hxv_handle *h, int32_t *dimup, int32_t *dimdw, int64_t *dim, int32_t *qdw, int64_t *ishift);

/* ---- the product -----------------------------------------------------------------------
 * spHtimesV_p(Nloc,v,Hv) with HOST arrays (cc_sparse_HxV, ED_VARS_GLOBAL.f90:72-78): v, Hv = this rank's slab of
 * vecDim_Hv_sector elements; Hv is overwritten (ED_HAMILTONIAN_SPARSE_HxV.f90:175,250); synchronous.  Serial sector
 * (nranks==1) = spMatVec_main; split sector (nranks>1) = spMatVec_MPI_main: the re-assembly the reference does with two
 * MPI transposes inside the product (ED_HAMILTONIAN_SPARSE_HxV.f90:272-296) is done here by the engine itself -- one
 * ncclAllGather of the slabs over xGMI -- once the handles share a communicator (hxv_comm_init below).
 * Pays two PCIe copies of the slab per call.                                                            */
int hxv_apply_host(hxv_handle *h, int64_t nloc, const void *v, void *hv);
/* The two PCIe copies ARE this call's cost (C3: 2 x 2.65 GB; the kernels between them take 4 ms): bench.py reports it as config.apply_host next
 * to the floor the box's own H2D / D2H rates give.  Host arrays that are page-locked move by DMA at the link's rate; a host program that keeps
 * passing the same work arrays (ED_DIAG's Lanczos vectors) registers them ONCE (hipHostRegister, ~0.1 s per GB) and unregisters them before it
 * frees them.  Registering an array that is registered already is not an error.                                                          */
int hxv_host_register(void *ptr, int64_t bytes);
int hxv_host_unregister(void *ptr);

/* ---- slab exchange of a split sector (MpiStatus=T side of the boundary) -------------------------------------------
 * One RCCL communicator over the nranks handles of a sector (one process per GPU).  Rank 0 draws an id, the host program
 * broadcasts its HXV_COMM_ID_BYTES bytes by its own means (the reference: MPI_Bcast over MpiComm), every rank calls
 * hxv_comm_init (collective).  Afterwards hxv_apply_host, hxv_apply_device_slab and the device Lanczos drivers work on
 * split sectors: the product all-gathers the slabs (equal counts, all-gather layout of hxv_apply_device) on the stream it
 * runs on, the drivers' dot products are ncclAllReduce sums.  hxv_destroy unbinds the handle from the communicator (below).
 * RCCL is loaded at the first of these calls (dlopen), not at link time.                                            */
#define HXV_COMM_ID_BYTES 128
int hxv_comm_unique_id(void *id128);
int hxv_comm_init(hxv_handle *h, const void *id128);
int hxv_comm_free(hxv_handle *h);
/* ONE communicator per process, not per sector.  The reference sets MpiComm once per solve (ED_VARS_GLOBAL.f90:365-380) and derives a
 * sub-communicator only for sectors with DimDw < MpiSize (ED_HAMILTONIAN.f90:63-89), while its callers open 289 sectors per ED_DIAG sweep
 * (ED_DIAG.f90:142-190) and 56 per Green's-function stage (ED_GF_NORMAL.f90:208-222).  hxv_comm_init therefore builds a communicator
 * (ncclCommInitRank, a collective that costs far more than opening a sector) only the FIRST time it sees a (RCCL library, nranks, rank, device);
 * every later call with the same key binds the handle to that communicator and does not consume the id it is given (the host may keep
 * drawing and broadcasting ids -- 128 bytes -- or pass the first one again; every rank takes the same branch because every rank has made the
 * same calls).  A sector with DimDw < nranks is opened by the first DimDw ranks with nranks' = DimDw: another key, built once as well.
 * hxv_destroy / hxv_comm_free unbind; the communicators live until hxv_comm_cache_clear (every rank, at the same point of the program: the
 * end of a solve) or the end of the process.  A communicator that was aborted (hxv_comm_abort) or failed a collective leaves the cache: the
 * next hxv_comm_init builds a new one from its id.  Environment HXV_COMM_CACHE=0: one communicator per handle, destroyed with it.
 * hxv_comm_cache_stats: communicators cached now, ncclCommInitRank calls made so far, hxv_comm_init calls served without one (any out NULL). */
int hxv_comm_cache_stats(int64_t *entries, int64_t *inits, int64_t *reuses);
int hxv_comm_cache_clear(int64_t *destroyed);
/* A rank has failed OUTSIDE the library (its host thread / process raised before its next collective): wake this handle's rank instead of
 * leaving it inside a collective waiting for the lost peer.  RCCL communicators: ncclCommAbort -- callable from another host thread while
 * the handle's own thread blocks in a collective; that collective returns an error, the communicator is gone (later calls report a missing
 * communicator; hxv_comm_free / hxv_destroy still clean up).  Thread-rank groups: same as hxv_comm_local_abort.  HXV_ERR_UNSUPPORTED when
 * the RCCL library in use exports no ncclCommAbort.                                                                                */
int hxv_comm_abort(hxv_handle *h);
/* The file the RCCL entry points of this handle's communicator were resolved from (dladdr; "" without a communicator): which librccl a
 * multi-GPU run actually used, next to the one torch.distributed loaded.                                                          */
const char *hxv_comm_library(const hxv_handle *h);
/* STATUS of N>1 through RCCL: no round of this project has had more than one GPU, and RCCL refuses two ranks on one device, so
 * librccl itself has only ever run with ONE rank.  Everything around it -- slab copies, uneven splits, halo lists and
 * offsets, the drivers' all-reduces, the collective error agreement -- is executed with several ranks by the second
 * transport below, which shares that code; the RCCL call sites themselves (counts, offsets and pointers of the grouped
 * ncclSend / ncclRecv, the in-place ncclAllGather, the ncclMax agreement) run with 2-4 ranks against a TEST double of RCCL's ten entry
 * points (tests/rccl_double: one for thread ranks of one process, one for separate processes on one GPU): the environment variable HXV_RCCL_LIB, read by hxv_comm_unique_id /
 * hxv_comm_init, names the library to load instead of the system's librccl (a communicator keeps the library it was made with).
 * THREAD RANKS: the nranks handles of a sector live in ONE process, one host thread per rank (same GPU or different GPUs of
 * the node); slabs travel by device-to-device copies ordered with HIP events, scalars through host memory.  Create the group
 * once, then every rank's thread calls hxv_comm_init_local (collective: it returns when all nranks have joined); from then on
 * hxv_apply_host / hxv_apply_device_slab / the device drivers behave exactly as with hxv_comm_init, each called from its
 * rank's thread.  hxv_comm_free / hxv_destroy leave the group; destroy it afterwards.
 * A rank that drops out must not leave its peers waiting: a collective of the group gives up with HXV_ERR_STATE when a peer has not
 * arrived within HXV_LOCAL_TIMEOUT_S seconds (environment, default 300; read by hxv_comm_local_create), or at once after
 * hxv_comm_local_abort(group) -- what the host program calls when one rank's thread fails outside the library.  A broken group stays
 * broken: free the handles, destroy it, create a new one.  A HIP error inside a collective is reported by the rank that had it
 * AFTER the collective's last barrier (its peers complete the step).                                                              */
int hxv_comm_local_create(int32_t nranks, void **group);
int hxv_comm_init_local(hxv_handle *h, void *group);
int hxv_comm_local_abort(void *group);
int hxv_comm_local_destroy(void *group);
/* d_hv_local = (H v)|slab from this rank's slab d_v_local (hxv_localvec_elems() elements each, padded device layout):
 * exchange + product, asynchronous on `stream`.  nranks==1 without a communicator: the plain product.              */
int hxv_apply_device_slab(hxv_handle *h, const void *d_v_local, void *d_hv_local, void *stream);
/* Measurement: nrep such products on the handle's own stream, timed with HIP events (collective on a split sector: every rank calls
 * it with the same nrep).  *ms_step = mean time of a whole product on this rank (exchange included), *ms_kernels = mean time of its
 * product kernels alone (both kernel regions in exchange mode 2; with option "exchange_overlap" also pass A's region on the second stream,
 * which runs BESIDE the exchange: hxv_get_option(h, "time_kernels_overlapped_us") tells that part, so the exchange's own share of a step is
 * *ms_step - (*ms_kernels - overlapped), never negative).                                                                        */
int hxv_time_apply_slab(hxv_handle *h, const void *d_v_local, void *d_hv_local, int32_t nrep, float *ms_step, float *ms_kernels);
/* Where the exchange wants this rank's slab (its slot of the gather buffer): a caller that builds its vector there and hands
 * THAT pointer to hxv_apply_device_slab saves the slab copy of every product.  [qdw columns][pitch] complex elements; allocated on
 * first use; valid until hxv_comm_free / hxv_destroy; the device Lanczos drivers of a split sector keep their own vectors at the same
 * place, so a vector left there does not survive a driver call -- a START vector of hxv_lanczos_tridiag[_pair] built there is fine: the
 * drivers stage it (one slab copy per run) before they clear the place.  The reference allocates the gathered vector per call and
 * copies (ED_HAMILTONIAN_SPARSE_HxV.f90:277-296).                                                                              */
int hxv_slab_home(hxv_handle *h, void **d_slab);
/* Exchange mode 2 (hxv_set_exchange_default(2) / HXV_EXCHANGE=alltoall before the sector is opened; not with the spH0nd block): the
 * reference's own two transposes (vector_transpose_MPI, ED_HAMILTONIAN_COMMON.f90:30-94; spMatVec_mpi_main :272-296) inside the
 * engine -- slab -> row panels (rows split like the columns, :274-275), dw hops on the panel, back, then diagonal + up hops on the slab.
 * Moves (P-1)/P of one slab per rank and transpose instead of P-1 slabs: the lowest-traffic exchange.  hxv_apply_host,
 * hxv_apply_device_slab and the device drivers (REAL-vector mode included) use it transparently; hxv_slab_home does not apply.   */
int64_t hxv_exchange_count(const hxv_handle *h); /* exchanges since creation */
/* HALO exchange (the lower-traffic alternative, replaces the transposes of ED_HAMILTONIAN_COMMON.f90:30-94 differently): with
 * the reference's own DimDw split a rank's rows of H_dw reference only a subset of the other ranks' columns (C3, 8 ranks:
 * 4.2 slabs instead of the 7 an all-gather moves).  A handle created in halo mode (hxv_set_exchange_default(1) or
 * HXV_EXCHANGE=halo before the create call; with Jx / Jp the partner columns of the spH0nd block's dw moves join the list) expects d_v_full of hxv_apply_device in the HALO
 * LAYOUT: its own qdw columns first, then the columns listed by hxv_halo_lists (ascending = grouped by owner rank);
 * hxv_fullvec_elems() reports the length.  hxv_apply_host / hxv_apply_device_slab / the drivers then exchange exactly
 * those columns (grouped ncclSend/ncclRecv).  recv_counts/send_counts: columns per peer rank; recv_cols: global column
 * indices in slot order; send_cols: LOCAL column indices grouped by destination rank.                                */
int hxv_set_exchange_default(int32_t mode); /* 0 all-gather [default], 1 halo, 2 two all-to-all transposes; applies to handles created afterwards */
int32_t hxv_exchange_mode(const hxv_handle *h);
int hxv_halo_counts(const hxv_handle *h, int32_t *recv_counts, int32_t *send_counts);
int hxv_halo_lists(const hxv_handle *h, int32_t *recv_cols, int32_t *send_cols);
/* The halo PLAN of any rank of any split, computed from H_dw alone (CSR as hxv_get_csr returns it / as spH0dws(1) stores it:
 * 1-based columns) -- no handle, no device, no communicator: counts[nranks] columns per peer; recv_cols (slot order) and send_cols
 * (grouped by destination) as GLOBAL 0-based column indices, *n_recv / *n_send their lengths (call once with NULL lists to size
 * them).  What one process needs to check that rank p's send list towards q IS rank q's receive list from p, for every pair.
 * (The hopping part only: a sector with Jx / Jp adds the spH0nd partners -- hxv_halo_lists of its handle is the full list.)      */
int hxv_halo_plan_from_csr(int32_t dimdw, const int64_t *dw_rowptr, const int32_t *dw_cols, int32_t rank, int32_t nranks,
                           int32_t *recv_counts, int32_t *send_counts, int32_t *recv_cols, int32_t *send_cols, int32_t *n_recv,
                           int32_t *n_send);

/* Device-resident product.  d_v_full: the FULL vector in the ALL-GATHER LAYOUT: nranks slabs of
 * cmax*pitch elements each, cmax = ceil(DimDw/nranks), slab r holding rank r's columns (ranks
 * that own one column less leave their last pitch elements unused) -- exactly what an
 * equal-count ncclAllGather / MPI_Allgather of the padded slabs produces; hxv_fullvec_elems()
 * elements in all.  d_hv_local: this rank's slab (hxv_localvec_elems() elements), overwritten.  Asynchronous on `stream` (a hipStream_t; NULL = the legacy
 * default stream, as in every HIP API), so it orders with the caller's other work on that
 * stream.  d_v_full and d_hv_local must not overlap.
 * When H is real (hxv_real_vectors_available) one complex product is two independent real products, H(x + i y) =
 * Hx + i Hy: a caller with several real vectors to multiply (the Green's-function channels of one solve) gets two
 * per call at 4.4 ms (C3) instead of 2.9 ms each through hxv_apply_device_real.                                     */
int hxv_apply_device(hxv_handle *h, const void *d_v_full, void *d_hv_local, void *stream);
/* DEVICE VECTOR LAYOUT.  On the device every column of DimUp elements is padded to hxv_pitch(h) =
 * DimUp rounded up to a multiple of 8 elements, so each column starts on a 128-byte line; element
 * (iup, column slot k) lives at k*pitch + iup.  The pad rows are never read as sources and never written by
 * the product; the Lanczos entries require them to be ZERO in the vectors they are given (dot products
 * run over the padded arrays).  hxv_apply_host converts from/to the reference's contiguous host layout.
 *   hxv_fullvec_elems : length of d_v_full  = nranks*cmax*pitch  (all-gather layout above)
 *   hxv_localvec_elems: length of d_hv_local and of every Lanczos vector = qdw*pitch
 * DEVICE ROW ORDER (round 6).  WITHIN a column the rows of a device vector need not follow the reference's basis order: a sector opened from a
 * model may store the up configurations in the order they take when the orbitals are renumbered (orbital o at bit pos[o]) so that the
 * out-of-block gathers of the product's up-hop pass run over long contiguous stretches (C3: a third fewer cache lines).
 * Reordering creation operators changes the sign of a basis vector, hence
 *        d_vec[k*pitch + perm[iup]] = sign[iup] * v_ref[k*DimUp + iup]              (perm, sign: hxv_row_order)
 * Every entry point that takes or returns HOST arrays (hxv_apply_host, the *_host drivers, hxv_vector_from_host / _to_host, evecs_host) and
 * every introspection call (hxv_get_maps / _csr / _diag) speaks the reference's order: the conversion happens at the boundary, on the device.
 * Device vectors are self-consistent among the device entry points (product, Lanczos drivers, ladder operators between sectors); a caller
 * that BUILDS or READS a device vector element by element uses hxv_row_order.  Columns (the dw index, the one that is split over ranks)
 * always keep the reference's order.  hxv_row_order returns 1 when the sector has a device row order (perm[DimUp]: reference row -> device
 * row; sign[DimUp]: +1 / -1 by reference row; either may be NULL), 0 when rows are in the reference's order (identity written), -1 on a NULL
 * handle.  Sectors opened from stored matrices, row panels and small sectors (DimUp < 2048) keep the reference's order; environment
 * HXV_ROW_ORDER=0 keeps it everywhere.                                                                                                 */
int32_t hxv_row_order(const hxv_handle *h, int32_t *perm, int8_t *sign);
int64_t hxv_fullvec_elems(const hxv_handle *h);
int64_t hxv_localvec_elems(const hxv_handle *h);
int32_t hxv_pitch(const hxv_handle *h);

/* ---- the two halves of the product, for the reference's own all-to-all exchange (ED_HAMILTONIAN_SPARSE_HxV.f90:272-296)
 * instead of the all-gather: rank r first receives the row panel X = v[rows U_r, ALL columns] (rows split like the
 * columns, mpiQup :274-275), computes Y = X H_dw^T with a PANEL handle, sends Y's column ranges back to their owners, and
 * finishes with  hv = D.v + H_up v + (assembled dw part)  on its own slab.  Each transpose moves (P-1)/P of a slab
 * instead of (P-1) slabs.
 * hxv_create_dw_panel : handle for a panel of `nrows` rows (dw-only: no basis/H_up/diagonal for the up index);
 *                       vectors are [DimDw columns][pitch = roundup8(nrows)].
 * hxv_apply_dw_panel  : d_y = d_x H_dw^T, same layout in and out.
 * hxv_apply_up_add    : d_hv_local = D.v + H_up v + d_w on the local slab; d_v_local and d_w are [qdw columns][pitch]
 *                       (d_v_local = this rank's slab only -- no gathered vector); not available with spH0nd (Jx/Jp). */
int hxv_create_dw_panel(const hxv_model *model, int32_t nup, int32_t ndw, int32_t nrows, int32_t device, hxv_handle **out);
int hxv_apply_dw_panel(hxv_handle *panel, const void *d_x, void *d_y, void *stream);
int hxv_apply_up_add(hxv_handle *h, const void *d_v_local, const void *d_w, void *d_hv_local, void *stream);

/* ---- REAL-vector mode.  When every amplitude of H is real (cdn_hm_* models; not BHZ), H maps real vectors to real
 * vectors, and a Lanczos run started from a real vector never leaves the reals: the same kernels then run on
 * double instead of complex(8) elements -- half the bytes of every pass.  The reference keeps complex arrays
 * throughout (cc_sparse_HxV, ED_VARS_GLOBAL.f90:72-78), so this mode exists only on the device side: the device
 * Lanczos drivers below select it by themselves when it applies (option "real_vectors", default 1) and convert at
 * their boundaries; hxv_apply_device_real is the product itself.
 *   layout: double[DimDw columns][hxv_pitch_real(h) = roundup16(DimUp)], pad rows zero / ignored like above.
 *   available iff H is real, the tiled kernels are in use, no spH0nd block (hxv_real_vectors_available).  On a split sector
 *   the drivers exchange REAL slabs (half the bytes on the links); hxv_apply_device_real itself takes the whole vector of an
 *   unsplit sector.                                                                                                  */
int32_t hxv_real_vectors_available(const hxv_handle *h);
int32_t hxv_pitch_real(const hxv_handle *h);
int64_t hxv_realvec_elems(const hxv_handle *h);
int hxv_apply_device_real(hxv_handle *h, const void *d_v_real, void *d_hv_real, void *stream);

/* Time `nrep` back-to-back device products with HIP events recorded on the stream the
 * kernels are launched on; returns the mean milliseconds per product.                    */
int hxv_time_apply(hxv_handle *h, const void *d_v_full, void *d_hv_local, int32_t nrep, float *ms_per_apply);

/* ---- Lanczos on device (SciFortran sp_lanc_tridiag / sp_lanc_eigh call shapes,
 * ED_GF_NORMAL.f90:215-220, ED_DIAG.f90:176-184).  Split sectors (nranks>1) after hxv_comm_init:
 * vectors are this rank's slab, dot products are all-reduced; the fused recurrence and the REAL-vector mode apply there too
 * (the epilogue's partial sums are a rank's share of alpha).  A rank that fails in its local preparations tells the others before
 * the first collective: all ranks return an error together instead of waiting for each other.
 * All vectors in the padded device layout (hxv_localvec_elems() elements, pad rows zero).
 * tridiag: d_vin = start vector (normalised by the driver if it is not, as SciFortran does); alanc[nlanc], blanc[nlanc]
 *   filled as alanc(k)=<q_k|H|q_k>, blanc(k+1)=beta_{k+1}, blanc(1)=0 (ED_GF_NORMAL.f90:949-951);
 *   *nsteps = iterations done (early exit when beta < threshold).
 * eigh: lowest eigenvalue *egs and eigenvector d_vect (device, Dim, written) from a
 *   deterministic start vector; stops when |dE| < threshold (and, if d_vect is wanted, the Ritz
 *   residual estimate |beta*y_last| < 1e-11*max(1,|E|)) or at nitermax (ED_DIAG.f90:176).
 * tridiag with nlanc >= 8 runs iterations 1.. on the device alone (scalars in device memory, three iterations per
 *   hipGraph, alanc/blanc copied back once; option "lanczos_graph", default 1): bit-identical to the stepwise run.
 * REAL-vector mode (above): with option "real_vectors" = 1 (default) these drivers and hxv_eigh_lowest run on real
 *   vectors whenever hxv_real_vectors_available(h) and the start vector is real (eigh / eigh_lowest: the engine's own
 *   start vector is then real; tridiag: d_vin must have exactly zero imaginary part, else the complex path runs).
 *   Inputs and outputs keep the complex layout; alanc/blanc/E are the same numbers.  hxv_get_option(h,"lanczos_real_last")
 *   tells which path the last run took.                                                               */
int hxv_lanczos_tridiag(hxv_handle *h, const void *d_vin, int32_t nlanc, double *alanc, double *blanc, double threshold,
                        int32_t *nsteps);
int hxv_lanczos_eigh(hxv_handle *h, int32_t nitermax, double threshold, double *egs, void *d_vect, int32_t *niter);
/* Same two drivers with HOST vectors in the reference's contiguous layout (what ED_DIAG / ED_GF_NORMAL hold):
 * one PCIe copy per Lanczos RUN instead of two per iteration.  vin_host: Dim complex, normalised by the caller;
 * vect_host: Dim complex, written (may be NULL).  These are what the Fortran glue's gpu_sp_lanc_tridiag /
 * gpu_sp_lanc_eigh (SciFortran call signatures) forward to.                                                 */
int hxv_lanczos_tridiag_host(hxv_handle *h, const void *vin_host, int32_t nlanc, double *alanc, double *blanc, double threshold,
                             int32_t *nsteps);
int hxv_lanczos_eigh_host(hxv_handle *h, int32_t nitermax, double threshold, double *egs, void *vect_host, int32_t *niter);
/* TWO tridiagonalisations on one product, for real H (hxv_real_vectors_available): the Green's-function channels of one solve
 * are independent runs of sp_lanc_tridiag (ED_GF_NORMAL.f90:123-306, one per call site :215,282,422,504,638,720,801,882), and
 * a complex product IS two real ones, H(x + i y) = H x + i H y.  The two REAL start vectors (complex layout, imaginary parts
 * exactly zero -- c / c^dagger applied to a real ground state; refused otherwise) travel as real and imaginary part of one
 * complex Lanczos vector; every scalar of the recurrence, every dot product and both early exits exist once per component.
 * alanc_x/blanc_x[nlanc], *nsteps_x as in hxv_lanczos_tridiag.  Each component's numbers are bit-identical to
 * hxv_lanczos_tridiag on that start vector through the same kernels, i.e. with option real_vectors = 0 (the complex-vector path; any job_up).
 * Against hxv_lanczos_tridiag's DEFAULT path (real_vectors = 1: the real-vector kernels, another summation order) they agree to rounding, not bit
 * for bit -- 1e-10 over the first steps, growing with the step like any two Lanczos runs; the continued fraction the consumer builds from them
 * (ED_GF_NORMAL.f90:915-975) agrees to 1e-9 (tests/test_gpu_lanczos.py::test_paired_tridiagonalisation_..., tests/test_gpu_solve_sweep.py).
 * Split sectors: slabs per rank, every sum all-reduced (any exchange).  The _host form takes the start vectors in the reference's
 * contiguous host layout.
 * What does NOT pair: the (c^+_i + xi c^+_j)|gs>, (c_i - xi c_j)|gs> channels of the reference's default chan4 form (ED_GF_NORMAL.f90:746-780,
 * :827-861) have COMPLEX start vectors even at real H -- 24 of the 56 runs of a 2x2 solve; they go through hxv_lanczos_tridiag on complex
 * vectors (C3: 5.5 ms per step against 2.8 ms per paired real channel-step).  For real symmetric H they carry nothing the real channels do
 * not; the reference's own ed_gf_symmetric (chan2) drops them: INTEGRATION.md section 4.                                            */
int hxv_lanczos_tridiag_pair(hxv_handle *h, const void *d_vin_a, const void *d_vin_b, int32_t nlanc, double *alanc_a, double *blanc_a,
                             double *alanc_b, double *blanc_b, double threshold, int32_t *nsteps_a, int32_t *nsteps_b);
int hxv_lanczos_tridiag_pair_host(hxv_handle *h, const void *vin_a_host, const void *vin_b_host, int32_t nlanc, double *alanc_a,
                                  double *blanc_a, double *alanc_b, double *blanc_b, double threshold, int32_t *nsteps_a,
                                  int32_t *nsteps_b);
/* ---- Several lowest eigenpairs on device: the call SciFortran's sp_eigh (P-ARPACK) serves at ED_DIAG.f90:152-160,
 *   call sp_eigh(spHtimesV_p, eig_values(Neigen), eig_basis(vecDim,Neigen), Nblock, Nitermax, tol=lanc_tolerance)
 * as a thick-restart Lanczos (the explicit-restart form of ARPACK's implicitly restarted Lanczos for Hermitian
 * operators) with a Krylov basis of ncv (= Nblock) vectors resident in HBM, full re-orthogonalisation and ARPACK's
 * convergence test |beta*s_mi| <= tol*max(eps^(2/3),|theta_i|)  (tol below machine epsilon -- the reference's default
 * lanc_tolerance=1e-18 -- is raised to epsilon).  Serial and split sectors (below).
 *   neigen       : number of lowest eigenpairs wanted (Neigen)
 *   ncv          : basis size (Nblock = lanc_ncv_factor*Neigen+lanc_ncv_add, ED_DIAG.f90:96); <=0 -> 10*neigen; max 64
 *   maxrestart   : restart limit (Nitermax)
 *   evals        : [neigen] ascending
 *   d_evecs      : device, neigen consecutive vectors of hxv_localvec_elems() elements (padded layout), or NULL
 *   evecs_host   : host, eig_basis(Dim,neigen) in the reference's contiguous layout, or NULL
 *   *nconv       : how many of the neigen pairs met the test; *nmatvec: H x V products spent.
 * Needs (ncv+1) vectors of HBM; fails with HXV_ERR_HIP and a message naming the shortfall otherwise.
 * A single Krylov space sees ONE vector of an exactly degenerate level -- ARPACK included: sp_eigh returns further copies only when
 * rounding happens to bring them up, while ED_DIAG.f90:234-244 keeps every state within gs_threshold of the minimum.  DEFAULT = what
 * ARPACK does: one Krylov space, no extra work.  Option "eigh_degenerate" = 1 ASKS for the copies: once the wanted pairs have converged
 * they are locked and the same iteration in their orthogonal complement looks for a state below the current neigen-th lowest value (a
 * hidden copy), repeatedly -- a round that finds nothing stops once the residual bound of its lowest Ritz value clears the level (C3: 380 products + 60; C2: up to as many again); every step of such a round
 * removes the locked eigenvectors (they are converged to `tol` only).  "Nothing below" is a HEURISTIC answer: the lowest Ritz value of
 * the complement minus its residual must clear the level from the second restart cycle on, with a falling residual -- a copy whose
 * overlap with the start vector is at rounding level can still be missed.  *nmatvec counts both; hxv_get_option(h,
 * "eigh_last_search_products" | "eigh_last_check_products") splits it.  Clusters with a non-abelian point group (the 2x2 plaquette: D4)
 * DO have degenerate levels inside a sector: INTEGRATION.md section 3 says when to ask.
 * Split sectors: after hxv_comm_init (vectors = slabs).  If Dim <= ncv the Krylov space closes and all returned pairs are exact. */
int hxv_eigh_lowest(hxv_handle *h, int32_t neigen, int32_t ncv, int32_t maxrestart, double tol, double *evals, void *d_evecs,
                    int32_t *nconv, int32_t *nmatvec);
int hxv_eigh_lowest_host(hxv_handle *h, int32_t neigen, int32_t ncv, int32_t maxrestart, double tol, double *evals,
                         void *evecs_host, int32_t *nconv, int32_t *nmatvec);
/* Time nrep full Lanczos iterations (HxV + recurrence + 2 reductions) on device. */
int hxv_time_lanczos(hxv_handle *h, void *d_work3 /* 3*hxv_localvec_elems() complex */, int32_t nrep, float *ms_per_iter);

/* ---- Green's-function start vectors (ED_GF_NORMAL.f90:174-214: the reference applies c / c^dagger to the
 * ground state serially on the master and scatters): d_out = c^(dagger)_{orbital,spin} d_psi, from the sector
 * open in `from` (vector of its Dim) into the sector open in `to` (N_spin +- 1; vector of its Dim), sign =
 * (-1)^(# occupied orbitals of the SAME spin below `orbital`) (c/cdg, ED_SETUP.f90:807-833; no cross-spin sign,
 * as in the reference).  orbital is 0-based (= pos-1), spin 0 = up, 1 = dw, create 1 = c^dagger, 0 = c.
 * *norm2 = <out|out> (the reference normalises by it, ED_GF_NORMAL.f90:197-199).  Both handles from_model, same device; padded
 * device layouts, d_out's pad rows are written as zero.
 * SPLIT sectors (both handles with the same rank / nranks, `to` with its communicator): d_psi / d_out are this rank's slabs and
 * every rank builds its own slab of the new vector -- the master-only loop + scatter of ED_GF_NORMAL.f90:174-214 is gone.
 * A spin-up operator is local to a slab (both sectors share DimDw and its split); a spin-dw operator maps whole columns of
 * one split onto the other: every rank derives from the two dw maps what it needs from whom and what the others need from it,
 * and one column exchange moves them.  *norm2 is the GLOBAL <out|out>.                                                        */
int hxv_apply_ladder(hxv_handle *from, hxv_handle *to, int32_t orbital, int32_t spin, int32_t create, const void *d_psi,
                     void *d_out, double *norm2);
/* Mixed channels (ED_GF_NORMAL.f90:370-406 (c^dagger_i + c^dagger_j)|gs>, :746-780 (c^dagger_i + xi c^dagger_j)|gs>, and the
 * c_i + c_j / c_i - xi c_j counterparts): d_out = (accumulate ? d_out : 0) + (coef_re + i coef_im) * c^(dagger) d_psi;
 * *norm2 = <out|out> after the update.  With accumulate != 0 the caller's d_out must have zero pad rows.     */
int hxv_apply_ladder_axpy(hxv_handle *from, hxv_handle *to, int32_t orbital, int32_t spin, int32_t create, double coef_re,
                          double coef_im, int32_t accumulate, const void *d_psi, void *d_out, double *norm2);

/* ---- device vectors owned by the library --------------------------------------------------------------------------------
 * For host programs without a HIP binding of their own (the Fortran glue): a local vector of the handle's sector in the padded device
 * layout (hxv_localvec_elems() complex elements, zeroed), from the engine's buffer cache.  Such a pointer is what the device drivers take
 * and return (hxv_lanczos_eigh's d_vect, hxv_apply_ladder's d_psi / d_out, hxv_lanczos_tridiag's d_vin), so a Green's-function channel --
 * ground state, c^dagger|gs>, tridiagonalisation (ED_GF_NORMAL.f90:174-217) -- runs without a Dim-sized PCIe transfer; the two copies
 * convert to / from the reference's contiguous host layout when a vector is wanted on the host after all.  hxv_vector_free takes the handle the
 * vector was allocated on, BEFORE that handle is destroyed; hxv_destroy returns whatever was not freed.                              */
int hxv_vector_alloc(hxv_handle *h, void **d_vec);
int hxv_vector_alloc_many(hxv_handle *h, int32_t count, void **d_vec); /* `count` vectors in one allocation, hxv_localvec_elems() apart: the d_evecs of hxv_eigh_lowest; freed as one */
int hxv_vector_free(hxv_handle *h, void *d_vec);
int hxv_vector_from_host(hxv_handle *h, const void *v_host, void *d_vec);
int hxv_vector_to_host(hxv_handle *h, const void *d_vec, void *v_host);

/* ---- device-buffer cache.  A fresh hipMalloc costs ~25 ms per GB on this platform (0.7 s for the 28 GB Krylov basis of
 * hxv_eigh_lowest at Ns=16), and an ED run opens sectors one after another, so the vector-sized buffers a handle frees
 * (dw-hop scratch, Lanczos vectors, staging, Krylov basis) are kept per device and reused by the next handle.
 * Environment: HXV_POOL=0 disables it, HXV_POOL_MAX_GB caps the cached bytes (default 40 % of the device memory).
 * hxv_pool_trim returns everything cached on `device` (all devices if < 0) to the driver.                        */
int hxv_pool_trim(int32_t device);
int hxv_pool_stats(int32_t device, int64_t *cached_bytes, int64_t *hits, int64_t *misses);

/* ---- sector-image cache.  The reference opens and closes a sector around EVERY Lanczos run: once per sector in ED_DIAG.f90:142-186 and
 * once per Green's-function channel in ED_GF_NORMAL.f90:208-222 (+7 siblings; 56 channels of a 2x2 cluster re-open the same four sectors
 * N+-1).  What hxv_create_from_model builds -- basis maps, one-spin matrices, tile plan, ~50 device tables -- depends only on the model
 * bytes, (nup, ndw), (rank, nranks), the exchange and the device; the engine keeps the images of closed sectors and a re-open with the
 * same inputs shares them (identical products bit for bit; a different bath is a different key).  Environment: HXV_SECTOR_CACHE=0 disables
 * it, HXV_SECTOR_CACHE_MB caps host + device bytes (default 2048, least recently used out first).  hxv_get_option(h, "open_cache_hit" |
 * "open_us_host" | "open_us_plan" | "open_us_upload" | "open_us_total") tells what THIS open cost.  Handles from hxv_create_from_csr and
 * panel handles are not cached.                                                                                                */
int hxv_sector_cache_clear(void);
int hxv_sector_cache_stats(int64_t *entries, int64_t *bytes, int64_t *hits, int64_t *misses); /* any out may be NULL */

/* Handles created (hxv_create_*; the internal row-panel handle of an exchange-mode-2 sector counts too) and not yet destroyed, process-wide:
 * what a host program's leak check reads after its last delete_Hv_sector (the Fortran demo asserts 0 after its lifetime cases).            */
int64_t hxv_live_handles(void);

/* ---- introspection (parity tests against spH0ups/spH0dws/spH0d) ------------------------ */
int hxv_get_maps(const hxv_handle *h, int32_t *map_up, int32_t *map_dw); /* Hs(1)%map, Hs(2)%map */
int64_t hxv_nnz(const hxv_handle *h, int32_t which);                      /* 0: H_up, 1: H_dw */
/* CSR of the one-spin matrix in the reference's row-list order; cols 1-based */
int hxv_get_csr(const hxv_handle *h, int32_t which, int64_t *rowptr, int32_t *cols, double *vals);
/* local diagonal (vecdim doubles; the engine requires a real diagonal) */
int hxv_get_diag(const hxv_handle *h, double *diag);

/* Options (name, value), in three groups.  A maintainer needs group 1 only.
 *
 * 1. BEHAVIOUR (what the engine computes or which algorithm runs; results stay within the stated tolerances)
 *   "kernel"            1 = tiled two-pass kernels [default], 0 = one thread per element (cross-check / fallback)
 *   "real_vectors"      1 [default] = device Lanczos drivers run on real vectors when H and the start vector are real
 *   "lanczos_fused"     1 [default] = recurrence fused into the product's epilogue; 0 = separate vector kernels
 *   "lanczos_graph"     1 [default] = fixed-length tridiagonalisations run device-only, three iterations per hipGraph
 *   "lanczos_inplace"   1 [default] = on a split sector the device Lanczos vectors live in their slot of a gather buffer (three full-size
 *                       buffers per rank instead of one plus three slabs; no slab copy per product; same numbers bit for bit);
 *                       get "slab_copies" counts the exchanges that had to copy
 *   "eigh_degenerate"   0 [default] = one Krylov space like ARPACK; 1 = hxv_eigh_lowest locks the converged pairs and looks for further copies of
 *                       degenerate levels (C3: +60 products on 380; at worst as many again)
 *   "eigh_measure_all"  0 [default] = partial re-orthogonalisation (loss of orthogonality estimated by the omega recurrence, whole-basis
 *                       Gram-Schmidt only when needed); 1 = every projection measured at every step
 *   "eigh_keep_pct"     5..80 [20]: share of the Krylov basis beyond the wanted pairs that a thick restart of hxv_eigh_lowest keeps
 *   "eigh_fuse_restart" 1 [default] = the restart rotation also measures the residual vector, the first step of a cycle removes the arrow and
 *                       measures in one pass; 0 = separate passes (same algorithm)
 *   "fold_nd"           1 [default] = spH0nd inside pass A; 0 = as its own pass over Hv
 *   "exchange_overlap"  0 [default] | 1: exchange mode 2 of a split sector runs diagonal + up hops on a second stream WHILE the two transposes and
 *                       the panel product are under way (the reference's order, ED_HAMILTONIAN_SPARSE_HxV.f90:250-296) and adds the dw part at the
 *                       end; plain products only; 32 B per local state more HBM traffic, for hiding pass A behind the links
 *
 * 2. TILE SHAPE AND SCHEDULING (results unchanged up to summation order; changing a shape rebuilds a handle-private plan, the shared sector image
 *    keeps the default one; invalid combinations are refused with a message)
 *   "cols_per_tile" 2|4|8 [4; complex vectors use at most 4 per tile], "rows_per_tile" 0|2|4|8 [0 = 4, or 8 for sectors whose row panels exceed
 *   the L2], "lds_budget_kb[_up|_dw]" 8..144 [64], "threads_up|_dw" 256|512|1024 [1024], "sort_mode" 0..2 [0], "sort_mode_dw" 0|1 [1],
 *   "wt_cols" 2|4|8|16 [4], "tile_bits_up|_dw" (force the block bits), "lds_min_kb_up|_dw", "spread_banks" 0|1 [1].
 *   Pass A as pipelined jobs (LDS-DMA tile ring, one workgroup per CU): "job_up" 2 [default: for the fused Lanczos product only] |
 *   1 (always) | 0 (one tile per workgroup), "job_groups" columns per job [about 100], "job_cols" 1, "job_stages" ring depth 2..8 [4],
 *   "job_max_blocks" [32].  The engine falls back to the one-tile kernels where jobs do not apply (real vectors, stored diagonal, more than
 *   24 in-block / 16 out-of-block entries per row, blocks over 960 rows).
 *   "wt_colmajor" 0|1 [1: the blocked dw-hop scratch holds column-major patches -- pass A's accumulator init reads R*16 contiguous bytes per column
 *   and patch instead of every lane its own 64-byte stretch; bit-identical],
 *   "real_dw_pairs" 0|1 [1: pass B of the REAL-vector product runs the complex kernel on pairs of rows -- one table decode and one 16-byte LDS
 *   gather per two elements, bit-identical],
 *   "pair_rows" -1|0|1 [-1 = by sector size: pass B runs the two row groups of a 128-byte line back to back],
 *   "block_order" -1|0|1|2 [-1: dispatch order of a group's blocks -- 1 by the particle number of the high orbitals (coupled blocks close
 *   together; the automatic choice where table classes are few), 0 largest block first, 2 natural].
 *
 * 3. TIMING EXPERIMENTS (results are wrong or partial when set; refused unless HXV_EXPERIMENTS=1 is in the environment)
 *   "passes" 1|2|3 [3], "debug" bit mask, "job_debug" bit mask.
 *
 * hxv_get_option additionally reports plan statistics ("tile_bits_up", "nblocks_up", "slots_in_up_x100", "max_outer_up", "job_up_active", ...),
 * driver read-backs ("lanczos_real_last", "eigh_last_full_passes", "eigh_last_local_passes", "eigh_last_search_products",
 * "eigh_last_check_products", "slab_copies") and what the open cost ("open_cache_hit", "open_us_host|plan|upload|total").        */
int hxv_set_option(hxv_handle *h, const char *name, int64_t value);
int64_t hxv_get_option(const hxv_handle *h, const char *name);

typedef struct {
  int64_t n_apply;          /* products since creation                     */
  int64_t algorithmic_bytes; /* 32 B x local rows per product   */
  int64_t device_bytes;     /* bytes of device memory held by the handle   */
  int32_t kernel;           /* active kernel variant                       */
  int32_t real_h;           /* 1 if all hopping amplitudes are real        */
  int32_t k_up, k_dw;       /* max stored entries per row of H_up / H_dw   */
  int32_t n_hops_up, n_hops_dw;
  int64_t h2d_bytes;        /* vector-sized host -> device traffic of the host-array entry points (hxv_apply_host, *_host drivers,   */
  int64_t d2h_bytes;        /* hxv_vector_from_host) and device -> host (results, hxv_vector_to_host) since creation: 0 + 0 for a    */
                            /* Green's-function channel that runs device-resident                                                  */
} hxv_stats;
int hxv_get_stats(const hxv_handle *h, hxv_stats *out);

/* last error message of the calling thread ("" if none) */
const char *hxv_last_error(void);
/* library version string */
const char *hxv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HXV_H */

!> ISO_C_BINDING glue between the reference's Fortran host and the HIP engine (include/hxv.h).
!!
!! It provides a module procedure with EXACTLY the abstract interface cc_sparse_HxV
!! (ED_VARS_GLOBAL.f90:72-78), so that the reference's pointer can be bound to it,
!!     spHtimesV_p => gpuMatVec_main            (ED_HAMILTONIAN.f90:129-141)
!! and ED_DIAG / ED_GF_NORMAL (the SciFortran Lanczos callers) stay untouched, plus the
!! open/close hooks that build_Hv_sector / delete_Hv_sector call (ED_HAMILTONIAN.f90:39-190).
!! Errors follow the reference's convention: a non-zero status from the C side becomes `stop`.
!! INTEGRATION.md shows the few lines added to ED_HAMILTONIAN.f90.
module ED_HAMILTONIAN_GPU_HXV
  use, intrinsic :: iso_c_binding
  implicit none
  private

  public :: gpu_build_Hv_sector
  public :: gpu_delete_Hv_sector
  public :: gpu_vecDim_Hv_sector
  public :: gpuMatVec_main
  public :: gpuMatVec_MPI_main
  public :: gpu_comm_unique_id
  public :: gpu_comm_init
  !sp_eigh's replacement can look for hidden copies of degenerate levels (engine option "eigh_degenerate", include/hxv.h).  THIS GLUE ASKS FOR
  !THEM BY DEFAULT (ADVICE r5): it is the drop-in path that feeds es_add_state, ED_DIAG.f90:234-244 keeps every state within gs_threshold of
  !the minimum, and a 2x2 plaquette (point group D4) has degenerate levels inside a sector -- losing a copy gives a symmetry-broken G.  It
  !costs the check rounds (C3: 60 products on 380).  Set .false. BEFORE build_Hv_sector for exactly what ARPACK does (one Krylov space, no
  !extra products; the raw C-ABI's default).
  logical,public,save :: gpu_eigh_degenerate=.true.
  public :: gpu_lanc_tridiag_host
  public :: gpu_sp_lanc_tridiag
  public :: gpu_sp_lanc_tridiag_pair
  public :: gpu_sp_lanc_eigh
  public :: gpu_sp_eigh
  !device-resident Green's-function pipeline (nothing Dim-sized crosses PCIe): see the block comment above gpu_sp_lanc_eigh_dev
  public :: gpu_vector
  public :: gpu_sp_lanc_eigh_dev
  public :: gpu_sp_eigh_dev
  public :: gpu_keep_sector
  public :: gpu_apply_ladder
  public :: gpu_sp_lanc_tridiag_dev
  public :: gpu_sp_lanc_tridiag_pair_dev
  public :: gpu_vector_to_host
  public :: gpu_vector_from_host
  public :: gpu_free_vector
  public :: gpu_pcie_bytes
  public :: gpu_live_sectors
  !the reference's own stored matrices handed over (spH0ups(1), spH0dws(1), spH0d, spH0nd flattened to CSR)
  public :: gpu_build_Hv_sector_from_csr
  public :: gpu_set_nonlocal_csr
  !what the engine built for the open sector, in the layout gpu_build_Hv_sector_from_csr takes (parity against spH0ups / spH0dws / spH0d)
  public :: gpu_sector_nnz
  public :: gpu_get_sector_csr
  public :: gpu_get_sector_diag

  !> A vector that lives on the device, in the layout of the sector it was made for.  Opaque: pass it back to the gpu_* routines.
  type :: gpu_vector
     type(c_ptr) :: d      = c_null_ptr    !device buffer (include/hxv.h: hxv_vector_alloc)
     type(c_ptr) :: sector = c_null_ptr    !the sector (engine handle) it belongs to
     logical     :: owns_sector = .false.  !gpu_keep_sector was called with this vector (informational: the sector's life is counted, below)
     logical     :: view = .false.         !part of another vector's allocation (eigenvectors 2.. of gpu_sp_eigh_dev)
     type(c_ptr) :: base   = c_null_ptr    !the allocation it lives in (= d unless a view)
     integer(8)  :: sector_id = 0          !serial numbers of the sector and of the allocation in the lifetime tables below: addresses are
     integer(8)  :: alloc_id  = 0          !re-issued by malloc / the engine's buffer cache once freed, serials never (ADVICE r5)
  end type gpu_vector

  !Lifetimes are COUNTED, so vectors may be freed in any order (ADVICE r4): an allocation goes back to the engine when the last
  !gpu_vector in it is freed (the eigenvectors of gpu_sp_eigh_dev share one), and a kept sector (gpu_keep_sector) is closed when the last
  !vector made on it is freed -- whichever vector that is.  A sector closed by gpu_delete_Hv_sector while vectors of it are still alive
  !takes their memory with it (hxv_destroy returns it): those vectors become empty shells and freeing them later is a no-op.
  !Entries are identified by a SERIAL NUMBER taken when the sector is opened / the allocation is made, never by address: a sector closed by
  !gpu_delete_Hv_sector with live vectors keeps its (dead) entry until those shells are freed, and the next build_Hv_sector very likely gets
  !the same handle address -- matched by address, the new sector would inherit dead=.true. and leak (ADVICE r5).
  type :: ref_entry
     integer(8)  :: id   = 0
     integer     :: n    = 0
     logical     :: kept = .false.         !sectors: stays open until its last vector is freed
     logical     :: dead = .false.         !sectors: destroyed while vectors were alive
  end type ref_entry
  integer,parameter    :: MAXREF=512
  type(ref_entry),save :: sector_refs(MAXREF), alloc_refs(MAXREF)
  integer(8),save      :: next_serial=0      !last serial number handed out (sectors and allocations share the counter)
  integer(8),save      :: handle_serial=0    !serial of the OPEN sector (`handle`), 0: none

  !> SciFortran's drivers are generic in exactly this way: the serial form takes the product first, the MPI form the
  !! communicator first (call sites ED_DIAG.f90:152-156,161-165,176-184; ED_GF_NORMAL.f90:215,217).  Both forms end in
  !! the same device driver: on a split sector the engine's own communicator (gpu_comm_init) does the all-reduces, so
  !! MpiComm is accepted for call compatibility only.
  interface gpu_sp_eigh
     module procedure gpu_sp_eigh_serial, gpu_sp_eigh_mpi
  end interface gpu_sp_eigh
  interface gpu_sp_lanc_eigh
     module procedure gpu_sp_lanc_eigh_serial, gpu_sp_lanc_eigh_mpi
  end interface gpu_sp_lanc_eigh
  interface gpu_sp_lanc_tridiag
     module procedure gpu_sp_lanc_tridiag_serial, gpu_sp_lanc_tridiag_mpi
  end interface gpu_sp_lanc_tridiag

  !> mirrors struct hxv_stats of include/hxv.h
  type, bind(C) :: hxv_stats
     integer(c_int64_t) :: n_apply, algorithmic_bytes, device_bytes
     integer(c_int32_t) :: kernel, real_h, k_up, k_dw, n_hops_up, n_hops_dw
     integer(c_int64_t) :: h2d_bytes, d2h_bytes
  end type hxv_stats

  !> mirrors struct hxv_model of include/hxv.h
  type, bind(C) :: hxv_model
     integer(c_int32_t) :: nlat, norb, nspin, nbath
     integer(c_int32_t) :: hfmode, reserved
     real(c_double)     :: uloc(5)
     real(c_double)     :: ust, jh, jx, jp, xmu
     type(c_ptr)        :: imphloc, hbath, vbath
  end type hxv_model

  interface
     integer(c_int) function hxv_create_from_model(model,nup,ndw,rank,nranks,device,out) bind(C,name="hxv_create_from_model")
       import :: c_int, c_int32_t, c_ptr, hxv_model
       type(hxv_model),intent(in)    :: model
       integer(c_int32_t),value      :: nup,ndw,rank,nranks,device
       type(c_ptr),intent(out)       :: out
     end function hxv_create_from_model
     integer(c_int) function hxv_set_option(h,name,value) bind(C,name="hxv_set_option")
       import :: c_ptr,c_int,c_char,c_int64_t
       type(c_ptr),value             :: h
       character(kind=c_char)        :: name(*)
       integer(c_int64_t),value      :: value
     end function hxv_set_option
     integer(c_int) function hxv_destroy(h) bind(C,name="hxv_destroy")
       import :: c_int, c_ptr
       type(c_ptr),value :: h
     end function hxv_destroy
     integer(c_int64_t) function hxv_live_handles() bind(C,name="hxv_live_handles")
       import :: c_int64_t
     end function hxv_live_handles
     integer(c_int64_t) function hxv_vecdim(h) bind(C,name="hxv_vecdim")
       import :: c_int64_t, c_ptr
       type(c_ptr),value :: h
     end function hxv_vecdim
     integer(c_int) function hxv_apply_host(h,nloc,v,hv) bind(C,name="hxv_apply_host")
       import :: c_int, c_int64_t, c_ptr, c_double_complex
       type(c_ptr),value                    :: h
       integer(c_int64_t),value             :: nloc
       complex(c_double_complex),intent(in) :: v(*)
       complex(c_double_complex)            :: hv(*)
     end function hxv_apply_host
     integer(c_int) function hxv_lanczos_tridiag_host(h,vin,nlanc,alanc,blanc,threshold,nsteps) bind(C,name="hxv_lanczos_tridiag_host")
       import :: c_int, c_int32_t, c_ptr, c_double, c_double_complex
       type(c_ptr),value                    :: h
       complex(c_double_complex),intent(in) :: vin(*)
       integer(c_int32_t),value             :: nlanc
       real(c_double)                       :: alanc(*),blanc(*)
       real(c_double),value                 :: threshold
       integer(c_int32_t)                   :: nsteps
     end function hxv_lanczos_tridiag_host
     integer(c_int) function hxv_lanczos_tridiag_pair_host(h,vin_a,vin_b,nlanc,alanc_a,blanc_a,alanc_b,blanc_b,threshold,nsteps_a,nsteps_b) &
          bind(C,name="hxv_lanczos_tridiag_pair_host")
       import :: c_int, c_int32_t, c_ptr, c_double, c_double_complex
       type(c_ptr),value                    :: h
       complex(c_double_complex),intent(in) :: vin_a(*),vin_b(*)
       integer(c_int32_t),value             :: nlanc
       real(c_double)                       :: alanc_a(*),blanc_a(*),alanc_b(*),blanc_b(*)
       real(c_double),value                 :: threshold
       integer(c_int32_t)                   :: nsteps_a,nsteps_b
     end function hxv_lanczos_tridiag_pair_host
     integer(c_int) function hxv_lanczos_eigh_host(h,nitermax,threshold,egs,vect,niter) bind(C,name="hxv_lanczos_eigh_host")
       import :: c_int, c_int32_t, c_ptr, c_double, c_double_complex
       type(c_ptr),value                    :: h
       integer(c_int32_t),value             :: nitermax
       real(c_double),value                 :: threshold
       real(c_double)                       :: egs
       complex(c_double_complex)            :: vect(*)
       integer(c_int32_t)                   :: niter
     end function hxv_lanczos_eigh_host
     integer(c_int) function hxv_eigh_lowest_host(h,neigen,ncv,maxrestart,tol,evals,evecs,nconv,nmatvec) bind(C,name="hxv_eigh_lowest_host")
       import :: c_int, c_int32_t, c_ptr, c_double, c_double_complex
       type(c_ptr),value                    :: h
       integer(c_int32_t),value             :: neigen,ncv,maxrestart
       real(c_double),value                 :: tol
       real(c_double)                       :: evals(*)
       complex(c_double_complex)            :: evecs(*)
       integer(c_int32_t)                   :: nconv,nmatvec
     end function hxv_eigh_lowest_host
     integer(c_int) function hxv_comm_unique_id(id) bind(C,name="hxv_comm_unique_id")
       import :: c_int, c_int8_t
       integer(c_int8_t) :: id(128)
     end function hxv_comm_unique_id
     integer(c_int) function hxv_comm_init(h,id) bind(C,name="hxv_comm_init")
       import :: c_int, c_int8_t, c_ptr
       type(c_ptr),value            :: h
       integer(c_int8_t),intent(in) :: id(128)
     end function hxv_comm_init
     integer(c_int) function hxv_create_from_csr(dimup,dimdw,up_rowptr,up_cols,up_vals,dw_rowptr,dw_cols,dw_vals,diag,rank,nranks,device,out) &
          bind(C,name="hxv_create_from_csr")
       import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double_complex
       integer(c_int32_t),value             :: dimup,dimdw,rank,nranks,device
       integer(c_int64_t),intent(in)        :: up_rowptr(*),dw_rowptr(*)
       integer(c_int32_t),intent(in)        :: up_cols(*),dw_cols(*)
       complex(c_double_complex),intent(in) :: up_vals(*),dw_vals(*),diag(*)
       type(c_ptr),intent(out)              :: out
     end function hxv_create_from_csr
     integer(c_int) function hxv_set_nonlocal_csr(h,rowptr,cols,vals) bind(C,name="hxv_set_nonlocal_csr")
       import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double_complex
       type(c_ptr),value                    :: h
       integer(c_int64_t),intent(in)        :: rowptr(*)
       integer(c_int32_t),intent(in)        :: cols(*)
       complex(c_double_complex),intent(in) :: vals(*)
     end function hxv_set_nonlocal_csr
     integer(c_int64_t) function hxv_nnz(h,which) bind(C,name="hxv_nnz")
       import :: c_int64_t, c_int32_t, c_ptr
       type(c_ptr),value        :: h
       integer(c_int32_t),value :: which
     end function hxv_nnz
     integer(c_int) function hxv_get_csr(h,which,rowptr,cols,vals) bind(C,name="hxv_get_csr")
       import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double_complex
       type(c_ptr),value         :: h
       integer(c_int32_t),value  :: which
       integer(c_int64_t)        :: rowptr(*)
       integer(c_int32_t)        :: cols(*)
       complex(c_double_complex) :: vals(*)
     end function hxv_get_csr
     integer(c_int) function hxv_get_diag(h,diag) bind(C,name="hxv_get_diag")
       import :: c_int, c_ptr, c_double
       type(c_ptr),value :: h
       real(c_double)    :: diag(*)
     end function hxv_get_diag
     integer(c_int) function hxv_vector_alloc(h,d_vec) bind(C,name="hxv_vector_alloc")
       import :: c_int, c_ptr
       type(c_ptr),value       :: h
       type(c_ptr),intent(out) :: d_vec
     end function hxv_vector_alloc
     integer(c_int) function hxv_vector_alloc_many(h,count,d_vec) bind(C,name="hxv_vector_alloc_many")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr),value        :: h
       integer(c_int32_t),value :: count
       type(c_ptr),intent(out)  :: d_vec
     end function hxv_vector_alloc_many
     integer(c_int64_t) function hxv_localvec_elems(h) bind(C,name="hxv_localvec_elems")
       import :: c_int64_t, c_ptr
       type(c_ptr),value :: h
     end function hxv_localvec_elems
     integer(c_int) function hxv_eigh_lowest(h,neigen,ncv,maxrestart,tol,evals,d_evecs,nconv,nmatvec) bind(C,name="hxv_eigh_lowest")
       import :: c_int, c_int32_t, c_ptr, c_double
       type(c_ptr),value        :: h,d_evecs
       integer(c_int32_t),value :: neigen,ncv,maxrestart
       real(c_double),value     :: tol
       real(c_double)           :: evals(*)
       integer(c_int32_t)       :: nconv,nmatvec
     end function hxv_eigh_lowest
     integer(c_int) function hxv_vector_free(h,d_vec) bind(C,name="hxv_vector_free")
       import :: c_int, c_ptr
       type(c_ptr),value :: h,d_vec
     end function hxv_vector_free
     integer(c_int) function hxv_vector_from_host(h,v_host,d_vec) bind(C,name="hxv_vector_from_host")
       import :: c_int, c_ptr, c_double_complex
       type(c_ptr),value                    :: h,d_vec
       complex(c_double_complex),intent(in) :: v_host(*)
     end function hxv_vector_from_host
     integer(c_int) function hxv_vector_to_host(h,d_vec,v_host) bind(C,name="hxv_vector_to_host")
       import :: c_int, c_ptr, c_double_complex
       type(c_ptr),value         :: h,d_vec
       complex(c_double_complex) :: v_host(*)
     end function hxv_vector_to_host
     integer(c_int) function hxv_lanczos_eigh(h,nitermax,threshold,egs,d_vect,niter) bind(C,name="hxv_lanczos_eigh")
       import :: c_int, c_int32_t, c_ptr, c_double
       type(c_ptr),value        :: h,d_vect
       integer(c_int32_t),value :: nitermax
       real(c_double),value     :: threshold
       real(c_double)           :: egs
       integer(c_int32_t)       :: niter
     end function hxv_lanczos_eigh
     integer(c_int) function hxv_lanczos_tridiag(h,d_vin,nlanc,alanc,blanc,threshold,nsteps) bind(C,name="hxv_lanczos_tridiag")
       import :: c_int, c_int32_t, c_ptr, c_double
       type(c_ptr),value        :: h,d_vin
       integer(c_int32_t),value :: nlanc
       real(c_double)           :: alanc(*),blanc(*)
       real(c_double),value     :: threshold
       integer(c_int32_t)       :: nsteps
     end function hxv_lanczos_tridiag
     integer(c_int) function hxv_lanczos_tridiag_pair(h,d_vin_a,d_vin_b,nlanc,alanc_a,blanc_a,alanc_b,blanc_b,threshold,nsteps_a,nsteps_b) &
          bind(C,name="hxv_lanczos_tridiag_pair")
       import :: c_int, c_int32_t, c_ptr, c_double
       type(c_ptr),value        :: h,d_vin_a,d_vin_b
       integer(c_int32_t),value :: nlanc
       real(c_double)           :: alanc_a(*),blanc_a(*),alanc_b(*),blanc_b(*)
       real(c_double),value     :: threshold
       integer(c_int32_t)       :: nsteps_a,nsteps_b
     end function hxv_lanczos_tridiag_pair
     integer(c_int) function hxv_apply_ladder_axpy(from,to,orbital,spin,create,coef_re,coef_im,accumulate,d_psi,d_out,norm2) &
          bind(C,name="hxv_apply_ladder_axpy")
       import :: c_int, c_int32_t, c_ptr, c_double
       type(c_ptr),value        :: from,to,d_psi,d_out
       integer(c_int32_t),value :: orbital,spin,create,accumulate
       real(c_double),value     :: coef_re,coef_im
       real(c_double)           :: norm2
     end function hxv_apply_ladder_axpy
     integer(c_int) function hxv_get_stats(h,st) bind(C,name="hxv_get_stats")
       import :: c_int, c_ptr, hxv_stats
       type(c_ptr),value :: h
       type(hxv_stats)   :: st
     end function hxv_get_stats
     type(c_ptr) function hxv_last_error() bind(C,name="hxv_last_error")
       import :: c_ptr
     end function hxv_last_error
  end interface

  type(c_ptr),save :: handle = c_null_ptr   !one open sector at a time (ED_HAMILTONIAN_COMMON.f90:17-18)

contains

  subroutine check(ierr,where)
    integer(c_int)   :: ierr
    character(len=*) :: where
    character(kind=c_char),pointer :: msg(:)
    integer :: n
    if(ierr==0)return
    call c_f_pointer(hxv_last_error(),msg,[512])
    n=1
    do while(n<512.and.msg(n)/=c_null_char)
       n=n+1
    enddo
    write(*,"(A)")trim(where)//" ERROR: "//transfer(msg(1:n-1),repeat(" ",n-1))
    stop "hxv engine error"
  end subroutine check

  !> Device-side part of build_Hv_sector(isector): arguments are the module globals the reference
  !! has in scope at ED_HAMILTONIAN.f90:129 (Nlat..Nbath ED_INPUT_VARS.f90:13-16; impHloc
  !! ED_VARS_GLOBAL.f90:119; Hbath_reconstructed / diag_hybr as built at
  !! ED_HAMILTONIAN_SPARSE_HxV.f90:62-76; Uloc..xmu,hfmode ED_INPUT_VARS.f90:129-135,164;
  !! nup,ndw = get_Nup/get_Ndw(isector); MpiRank,MpiSize after the communicator shrink).
  subroutine gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,nup,ndw,impHloc,Hbath,Vbath,Uloc,Ust,Jh,Jx,Jp,xmu,hfmode,MpiRank,MpiSize,device)
    integer,intent(in)                   :: Nlat,Norb,Nspin,Nbath,nup,ndw
    complex(8),intent(in),target,contiguous :: impHloc(:,:,:,:,:,:)   ![Nlat,Nlat,Nspin,Nspin,Norb,Norb]
    complex(8),intent(in),target,contiguous :: Hbath(:,:,:,:,:,:,:)   ![Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath]
    real(8),intent(in),target,contiguous    :: Vbath(:,:,:,:)         ![Nlat,Nspin,Norb,Nbath]
    real(8),intent(in)                   :: Uloc(5),Ust,Jh,Jx,Jp,xmu
    logical,intent(in)                   :: hfmode
    integer,intent(in)                   :: MpiRank,MpiSize
    integer,intent(in),optional          :: device
    type(hxv_model)                      :: m
    integer                              :: dev
    if(c_associated(handle))stop "gpu_build_Hv_sector ERROR: a sector is already open"
    dev=0;if(present(device))dev=device
    m%nlat=Nlat; m%norb=Norb; m%nspin=Nspin; m%nbath=Nbath
    m%hfmode=0; if(hfmode)m%hfmode=1
    m%reserved=0
    m%uloc=Uloc
    m%ust=Ust; m%jh=Jh; m%jx=Jx; m%jp=Jp; m%xmu=xmu
    m%imphloc=c_loc(impHloc)
    m%hbath=c_null_ptr; m%vbath=c_null_ptr
    if(Nbath>0)then
       m%hbath=c_loc(Hbath)
       m%vbath=c_loc(Vbath)
    endif
    call check(hxv_create_from_model(m,int(nup,c_int32_t),int(ndw,c_int32_t),int(MpiRank,c_int32_t),&
         int(MpiSize,c_int32_t),int(dev,c_int32_t),handle),"gpu_build_Hv_sector")
    call sector_opened()
    call check(hxv_set_option(handle,"eigh_degenerate"//c_null_char,merge(1_c_int64_t,0_c_int64_t,gpu_eigh_degenerate)),"gpu_build_Hv_sector")
  end subroutine gpu_build_Hv_sector

  !---- reference tables of the device vectors (see type ref_entry) ----
  integer function ref_find(tab,id) result(i)
    type(ref_entry),intent(in) :: tab(:)
    integer(8),intent(in)      :: id
    if(id/=0)then
       do i=1,size(tab)
          if(tab(i)%n>0.and.tab(i)%id==id)return
       enddo
    endif
    i=0
  end function ref_find
  subroutine ref_add(tab,id)
    type(ref_entry),intent(inout) :: tab(:)
    integer(8),intent(in)         :: id
    integer                       :: i
    i=ref_find(tab,id)
    if(i==0)then
       do i=1,size(tab)
          if(tab(i)%n==0)exit
       enddo
       if(i>size(tab))stop "ED_HAMILTONIAN_GPU_HxV ERROR: more than 512 live device vectors / sectors"
       tab(i)%id=id; tab(i)%kept=.false.; tab(i)%dead=.false.
    endif
    tab(i)%n=tab(i)%n+1
  end subroutine ref_add
  !a new gpu_vector of the OPEN sector, in allocation `base`: its own (view=.false.: a fresh serial) or the one `owner` lives in
  subroutine vec_born(vect,base,view,owner)
    type(gpu_vector),intent(inout)       :: vect
    type(c_ptr),intent(in)               :: base
    logical,intent(in)                   :: view
    type(gpu_vector),intent(in),optional :: owner
    vect%sector=handle; vect%sector_id=handle_serial; vect%base=base; vect%view=view; vect%owns_sector=.false.
    if(view)then
       if(.not.present(owner))stop "ED_HAMILTONIAN_GPU_HxV ERROR: a view needs the vector that owns its allocation"
       vect%alloc_id=owner%alloc_id
    else
       next_serial=next_serial+1
       vect%alloc_id=next_serial
    endif
    call ref_add(alloc_refs,vect%alloc_id)
    call ref_add(sector_refs,vect%sector_id)
  end subroutine vec_born
  !a device vector that can still be used: made, not freed, its sector not closed under it
  logical function vec_alive(vect) result(ok)
    type(gpu_vector),intent(in) :: vect
    integer                     :: i
    ok=.false.
    if(.not.c_associated(vect%d))return
    i=ref_find(sector_refs,vect%sector_id)
    if(i==0)return
    ok=.not.sector_refs(i)%dead
  end function vec_alive
  !the open sector got its serial (every build_Hv_sector variant calls this right after the engine returned the handle)
  subroutine sector_opened()
    next_serial=next_serial+1
    handle_serial=next_serial
  end subroutine sector_opened

  !> delete_Hv_sector hook (ED_HAMILTONIAN.f90:149-190)
  subroutine gpu_delete_Hv_sector()
    integer :: i
    if(c_associated(handle))then
       i=ref_find(sector_refs,handle_serial)
       if(i>0)sector_refs(i)%dead=.true.   !vectors of this sector are still alive: hxv_destroy takes their memory back, their shells stay
       call check(hxv_destroy(handle),"gpu_delete_Hv_sector")
    endif
    handle=c_null_ptr; handle_serial=0
  end subroutine gpu_delete_Hv_sector

  !> vecDim_Hv_sector of the open sector (ED_HAMILTONIAN.f90:197-221)
  function gpu_vecDim_Hv_sector() result(vecDim)
    integer :: vecDim
    if(.not.c_associated(handle))stop "gpu_vecDim_Hv_sector ERROR: Hsector NOT set"
    vecDim=int(hxv_vecdim(handle))
  end function gpu_vecDim_Hv_sector

  !> The product, with the cc_sparse_HxV interface (ED_VARS_GLOBAL.f90:72-78): Hv = H*v.
  subroutine gpuMatVec_main(Nloc,v,Hv)
    integer                    :: Nloc
    complex(8),dimension(Nloc) :: v
    complex(8),dimension(Nloc) :: Hv
    if(.not.c_associated(handle))stop "gpuMatVec_main ERROR: Hsector NOT set"
    call check(hxv_apply_host(handle,int(Nloc,c_int64_t),v,Hv),"gpuMatVec_main")
  end subroutine gpuMatVec_main

  !> Target of the pointer when MpiStatus=T (the reference binds spMatVec_MPI_main there, ED_HAMILTONIAN.f90:131-134):
  !! v, Hv are this rank's slab of vecDim_Hv_sector elements; the engine all-gathers the slabs itself (hxv_apply_host).
  subroutine gpuMatVec_MPI_main(Nloc,v,Hv)
    integer                    :: Nloc
    complex(8),dimension(Nloc) :: v
    complex(8),dimension(Nloc) :: Hv
    if(.not.c_associated(handle))stop "gpuMatVec_MPI_main ERROR: Hsector NOT set"
    call check(hxv_apply_host(handle,int(Nloc,c_int64_t),v,Hv),"gpuMatVec_MPI_main")
  end subroutine gpuMatVec_MPI_main

  !> Communicator of the open (split) sector.  Host side, after gpu_build_Hv_sector on every rank of MpiComm:
  !!     if(MpiRank==0)call gpu_comm_unique_id(id)
  !!     call MPI_Bcast(id,128,MPI_BYTE,0,MpiComm,ierr)
  !!     call gpu_comm_init(id)
  !! (gpu_delete_Hv_sector releases it with the sector.)
  subroutine gpu_comm_unique_id(id)
    integer(c_int8_t),intent(out) :: id(128)
    call check(hxv_comm_unique_id(id),"gpu_comm_unique_id")
  end subroutine gpu_comm_unique_id

  subroutine gpu_comm_init(id)
    integer(c_int8_t),intent(in) :: id(128)
    if(.not.c_associated(handle))stop "gpu_comm_init ERROR: Hsector NOT set"
    call check(hxv_comm_init(handle,id),"gpu_comm_init")
  end subroutine gpu_comm_init

  !> Plain Lanczos tridiagonalisation driven through a cc_sparse_HxV procedure on HOST vectors:
  !! the call shape of SciFortran's sp_lanc_tridiag(MatVec,vin,alanc,blanc) as consumed at
  !! ED_GF_NORMAL.f90:215-220,949-951 (alanc(k)=<q_k|H|q_k>, blanc(k+1)=beta_{k+1}, blanc(1) unused).
  !! SciFortran is not vendored with the reference; this stands where it would be linked.
  subroutine gpu_lanc_tridiag_host(MatVec,vin,alanc,blanc,threshold)
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    complex(8),intent(inout) :: vin(:)
    real(8),intent(inout)    :: alanc(:),blanc(:)
    real(8),intent(in),optional :: threshold
    complex(8),allocatable   :: q(:),qm(:),w(:)
    real(8)                  :: a,b,thr
    integer                  :: k,n,nlanc
    n=size(vin); nlanc=size(alanc)
    thr=1d-12; if(present(threshold))thr=threshold
    allocate(q(n),qm(n),w(n))
    q=vin/sqrt(dble(dot_product(vin,vin))); qm=(0d0,0d0); b=0d0
    alanc=0d0; blanc=0d0
    do k=1,nlanc
       call MatVec(n,q,w)
       w=w-b*qm
       a=dble(dot_product(q,w))
       w=w-a*q
       alanc(k)=a
       b=sqrt(dble(dot_product(w,w)))
       if(k<nlanc)blanc(k+1)=b
       if(abs(b)<thr)exit
       qm=q
       q=w/b
    enddo
    deallocate(q,qm,w)
  end subroutine gpu_lanc_tridiag_host

  !> Device-resident Lanczos with the SciFortran CALL SIGNATURES, so that the two call sites can switch by
  !! changing one `use`:   sp_lanc_tridiag(MatVec,vin,alanc,blanc)        ED_GF_NORMAL.f90:215-220 (+7)
  !!                        sp_lanc_eigh(MatVec,egs,vect,Nitermax,iverbose,threshold)   ED_DIAG.f90:176-184
  !! MatVec is accepted for signature compatibility and not called: the product of the OPEN sector runs on the
  !! device; vin/vect cross PCIe once per run instead of twice per iteration.
  subroutine gpu_sp_lanc_tridiag_serial(MatVec,vin,alanc,blanc,threshold)
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    complex(8),intent(inout)    :: vin(:)
    real(8),intent(inout)       :: alanc(:),blanc(:)
    real(8),intent(in),optional :: threshold
    real(8)                     :: thr
    integer(c_int32_t)          :: nsteps
    if(.not.c_associated(handle))stop "gpu_sp_lanc_tridiag ERROR: Hsector NOT set"
    thr=1d-12; if(present(threshold))thr=threshold
    !(the engine normalises the start vector itself, as SciFortran's sp_lanc_tridiag does on its first iteration; on a
    ! split sector the norm is the global one)
    call check(hxv_lanczos_tridiag_host(handle,vin,int(size(alanc),c_int32_t),alanc,blanc,thr,nsteps),"gpu_sp_lanc_tridiag")
  end subroutine gpu_sp_lanc_tridiag_serial

  !> TWO tridiagonalisations on one product (real H): two of the independent channels of lanc_build_gf_normal_*
  !! (ED_GF_NORMAL.f90:123-306; one sp_lanc_tridiag each at :215,282,422,504,638,720,801,882) share the product's passes, their
  !! real start vectors travelling as real and imaginary part of one complex Lanczos vector (H(x+iy) = Hx + iHy).  Same
  !! argument meaning as sp_lanc_tridiag, once per channel; vin_a, vin_b must have zero imaginary parts.
  subroutine gpu_sp_lanc_tridiag_pair(MatVec,vin_a,vin_b,alanc_a,blanc_a,alanc_b,blanc_b,threshold)
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    complex(8),intent(inout)    :: vin_a(:),vin_b(:)
    real(8),intent(inout)       :: alanc_a(:),blanc_a(:),alanc_b(:),blanc_b(:)
    real(8),intent(in),optional :: threshold
    real(8)                     :: thr
    integer(c_int32_t)          :: na,nb
    if(.not.c_associated(handle))stop "gpu_sp_lanc_tridiag_pair ERROR: Hsector NOT set"
    if(size(alanc_a)/=size(alanc_b))stop "gpu_sp_lanc_tridiag_pair ERROR: the two channels need equally long alanc/blanc"
    thr=1d-12; if(present(threshold))thr=threshold
    call check(hxv_lanczos_tridiag_pair_host(handle,vin_a,vin_b,int(size(alanc_a),c_int32_t),alanc_a,blanc_a,alanc_b,blanc_b,thr,na,nb),&
         "gpu_sp_lanc_tridiag_pair")
  end subroutine gpu_sp_lanc_tridiag_pair

  subroutine gpu_sp_lanc_eigh_serial(MatVec,egs,vect,Nitermax,iverbose,threshold)
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    real(8),intent(inout)       :: egs
    complex(8),intent(inout)    :: vect(:)
    integer,intent(in)          :: Nitermax
    logical,intent(in),optional :: iverbose
    real(8),intent(in),optional :: threshold
    real(8)                     :: thr
    integer(c_int32_t)          :: niter
    if(.not.c_associated(handle))stop "gpu_sp_lanc_eigh ERROR: Hsector NOT set"
    thr=1d-12; if(present(threshold))thr=max(threshold,1d-15)
    call check(hxv_lanczos_eigh_host(handle,int(Nitermax,c_int32_t),thr,egs,vect,niter),"gpu_sp_lanc_eigh")
    if(present(iverbose))then
       if(iverbose)write(*,"(A,I6,A,F20.12)")"gpu_sp_lanc_eigh: iterations=",niter," E0=",egs
    endif
  end subroutine gpu_sp_lanc_eigh_serial

  !> sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol,iverbose) -- the default (lanc_method="arpack") spectrum call at
  !! ED_DIAG.f90:152-160 -- on the device: size(eval) lowest eigenpairs, Krylov basis of Nblock vectors in HBM.
  subroutine gpu_sp_eigh_serial(MatVec,eval,evec,Nblock,Nitermax,tol,iverbose)
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    real(8),intent(inout)       :: eval(:)
    complex(8),intent(inout)    :: evec(:,:)
    integer,intent(in),optional :: Nblock,Nitermax
    real(8),intent(in),optional :: tol
    logical,intent(in),optional :: iverbose
    integer(c_int32_t)          :: ncv,nit,nconv,nmv
    real(8)                     :: tl
    if(.not.c_associated(handle))stop "gpu_sp_eigh ERROR: Hsector NOT set"
    if(size(evec,2)<size(eval))stop "gpu_sp_eigh ERROR: size(evec,2) < size(eval)"
    ncv=0;   if(present(Nblock))ncv=int(Nblock,c_int32_t)
    !Nblock = min(dim,lanc_ncv_factor*max(Neigen,lanc_nstates_sector)+lanc_ncv_add) (ED_DIAG.f90:96) exceeds the engine's
    !largest Krylov basis (64 vectors) once lanc_nstates_sector >= 7: clamp instead of stopping the run
    if(ncv>64)then
       write(*,"(A,I6,A)")"gpu_sp_eigh WARNING: Nblock=",ncv," > 64: using a Krylov basis of 64 vectors"
       ncv=64
    endif
    if(size(eval)>=64)stop "gpu_sp_eigh ERROR: more than 63 eigenpairs per sector exceed the engine's Krylov basis (64 vectors)"
    !Nitermax bounds the RESTARTS here (each restart is up to ncv products; ARPACK's bound counts restarts too)
    nit=512; if(present(Nitermax))nit=int(Nitermax,c_int32_t)
    tl=0d0;  if(present(tol))tl=tol
    call check(hxv_eigh_lowest_host(handle,int(size(eval),c_int32_t),ncv,nit,tl,eval,evec,nconv,nmv),"gpu_sp_eigh")
    if(present(iverbose))then
       if(iverbose)write(*,"(A,I4,A,I6,A,F20.12)")"gpu_sp_eigh: converged=",nconv," matvecs=",nmv," E0=",eval(1)
    endif
  end subroutine gpu_sp_eigh_serial

  !> ---------------------------------------------------------------------------------------------------------------------------
  !! DEVICE-RESIDENT Green's-function channel.  The reference builds every channel on the host (ED_GF_NORMAL.f90:174-217): the ground
  !! state comes out of the eigensolver into a host array, c^dagger|gs> is formed by a serial loop on the master, scattered, and handed
  !! to sp_lanc_tridiag -- with the drivers above that is three Dim-sized PCIe transfers per channel.  Here the ground state stays where
  !! the eigensolver left it and the start vector is made next to it:
  !!     call gpu_build_Hv_sector(...isector...)                        !sector of the ground state
  !!     call gpu_sp_lanc_eigh_dev(egs,gs,Nitermax,threshold)           !gs: type(gpu_vector), stays on the device
  !!     call gpu_keep_sector(gs)                                       !the sector stays open for gs; another one can be built
  !!     call gpu_build_Hv_sector(...jsector...)                        !sector of c^dagger|gs>  (ED_GF_NORMAL.f90:176)
  !!     call gpu_apply_ladder(gs,ipos,ispin,.true.,vv,norm2)           !vv = c^dagger_{ipos,ispin}|gs>, norm2 = <vv|vv>  (:180-199)
  !!     call gpu_sp_lanc_tridiag_dev(vv,alfa_,beta_)                   !(:215) the engine normalises vv itself
  !!     call gpu_free_vector(vv); call gpu_delete_Hv_sector()
  !!     ... further channels from the same gs ...
  !!     call gpu_free_vector(gs)                                       !closes the ground state's sector too
  !! Mixed channels (:370-406, :746-780): a second gpu_apply_ladder into the same vv with accumulate=.true. and coef.
  !! On a split sector (after gpu_comm_init) every rank holds and builds its own slab.
  !! ---------------------------------------------------------------------------------------------------------------------------
  subroutine gpu_sp_lanc_eigh_dev(egs,vect,Nitermax,iverbose,threshold)
    real(8),intent(inout)          :: egs
    type(gpu_vector),intent(inout) :: vect
    integer,intent(in)             :: Nitermax
    logical,intent(in),optional    :: iverbose
    real(8),intent(in),optional    :: threshold
    real(8)                        :: thr
    integer(c_int32_t)             :: niter
    if(.not.c_associated(handle))stop "gpu_sp_lanc_eigh_dev ERROR: Hsector NOT set"
    if(c_associated(vect%d))stop "gpu_sp_lanc_eigh_dev ERROR: the vector is in use (gpu_free_vector it first)"
    call check(hxv_vector_alloc(handle,vect%d),"gpu_sp_lanc_eigh_dev")
    call vec_born(vect,vect%d,.false.)
    thr=1d-12; if(present(threshold))thr=max(threshold,1d-15)
    call check(hxv_lanczos_eigh(handle,int(Nitermax,c_int32_t),thr,egs,vect%d,niter),"gpu_sp_lanc_eigh_dev")
    if(present(iverbose))then
       if(iverbose)write(*,"(A,I6,A,F20.12)")"gpu_sp_lanc_eigh_dev: iterations=",niter," E0=",egs
    endif
  end subroutine gpu_sp_lanc_eigh_dev

  !> sp_eigh (the default lanc_method="arpack" call, ED_DIAG.f90:152-160) with the eigenvectors left on the device: vects(1:size(eval)), all in
  !! one allocation that is returned to the engine when the LAST of them is freed (any order).  Any of them can be handed to
  !! gpu_keep_sector / gpu_apply_ladder like the vector of gpu_sp_lanc_eigh_dev.
  subroutine gpu_sp_eigh_dev(eval,vects,Nblock,Nitermax,tol,iverbose)
    real(8),intent(inout)          :: eval(:)
    type(gpu_vector),intent(inout) :: vects(:)
    integer,intent(in),optional    :: Nblock,Nitermax
    real(8),intent(in),optional    :: tol
    logical,intent(in),optional    :: iverbose
    integer(c_int32_t)             :: ncv,nit,nconv,nmv
    real(8)                        :: tl
    integer                        :: i
    integer(c_int64_t)             :: stride
    complex(c_double_complex),pointer :: base(:)
    if(.not.c_associated(handle))stop "gpu_sp_eigh_dev ERROR: Hsector NOT set"
    if(size(vects)<size(eval))stop "gpu_sp_eigh_dev ERROR: size(vects) < size(eval)"
    if(size(eval)>=64)stop "gpu_sp_eigh_dev ERROR: more than 63 eigenpairs per sector exceed the engine's Krylov basis (64 vectors)"
    do i=1,size(eval)
       if(c_associated(vects(i)%d))stop "gpu_sp_eigh_dev ERROR: a vector is in use (gpu_free_vector it first)"
    enddo
    ncv=0;   if(present(Nblock))ncv=int(min(Nblock,64),c_int32_t)
    nit=512; if(present(Nitermax))nit=int(Nitermax,c_int32_t)
    tl=0d0;  if(present(tol))tl=tol
    call check(hxv_vector_alloc_many(handle,int(size(eval),c_int32_t),vects(1)%d),"gpu_sp_eigh_dev")
    stride=hxv_localvec_elems(handle)
    call c_f_pointer(vects(1)%d,base,[stride*size(eval)])      !(address arithmetic only: the memory is on the device)
    do i=1,size(eval)
       if(i>1)vects(i)%d=c_loc(base(1+(i-1)*stride))
       if(i==1)then; call vec_born(vects(1),vects(1)%d,.false.); else; call vec_born(vects(i),vects(1)%d,.true.,vects(1)); endif
    enddo
    call check(hxv_eigh_lowest(handle,int(size(eval),c_int32_t),ncv,nit,tl,eval,vects(1)%d,nconv,nmv),"gpu_sp_eigh_dev")
    if(present(iverbose))then
       if(iverbose)write(*,"(A,I4,A,I6,A,F20.12)")"gpu_sp_eigh_dev: converged=",nconv," matvecs=",nmv," E0=",eval(1)
    endif
  end subroutine gpu_sp_eigh_dev

  !> The open sector stays open for the vectors made on it (it is needed again when c / c^dagger act on them); the module's "one open
  !! sector" slot becomes free, so build_Hv_sector of another sector may follow.  The kept sector is closed when the LAST vector made on
  !! it is freed (gpu_free_vector), in whatever order -- eigenvectors 2.. of gpu_sp_eigh_dev included.
  subroutine gpu_keep_sector(vect)
    type(gpu_vector),intent(inout) :: vect
    integer                        :: i
    if(.not.c_associated(handle))stop "gpu_keep_sector ERROR: Hsector NOT set"
    if(vect%sector_id/=handle_serial.or..not.c_associated(vect%d))stop "gpu_keep_sector ERROR: the vector does not belong to the open sector"
    i=ref_find(sector_refs,handle_serial)
    if(i==0)stop "gpu_keep_sector ERROR: no live vector of the open sector"
    sector_refs(i)%kept=.true.
    vect%owns_sector=.true.
    handle=c_null_ptr; handle_serial=0
  end subroutine gpu_keep_sector

  !> out = [out +] coef * c^(dagger)_{ipos,ispin} psi, from psi's sector into the OPEN sector; norm2 = <out|out> afterwards.
  !! ipos = 1-based orbital position in the spin string (iorb+(ilat-1)*Norb, ED_GF_NORMAL.f90:176-199), ispin = 1 (up) | 2 (dw).
  subroutine gpu_apply_ladder(psi,ipos,ispin,create,out,norm2,coef,accumulate)
    type(gpu_vector),intent(in)    :: psi
    integer,intent(in)             :: ipos,ispin
    logical,intent(in)             :: create
    type(gpu_vector),intent(inout) :: out
    real(8),intent(out)            :: norm2
    complex(8),intent(in),optional :: coef
    logical,intent(in),optional    :: accumulate
    complex(8)                     :: cf
    integer(c_int32_t)             :: acc,cr
    if(.not.c_associated(handle))stop "gpu_apply_ladder ERROR: Hsector NOT set (build the target sector first)"
    if(.not.vec_alive(psi))stop "gpu_apply_ladder ERROR: empty source vector, or its sector was closed under it (gpu_keep_sector keeps it open)"
    cf=(1d0,0d0); if(present(coef))cf=coef
    acc=0; if(present(accumulate))then; if(accumulate)acc=1; endif
    if(.not.c_associated(out%d))then
       if(acc==1)stop "gpu_apply_ladder ERROR: accumulate into an empty vector"
       call check(hxv_vector_alloc(handle,out%d),"gpu_apply_ladder")
       call vec_born(out,out%d,.false.)
    endif
    if(out%sector_id/=handle_serial)stop "gpu_apply_ladder ERROR: the target vector does not belong to the open sector"
    cr=0; if(create)cr=1
    call check(hxv_apply_ladder_axpy(psi%sector,handle,int(ipos-1,c_int32_t),int(ispin-1,c_int32_t),cr,dble(cf),aimag(cf),acc,psi%d,out%d,norm2),&
         "gpu_apply_ladder")
  end subroutine gpu_apply_ladder

  !> sp_lanc_tridiag (ED_GF_NORMAL.f90:215) from a start vector that is on the device already (normalised by the engine).
  subroutine gpu_sp_lanc_tridiag_dev(vin,alanc,blanc,threshold)
    type(gpu_vector),intent(in) :: vin
    real(8),intent(inout)       :: alanc(:),blanc(:)
    real(8),intent(in),optional :: threshold
    real(8)                     :: thr
    integer(c_int32_t)          :: nsteps
    if(.not.c_associated(handle))stop "gpu_sp_lanc_tridiag_dev ERROR: Hsector NOT set"
    if(vin%sector_id/=handle_serial.or..not.c_associated(vin%d))stop "gpu_sp_lanc_tridiag_dev ERROR: the start vector does not belong to the open sector"
    thr=1d-12; if(present(threshold))thr=threshold
    call check(hxv_lanczos_tridiag(handle,vin%d,int(size(alanc),c_int32_t),alanc,blanc,thr,nsteps),"gpu_sp_lanc_tridiag_dev")
  end subroutine gpu_sp_lanc_tridiag_dev

  !> TWO channels of the same sector on one product (real H; gpu_sp_lanc_tridiag_pair with start vectors that are on the device already):
  !! e.g. c^dagger_{i,up}|gs> and c^dagger_{j,up}|gs> of one ground state -- both real, both in the open sector.
  subroutine gpu_sp_lanc_tridiag_pair_dev(vin_a,vin_b,alanc_a,blanc_a,alanc_b,blanc_b,threshold)
    type(gpu_vector),intent(in) :: vin_a,vin_b
    real(8),intent(inout)       :: alanc_a(:),blanc_a(:),alanc_b(:),blanc_b(:)
    real(8),intent(in),optional :: threshold
    real(8)                     :: thr
    integer(c_int32_t)          :: na,nb
    if(.not.c_associated(handle))stop "gpu_sp_lanc_tridiag_pair_dev ERROR: Hsector NOT set"
    if(vin_a%sector_id/=handle_serial.or.vin_b%sector_id/=handle_serial.or..not.c_associated(vin_a%d).or..not.c_associated(vin_b%d))&
         stop "gpu_sp_lanc_tridiag_pair_dev ERROR: the start vectors do not belong to the open sector"
    if(size(alanc_a)/=size(alanc_b))stop "gpu_sp_lanc_tridiag_pair_dev ERROR: the two channels need equally long alanc/blanc"
    thr=1d-12; if(present(threshold))thr=threshold
    call check(hxv_lanczos_tridiag_pair(handle,vin_a%d,vin_b%d,int(size(alanc_a),c_int32_t),alanc_a,blanc_a,alanc_b,blanc_b,thr,na,nb),&
         "gpu_sp_lanc_tridiag_pair_dev")
  end subroutine gpu_sp_lanc_tridiag_pair_dev

  !> the vector in the reference's host layout (this rank's slab), when it is wanted there after all (e.g. state_list of ED_DIAG)
  subroutine gpu_vector_to_host(vect,v)
    type(gpu_vector),intent(in) :: vect
    complex(8),intent(inout)    :: v(:)
    if(.not.vec_alive(vect))stop "gpu_vector_to_host ERROR: empty vector, or its sector was closed under it"
    if(int(size(v),c_int64_t)/=hxv_vecdim(vect%sector))stop "gpu_vector_to_host ERROR: size(v) /= vecDim of the vector's sector"
    call check(hxv_vector_to_host(vect%sector,vect%d,v),"gpu_vector_to_host")
  end subroutine gpu_vector_to_host

  subroutine gpu_vector_from_host(v,vect)
    complex(8),intent(in)          :: v(:)
    type(gpu_vector),intent(inout) :: vect
    if(.not.c_associated(vect%d))then
       if(.not.c_associated(handle))stop "gpu_vector_from_host ERROR: Hsector NOT set"
       call check(hxv_vector_alloc(handle,vect%d),"gpu_vector_from_host")
       call vec_born(vect,vect%d,.false.)
    endif
    if(.not.vec_alive(vect))stop "gpu_vector_from_host ERROR: the vector's sector was closed under it"
    if(int(size(v),c_int64_t)/=hxv_vecdim(vect%sector))stop "gpu_vector_from_host ERROR: size(v) /= vecDim of the vector's sector"
    call check(hxv_vector_from_host(vect%sector,v,vect%d),"gpu_vector_from_host")
  end subroutine gpu_vector_from_host

  !> Any order: the allocation goes back when its last vector is freed, a kept sector is closed when its last vector is freed.
  subroutine gpu_free_vector(vect)
    type(gpu_vector),intent(inout) :: vect
    integer                        :: ia,is
    if(c_associated(vect%d))then
       is=ref_find(sector_refs,vect%sector_id)
       ia=ref_find(alloc_refs,vect%alloc_id)
       if(is==0.or.ia==0)stop "gpu_free_vector ERROR: not a live device vector (a copy of one that was freed already?)"
       alloc_refs(ia)%n=alloc_refs(ia)%n-1
       if(alloc_refs(ia)%n==0)then
          if(.not.sector_refs(is)%dead)call check(hxv_vector_free(vect%sector,vect%base),"gpu_free_vector")
          alloc_refs(ia)%id=0
       endif
       sector_refs(is)%n=sector_refs(is)%n-1
       if(sector_refs(is)%n==0)then
          if(sector_refs(is)%kept.and..not.sector_refs(is)%dead)then
             if(vect%sector_id==handle_serial)then; handle=c_null_ptr; handle_serial=0; endif   !(kept sectors are never the open one; belt and braces)
             call check(hxv_destroy(vect%sector),"gpu_free_vector")
          endif
          sector_refs(is)%id=0; sector_refs(is)%kept=.false.; sector_refs(is)%dead=.false.
       endif
    endif
    vect%d=c_null_ptr; vect%sector=c_null_ptr; vect%base=c_null_ptr; vect%owns_sector=.false.; vect%view=.false.; vect%sector_id=0; vect%alloc_id=0
  end subroutine gpu_free_vector

  !> Sectors (engine handles) opened and not yet closed, process-wide (include/hxv.h: hxv_live_handles): a host program's leak check.
  function gpu_live_sectors() result(n)
    integer(8) :: n
    n=hxv_live_handles()
  end function gpu_live_sectors

  !> Vector-sized bytes the engine has moved over PCIe for a sector since it was opened (include/hxv.h: hxv_stats): the open sector,
  !! or the kept sector of `vect`.
  subroutine gpu_pcie_bytes(h2d,d2h,vect)
    integer(8),intent(out)               :: h2d,d2h
    type(gpu_vector),intent(in),optional :: vect
    type(hxv_stats)                      :: st
    type(c_ptr)                          :: hh
    hh=handle
    if(present(vect))then
       if(.not.vec_alive(vect))stop "gpu_pcie_bytes ERROR: empty vector, or its sector was closed under it"
       hh=vect%sector
    endif
    if(.not.c_associated(hh))stop "gpu_pcie_bytes ERROR: no sector"
    call check(hxv_get_stats(hh,st),"gpu_pcie_bytes")
    h2d=st%h2d_bytes; d2h=st%d2h_bytes
  end subroutine gpu_pcie_bytes

  !> build_Hv_sector from the reference's OWN stored matrices (ED_HAMILTONIAN_SPARSE_HxV.f90:40-152 has built spH0ups(1), spH0dws(1),
  !! spH0d): flatten each sparse_matrix_csr (ED_SPARSE_MATRIX.f90:13-30) row by row --
  !!     rowptr(1)=0; do i=1,Nrow; rowptr(i+1)=rowptr(i)+sparse%row(i)%Size
  !!                              cols(rowptr(i)+1:rowptr(i+1))=sparse%row(i)%cols; vals(...)=sparse%row(i)%vals; enddo
  !! (columns stay 1-based, as sp_insert_element stores them) -- and the diagonal spH0d as its local rows, one value per row.
  subroutine gpu_build_Hv_sector_from_csr(DimUp,DimDw,up_rowptr,up_cols,up_vals,dw_rowptr,dw_cols,dw_vals,diag,MpiRank,MpiSize,device)
    integer,intent(in)          :: DimUp,DimDw
    integer(8),intent(in)       :: up_rowptr(:),dw_rowptr(:)     ![DimUp+1], [DimDw+1], 0-based offsets
    integer,intent(in)          :: up_cols(:),dw_cols(:)         !1-based columns
    complex(8),intent(in)       :: up_vals(:),dw_vals(:),diag(:) !diag: mpiQdw*DimUp local rows
    integer,intent(in)          :: MpiRank,MpiSize
    integer,intent(in),optional :: device
    integer                     :: dev
    if(c_associated(handle))stop "gpu_build_Hv_sector_from_csr ERROR: a sector is already open"
    if(size(up_rowptr)/=DimUp+1.or.size(dw_rowptr)/=DimDw+1)stop "gpu_build_Hv_sector_from_csr ERROR: rowptr sizes"
    dev=0;if(present(device))dev=device
    call check(hxv_create_from_csr(int(DimUp,c_int32_t),int(DimDw,c_int32_t),int(up_rowptr,c_int64_t),int(up_cols,c_int32_t),up_vals,&
         int(dw_rowptr,c_int64_t),int(dw_cols,c_int32_t),dw_vals,diag,int(MpiRank,c_int32_t),int(MpiSize,c_int32_t),int(dev,c_int32_t),handle),&
         "gpu_build_Hv_sector_from_csr")
    call sector_opened()
    call check(hxv_set_option(handle,"eigh_degenerate"//c_null_char,merge(1_c_int64_t,0_c_int64_t,gpu_eigh_degenerate)),"gpu_build_Hv_sector_from_csr")
  end subroutine gpu_build_Hv_sector_from_csr

  !> stored elements of H_up (which=1) / H_dw (which=2) of the open sector
  function gpu_sector_nnz(which) result(nnz)
    integer,intent(in) :: which
    integer(8)         :: nnz
    if(.not.c_associated(handle))stop "gpu_sector_nnz ERROR: Hsector NOT set"
    nnz=hxv_nnz(handle,int(which-1,c_int32_t))
  end function gpu_sector_nnz

  !> H_up (which=1) / H_dw (which=2) of the open sector as the engine built it, flattened like spH0ups(1) / spH0dws(1) would be for
  !! gpu_build_Hv_sector_from_csr: rowptr(DimSigma+1) 0-based offsets, cols 1-based, vals complex
  subroutine gpu_get_sector_csr(which,rowptr,cols,vals)
    integer,intent(in)       :: which
    integer(8),intent(out)   :: rowptr(:)
    integer,intent(out)      :: cols(:)
    complex(8),intent(out)   :: vals(:)
    integer(c_int64_t),allocatable :: rp(:)
    integer(c_int32_t),allocatable :: cl(:)
    if(.not.c_associated(handle))stop "gpu_get_sector_csr ERROR: Hsector NOT set"
    if(int(size(cols),8)<hxv_nnz(handle,int(which-1,c_int32_t)).or.size(vals)<size(cols))stop "gpu_get_sector_csr ERROR: cols / vals too short"
    allocate(rp(size(rowptr)),cl(size(cols)))
    call check(hxv_get_csr(handle,int(which-1,c_int32_t),rp,cl,vals),"gpu_get_sector_csr")
    rowptr=rp; cols=cl
    deallocate(rp,cl)
  end subroutine gpu_get_sector_csr

  !> the (real) diagonal of the open sector, local rows -- spH0d's values
  subroutine gpu_get_sector_diag(diag)
    real(8),intent(out) :: diag(:)
    if(.not.c_associated(handle))stop "gpu_get_sector_diag ERROR: Hsector NOT set"
    if(int(size(diag),c_int64_t)/=hxv_vecdim(handle))stop "gpu_get_sector_diag ERROR: size(diag) /= vecDim"
    call check(hxv_get_diag(handle,diag),"gpu_get_sector_diag")
  end subroutine gpu_get_sector_diag

  !> spH0nd (Jx, Jp; ED_VARS_GLOBAL.f90:145, sparse/H_non_local.f90:23-98) of a sector opened from stored matrices: the LOCAL rows with the
  !! reference's GLOBAL 1-based columns, flattened like the others.
  subroutine gpu_set_nonlocal_csr(rowptr,cols,vals)
    integer(8),intent(in) :: rowptr(:)
    integer,intent(in)    :: cols(:)
    complex(8),intent(in) :: vals(:)
    if(.not.c_associated(handle))stop "gpu_set_nonlocal_csr ERROR: Hsector NOT set"
    call check(hxv_set_nonlocal_csr(handle,int(rowptr,c_int64_t),int(cols,c_int32_t),vals),"gpu_set_nonlocal_csr")
  end subroutine gpu_set_nonlocal_csr

  !> MpiComm-first forms: the call text of the reference's MpiStatus=T branches compiles against these unchanged,
  !!   call sp_eigh(MpiComm,spHtimesV_p,eig_values,eig_basis,Nblock,Nitermax,tol=lanc_tolerance,iverbose=(ed_verbose>3))   ED_DIAG.f90:152-156
  !!   call sp_lanc_eigh(MpiComm,spHtimesV_p,eig_values(1),eig_basis(:,1),Nitermax,iverbose=...,threshold=lanc_tolerance)    ED_DIAG.f90:176-177
  !!   call sp_lanc_tridiag(MpiComm,spHtimesV_p,vvloc,alfa_,beta_)                                                            ED_GF_NORMAL.f90:215
  !! with vectors = this rank's slab (vecDim_Hv_sector elements).  MpiComm is the reference's integer communicator.
  subroutine gpu_sp_eigh_mpi(MpiComm,MatVec,eval,evec,Nblock,Nitermax,tol,iverbose)
    integer,intent(in)          :: MpiComm
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    real(8),intent(inout)       :: eval(:)
    complex(8),intent(inout)    :: evec(:,:)
    integer,intent(in),optional :: Nblock,Nitermax
    real(8),intent(in),optional :: tol
    logical,intent(in),optional :: iverbose
    call gpu_sp_eigh_serial(MatVec,eval,evec,Nblock,Nitermax,tol,iverbose)
  end subroutine gpu_sp_eigh_mpi

  subroutine gpu_sp_lanc_eigh_mpi(MpiComm,MatVec,egs,vect,Nitermax,iverbose,threshold)
    integer,intent(in)          :: MpiComm
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    real(8),intent(inout)       :: egs
    complex(8),intent(inout)    :: vect(:)
    integer,intent(in)          :: Nitermax
    logical,intent(in),optional :: iverbose
    real(8),intent(in),optional :: threshold
    call gpu_sp_lanc_eigh_serial(MatVec,egs,vect,Nitermax,iverbose,threshold)
  end subroutine gpu_sp_lanc_eigh_mpi

  subroutine gpu_sp_lanc_tridiag_mpi(MpiComm,MatVec,vin,alanc,blanc,threshold)
    integer,intent(in)          :: MpiComm
    interface
       subroutine MatVec(Nloc,v,Hv)
         integer                    :: Nloc
         complex(8),dimension(Nloc) :: v,Hv
       end subroutine MatVec
    end interface
    complex(8),intent(inout)    :: vin(:)
    real(8),intent(inout)       :: alanc(:),blanc(:)
    real(8),intent(in),optional :: threshold
    call gpu_sp_lanc_tridiag_serial(MatVec,vin,alanc,blanc,threshold)
  end subroutine gpu_sp_lanc_tridiag_mpi

end module ED_HAMILTONIAN_GPU_HXV

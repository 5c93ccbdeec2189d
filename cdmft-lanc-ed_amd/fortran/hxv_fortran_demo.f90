!> Fortran host (the reference's own language) driving the HIP engine through the reference's
!! procedure-pointer surface.  BASELINE config C1: 2x2 Hubbard plaquette, no bath, U=4, t=1,
!! hfmode=F, sector (nup,ndw)=(2,2); the reference's dense H gives E0 = -2.10274848
!! (SURVEY.md 8c).  Also C2-like: 4-site chain + 2 replica baths (Ns=12), sector (6,6).
program hxv_fortran_demo
  use ED_HAMILTONIAN_GPU_HXV
  implicit none
  !the reference's abstract interface and pointer (ED_VARS_GLOBAL.f90:72-78,146)
  abstract interface
     subroutine cc_sparse_HxV(Nloc,v,Hv)
       integer                    :: Nloc
       complex(8),dimension(Nloc) :: v
       complex(8),dimension(Nloc) :: Hv
     end subroutine cc_sparse_HxV
  end interface
  procedure(cc_sparse_HxV),pointer :: spHtimesV_p=>null()

  call plaquette()
  call chain_ns12()
  call gf_channel_on_device()
  call stored_matrices()

contains

  subroutine plaquette()
    integer,parameter :: Nlat=4,Norb=1,Nspin=1,Nbath=0
    complex(8) :: impHloc(Nlat,Nlat,Nspin,Nspin,Norb,Norb),Hbath(Nlat,Nlat,Nspin,Nspin,Norb,Norb,1)
    real(8)    :: Vbath(Nlat,Nspin,Norb,1),Uloc(5)
    complex(8),allocatable :: v(:)
    real(8)    :: alanc(36),blanc(36),e0
    integer    :: dim,i
    impHloc=(0d0,0d0); Hbath=(0d0,0d0); Vbath=0d0
    !bonds 1-2,1-3,2-4,3-4 (drivers/cdn_hm_2dsquare.f90:231-250), t=1
    impHloc(1,2,1,1,1,1)=-1d0; impHloc(2,1,1,1,1,1)=-1d0
    impHloc(1,3,1,1,1,1)=-1d0; impHloc(3,1,1,1,1,1)=-1d0
    impHloc(2,4,1,1,1,1)=-1d0; impHloc(4,2,1,1,1,1)=-1d0
    impHloc(3,4,1,1,1,1)=-1d0; impHloc(4,3,1,1,1,1)=-1d0
    Uloc=0d0; Uloc(1)=4d0
    call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,2,2,impHloc,Hbath(:,:,:,:,:,:,1:0),Vbath(:,:,:,1:0),Uloc,0d0,0d0,0d0,0d0,0d0,.false.,0,1)
    spHtimesV_p => gpuMatVec_main
    dim=gpu_vecDim_Hv_sector()
    allocate(v(dim))
    do i=1,dim
       v(i)=cmplx(sin(0.37d0*(i-1)+0.11d0),cos(0.23d0*(i-1)+0.05d0),8)
    enddo
    call gpu_lanc_tridiag_host(spHtimesV_p,v,alanc,blanc)
    e0=lowest_tridiag(alanc,blanc)
    write(*,"(A,I6,A,F16.10)")"C1 plaquette sector(2,2) Dim=",dim," E0=",e0
    spHtimesV_p => null()
    call gpu_delete_Hv_sector()
    deallocate(v)
  end subroutine plaquette

  subroutine chain_ns12()
    integer,parameter :: Nlat=4,Norb=1,Nspin=1,Nbath=2
    complex(8) :: impHloc(Nlat,Nlat,Nspin,Nspin,Norb,Norb),Hbath(Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath)
    real(8)    :: Vbath(Nlat,Nspin,Norb,Nbath),Uloc(5),eps(Nbath)
    complex(8),allocatable :: v(:),hv(:),eig_basis(:,:)
    real(8),allocatable    :: eig_values(:)
    real(8)    :: alanc(200),blanc(200),e0
    integer    :: dim,i,ib
    impHloc=(0d0,0d0); Hbath=(0d0,0d0)
    do i=1,Nlat
       if(i>1)impHloc(i,i-1,1,1,1,1)=-0.25d0
       if(i<Nlat)impHloc(i,i+1,1,1,1,1)=-0.25d0
    enddo
    eps=[0.3d0,0.6d0]
    do ib=1,Nbath
       Hbath(:,:,:,:,:,:,ib)=-abs(impHloc)     !Hsym1=|Hloc|, lambda=-1 (drivers/cdn_hm_1dchain.f90:80-81)
       do i=1,Nlat
          Hbath(i,i,1,1,1,1,ib)=eps(ib)
       enddo
    enddo
    Vbath=1d0/sqrt(2d0)
    Uloc=0d0; Uloc(1)=2d0
    call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
    spHtimesV_p => gpuMatVec_main
    dim=gpu_vecDim_Hv_sector()
    allocate(v(dim),hv(dim))
    do i=1,dim
       v(i)=cmplx(sin(0.37d0*(i-1)+0.11d0),cos(0.23d0*(i-1)+0.05d0),8)
    enddo
    call spHtimesV_p(dim,v,hv)
    write(*,"(A,I8,A,4ES24.16)")"C2 chain sector(6,6) Dim=",dim," Hv(1),Hv(Dim)=",hv(1),hv(dim)
    call gpu_lanc_tridiag_host(spHtimesV_p,v,alanc,blanc)
    e0=lowest_tridiag(alanc,blanc)
    write(*,"(A,I8,A,F16.10)")"C2 chain sector(6,6) Dim=",dim," E0=",e0
    !the same two calls as ED_GF_NORMAL.f90:215 / ED_DIAG.f90:176, SciFortran signatures, recurrence on the device
    v=v/sqrt(dble(dot_product(v,v)))
    call gpu_sp_lanc_tridiag(spHtimesV_p,v,alanc,blanc)
    write(*,"(A,F16.10)")"C2 device tridiag E0=",lowest_tridiag(alanc,blanc)
    call gpu_sp_lanc_eigh(spHtimesV_p,e0,hv,512,threshold=1d-14)
    write(*,"(A,F16.10,A,ES12.4)")"C2 device eigh E0=",e0," |vec|^2-1=",dble(dot_product(hv,hv))-1d0
    !two Green's-function-style channels on one product: real start vectors as Re / Im of one complex Lanczos vector
    block
      complex(8),allocatable :: va(:),vb(:)
      real(8),allocatable    :: aa(:),ba(:),ab(:),bb(:)
      allocate(va(dim),vb(dim),aa(size(alanc)),ba(size(alanc)),ab(size(alanc)),bb(size(alanc)))
      va=cmplx(dble(v),0d0,8);   va=va/sqrt(dble(dot_product(va,va)))
      vb=cmplx(aimag(v),0d0,8);  vb=vb/sqrt(dble(dot_product(vb,vb)))
      call gpu_sp_lanc_tridiag_pair(spHtimesV_p,va,vb,aa,ba,ab,bb)
      write(*,"(A,2F16.10)")"C2 device tridiag pair E0=",lowest_tridiag(aa,ba),lowest_tridiag(ab,bb)
      call gpu_sp_lanc_tridiag(spHtimesV_p,va,alanc,blanc)
      write(*,"(A,F16.10)")"C2 device tridiag channel a alone E0=",lowest_tridiag(alanc,blanc)
      deallocate(va,vb,aa,ba,ab,bb)
    end block
    !the default spectrum call of ED_DIAG.f90:152-160: sp_eigh(spHtimesV_p,eig_values,eig_basis,Nblock,Nitermax,tol=...)
    allocate(eig_values(2),eig_basis(dim,2))
    call gpu_sp_eigh(spHtimesV_p,eig_values,eig_basis,20,512,tol=1d-18)
    call spHtimesV_p(dim,eig_basis(:,2),hv)
    write(*,"(A,2F16.10,A,ES12.4)")"C2 device sp_eigh E=",eig_values," resid2=",&
         sqrt(dble(dot_product(hv-eig_values(2)*eig_basis(:,2),hv-eig_values(2)*eig_basis(:,2))))
    deallocate(eig_values,eig_basis)
    spHtimesV_p => null()
    call gpu_delete_Hv_sector()
    !
    !MpiStatus=T side: the same sector with the slab exchange behind the C-ABI (one-rank communicator here: the RCCL
    !all-gather / all-reduce path runs; with more ranks every rank does exactly this on its slab).  The three call
    !texts are those of ED_DIAG.f90:152-156,176-177 and ED_GF_NORMAL.f90:215.
    call mpi_branch(impHloc,Hbath,Vbath,Uloc)
  end subroutine chain_ns12

  subroutine mpi_branch(impHloc,Hbath,Vbath,Uloc)
    use, intrinsic :: iso_c_binding, only: c_int8_t
    integer,parameter :: Nlat=4,Norb=1,Nspin=1,Nbath=2
    complex(8) :: impHloc(Nlat,Nlat,Nspin,Nspin,Norb,Norb),Hbath(Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath)
    real(8)    :: Vbath(Nlat,Nspin,Norb,Nbath),Uloc(5)
    integer    :: MpiComm,MpiRank,MpiSize,vecDim,Neigen,Nblock,Nitermax,ed_verbose,i
    real(8)    :: lanc_tolerance
    integer(c_int8_t)      :: id(128)
    complex(8),allocatable :: eig_basis(:,:),vvloc(:)
    real(8),allocatable    :: eig_values(:),alfa_(:),beta_(:)
    MpiComm=0; MpiRank=0; MpiSize=1       !(no MPI library in this build: the communicator value is not used by the engine)
    Neigen=2; Nblock=20; Nitermax=512; lanc_tolerance=1d-18; ed_verbose=0
    call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,MpiRank,MpiSize)
    if(MpiRank==0)call gpu_comm_unique_id(id)
    !call MPI_Bcast(id,128,MPI_BYTE,0,MpiComm,ierr)
    call gpu_comm_init(id)
    spHtimesV_p => gpuMatVec_MPI_main
    vecDim=gpu_vecDim_Hv_sector()
    allocate(eig_values(Neigen),eig_basis(vecDim,Neigen))
    call gpu_sp_eigh(MpiComm,spHtimesV_p,eig_values,eig_basis,&
         Nblock,&
         Nitermax,&
         tol=lanc_tolerance,&
         iverbose=(ed_verbose>3))
    write(*,"(A,2F16.10)")"C2 MPI-branch sp_eigh E=",eig_values
    call gpu_sp_lanc_eigh(MpiComm,spHtimesV_p,eig_values(1),eig_basis(:,1),Nitermax,&
         iverbose=(ed_verbose>3),threshold=lanc_tolerance)
    write(*,"(A,F16.10)")"C2 MPI-branch sp_lanc_eigh E0=",eig_values(1)
    allocate(vvloc(vecDim),alfa_(100),beta_(100))
    do i=1,vecDim
       vvloc(i)=cmplx(sin(0.37d0*(i-1)+0.11d0),cos(0.23d0*(i-1)+0.05d0),8)
    enddo
    vvloc=vvloc/sqrt(dble(dot_product(vvloc,vvloc)))
    call gpu_sp_lanc_tridiag(MpiComm,spHtimesV_p,vvloc,alfa_,beta_)
    write(*,"(A,F16.10)")"C2 MPI-branch sp_lanc_tridiag E0=",lowest_tridiag(alfa_,beta_)
    deallocate(eig_values,eig_basis,vvloc,alfa_,beta_)
    spHtimesV_p => null()
    call gpu_delete_Hv_sector()
  end subroutine mpi_branch

  !> One Green's-function channel of lanc_build_gf_normal (ED_GF_NORMAL.f90:174-217), C2 model: ground state of sector (6,6),
  !! c^dagger_{1,up}|gs> in sector (7,6), 100 Lanczos steps there -- first DEVICE-RESIDENT (gpu_sp_lanc_eigh_dev -> gpu_apply_ladder ->
  !! gpu_sp_lanc_tridiag_dev: the engine's PCIe counters must stay at zero), then the way the reference does it (ground state in a host
  !! array, the master's serial loop :180-199 restated below, sp_lanc_tridiag on the host array).  alanc / blanc must agree.
  subroutine gf_channel_on_device()
    integer,parameter :: Nlat=4,Norb=1,Nspin=1,Nbath=2,Ns=12,ipos=1,ispin=1,nl=100
    complex(8) :: impHloc(Nlat,Nlat,Nspin,Nspin,Norb,Norb),Hbath(Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath)
    real(8)    :: Vbath(Nlat,Nspin,Norb,Nbath),Uloc(5),eps(Nbath)
    type(gpu_vector) :: gs,vv
    real(8)    :: egs,norm2,norm2h,a1(nl),b1(nl),a2(nl),b2(nl)
    integer(8) :: h2d_a,d2h_a,h2d_b,d2h_b
    complex(8),allocatable :: psi(:),vvinit(:)
    integer,allocatable    :: map6(:),map7(:),pos7(:)
    integer    :: i,ib,m,n6,n7,iup,idw,jup,dimA,dimB,sgn,k
    impHloc=(0d0,0d0); Hbath=(0d0,0d0)
    do i=1,Nlat
       if(i>1)impHloc(i,i-1,1,1,1,1)=-0.25d0
       if(i<Nlat)impHloc(i,i+1,1,1,1,1)=-0.25d0
    enddo
    eps=[0.3d0,0.6d0]
    do ib=1,Nbath
       Hbath(:,:,:,:,:,:,ib)=-abs(impHloc)
       do i=1,Nlat
          Hbath(i,i,1,1,1,1,ib)=eps(ib)
       enddo
    enddo
    Vbath=1d0/sqrt(2d0)
    Uloc=0d0; Uloc(1)=2d0
    !--- device-resident channel
    call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
    dimA=gpu_vecDim_Hv_sector()
    call gpu_sp_lanc_eigh_dev(egs,gs,512,threshold=1d-14)
    !the default spectrum call (sp_eigh, ED_DIAG.f90:152-160) with its eigenvectors left on the device as well: same ground state
    block
      type(gpu_vector) :: ev(2)
      real(8) :: e2(2)
      call gpu_sp_eigh_dev(e2,ev,20,512,tol=1d-18)
      write(*,"(A,2F16.10,A,ES12.4)")"GF device sp_eigh: E=",e2," |E0 - lanc_eigh E0|=",abs(e2(1)-egs)
      call gpu_free_vector(ev(2)); call gpu_free_vector(ev(1))
    end block
    call gpu_keep_sector(gs)
    call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,7,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
    dimB=gpu_vecDim_Hv_sector()
    call gpu_apply_ladder(gs,ipos,ispin,.true.,vv,norm2)
    a1=0d0; b1=0d0
    call gpu_sp_lanc_tridiag_dev(vv,a1,b1)
    call gpu_pcie_bytes(h2d_a,d2h_a,gs)
    call gpu_pcie_bytes(h2d_b,d2h_b)
    write(*,"(A,F16.10,A,ES24.16,A,4I12)")"GF device channel: E0=",egs," norm2=",norm2," PCIe bytes (h2d,d2h) gs-sector, channel-sector=",&
         h2d_a,d2h_a,h2d_b,d2h_b
    !two channels at once: c^dagger_{1,up}|gs> and c^dagger_{2,up}|gs> as Re / Im of one complex Lanczos vector; channel a must reproduce a1, b1
    block
      type(gpu_vector) :: v2
      real(8) :: n2b,pa(nl),pb(nl),qa(nl),qb(nl)
      call gpu_apply_ladder(gs,2,ispin,.true.,v2,n2b)
      pa=0d0; pb=0d0; qa=0d0; qb=0d0
      call gpu_sp_lanc_tridiag_pair_dev(vv,v2,pa,pb,qa,qb)
      call gpu_pcie_bytes(h2d_b,d2h_b)
      write(*,"(A,ES12.4,A,ES12.4,A,2F16.10,A,2I12)")"GF device pair: max|alanc_a-single|(8)=",maxval(abs(pa(1:8)-a1(1:8)))," max|blanc_a-single|(8)=",maxval(abs(pb(1:8)-b1(1:8))),&
           " lowest Ritz values a,b=",lowest_tridiag(pa,pb),lowest_tridiag(qa,qb)," PCIe bytes=",h2d_b,d2h_b
      call gpu_free_vector(v2)
    end block
    call gpu_free_vector(vv)
    !--- the reference's way: ground state on the host, c^dagger by the master's loop, host start vector
    allocate(psi(dimA),vvinit(dimB))
    call gpu_vector_to_host(gs,psi)
    allocate(map6(924),map7(792),pos7(0:2**Ns-1))
    n6=0; n7=0; pos7=0
    do m=0,2**Ns-1
       if(popcnt(m)==6)then; n6=n6+1; map6(n6)=m; endif
       if(popcnt(m)==7)then; n7=n7+1; map7(n7)=m; pos7(m)=n7; endif
    enddo
    vvinit=(0d0,0d0)
    do idw=1,n6                       !sector (6,6) -> (7,6): DimDw = C(12,6) both, DimUp 924 -> 792
       do iup=1,n6
          m=map6(iup)
          if(btest(m,ipos-1))cycle
          sgn=1; if(mod(popcnt(iand(m,2**(ipos-1)-1)),2)==1)sgn=-1
          jup=pos7(ibset(m,ipos-1))
          vvinit(jup+(idw-1)*n7)=sgn*psi(iup+(idw-1)*n6)
       enddo
    enddo
    norm2h=dble(dot_product(vvinit,vvinit))
    vvinit=vvinit/sqrt(norm2h)
    a2=0d0; b2=0d0
    spHtimesV_p => gpuMatVec_main
    call gpu_sp_lanc_tridiag(spHtimesV_p,vvinit,a2,b2)
    !(entry by entry on the early steps only: once the extremal Ritz values converge the recurrence amplifies rounding differences,
    ! and the quantity the consumer uses -- the spectrum of the tridiagonal matrix, ED_GF_NORMAL.f90:949-953 -- is compared instead)
    k=8
    write(*,"(A,ES12.4,A,ES12.4,A,ES12.4,A,2F16.10)")"GF host-array channel: |norm2 diff|=",abs(norm2h-norm2)," max|da|(8)=",maxval(abs(a1(1:k)-a2(1:k))),&
         " max|db|(8)=",maxval(abs(b1(1:k)-b2(1:k)))," lowest Ritz values=",lowest_tridiag(a1,b1),lowest_tridiag(a2,b2)
    call gpu_pcie_bytes(h2d_b,d2h_b)
    write(*,"(A,2I12)")"GF host-array channel: PCIe bytes (h2d,d2h) channel-sector=",h2d_b,d2h_b
    spHtimesV_p => null()
    call gpu_delete_Hv_sector()
    call gpu_free_vector(gs)
    !--- lifetimes in the awkward order (ADVICE r4): the sector is kept through a VIEW (eigenvector 2 of gpu_sp_eigh_dev), the first
    !    eigenvector -- whose pointer is the allocation -- is freed first, and a further vector made on the kept sector outlives the keeper
    block
      type(gpu_vector) :: ev(2),extra,w1,w2
      real(8) :: e2(2),n2c,n2d
      complex(8),allocatable :: back(:)
      call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
      call gpu_sp_eigh_dev(e2,ev,20,512,tol=1d-18)
      call gpu_vector_from_host(psi,extra)                 !a third vector on the same sector
      call gpu_keep_sector(ev(2))                          !kept through the view
      call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,7,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
      call gpu_apply_ladder(ev(1),ipos,ispin,.true.,w1,n2c)
      call gpu_free_vector(ev(1))                          !the allocation must survive: ev(2) lives in it
      call gpu_apply_ladder(ev(2),ipos,ispin,.true.,w2,n2d)
      call gpu_free_vector(w1); call gpu_free_vector(w2)
      call gpu_free_vector(ev(2))                          !the keeper goes; `extra` still needs the sector
      allocate(back(dimA))
      call gpu_vector_to_host(extra,back)
      call gpu_free_vector(extra)                          !last vector of the kept sector: closes it
      call gpu_delete_Hv_sector()
      write(*,"(A,ES12.4,A,ES12.4,A,ES12.4)")"GF lifetimes any order: |norm2(ev1) - norm2(gs)|=",abs(n2c-norm2)," norm2(ev2)=",n2d,&
           " max|roundtrip - psi|=",maxval(abs(back-psi))
      deallocate(back)
    end block
    !--- a sector closed UNDER a live vector, then the next sector opened at (very likely) the same handle address and the next vectors at the
    !    same device addresses (ADVICE r5): the new sector must not inherit the dead entry of the old one -- kept through a vector and freed,
    !    it has to be closed and its vector memory returned; the shell of the old sector's vector is freed last, a no-op
    block
      type(gpu_vector) :: orphan,keeper,w
      real(8) :: eo,ek,n2k
      integer(8) :: live0,live1,live2
      live0=gpu_live_sectors()
      call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
      call gpu_sp_lanc_eigh_dev(eo,orphan,300,threshold=1d-13)
      call gpu_delete_Hv_sector()                          !closed with `orphan` alive: its memory goes with the sector, the shell stays
      call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
      call gpu_sp_lanc_eigh_dev(ek,keeper,300,threshold=1d-13)
      call gpu_keep_sector(keeper)
      call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,7,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
      call gpu_apply_ladder(keeper,ipos,ispin,.true.,w,n2k)
      live1=gpu_live_sectors()                             !the kept sector and the open one
      call gpu_free_vector(w)
      call gpu_delete_Hv_sector()
      call gpu_free_vector(keeper)                         !last vector of the kept sector: must CLOSE it
      live2=gpu_live_sectors()
      call gpu_free_vector(orphan)                         !shell of a vector whose sector is long gone: no-op
      write(*,"(A,3I4,A,ES12.4,A,ES12.4)")"GF lifetimes after a sector closed under a live vector: live sectors before / kept+open / after=",&
           live0,live1,live2," |E0(new) - E0(old)|=",abs(ek-eo)," |norm2 - norm2(gs)|=",abs(n2k-norm2)
    end block
    deallocate(psi,vvinit,map6,map7,pos7)
  end subroutine gf_channel_on_device

  !> The engine fed with STORED matrices, as a maintainer would hand over spH0ups(1), spH0dws(1), spH0d (gpu_build_Hv_sector_from_csr): here the
  !! matrices come from a sector the engine built from the model itself (gpu_get_sector_csr / _diag), the C2-like chain, sector (6,6); the
  !! product through the stored matrices must equal the product of the model-built sector.
  subroutine stored_matrices()
    integer,parameter :: Nlat=4,Norb=1,Nspin=1,Nbath=2
    complex(8) :: impHloc(Nlat,Nlat,Nspin,Nspin,Norb,Norb),Hbath(Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath)
    real(8)    :: Vbath(Nlat,Nspin,Norb,Nbath),Uloc(5),eps(Nbath)
    complex(8),allocatable :: v(:),hv1(:),hv2(:),uv(:),dv(:),dg(:)
    real(8),allocatable    :: diag(:)
    integer(8),allocatable :: urp(:),drp(:)
    integer,allocatable    :: ucl(:),dcl(:)
    integer    :: dim,i,ib,DimUp,DimDw
    integer(8) :: nu,nd
    impHloc=(0d0,0d0); Hbath=(0d0,0d0)
    do i=1,Nlat
       if(i>1)impHloc(i,i-1,1,1,1,1)=-0.25d0
       if(i<Nlat)impHloc(i,i+1,1,1,1,1)=-0.25d0
    enddo
    eps=[0.3d0,0.6d0]
    do ib=1,Nbath
       Hbath(:,:,:,:,:,:,ib)=-abs(impHloc)
       do i=1,Nlat
          Hbath(i,i,1,1,1,1,ib)=eps(ib)
       enddo
    enddo
    Vbath=1d0/sqrt(2d0)
    Uloc=0d0; Uloc(1)=2d0
    DimUp=924; DimDw=924
    call gpu_build_Hv_sector(Nlat,Norb,Nspin,Nbath,6,6,impHloc,Hbath,Vbath,Uloc,0d0,0d0,0d0,0d0,0d0,.true.,0,1)
    dim=gpu_vecDim_Hv_sector()
    nu=gpu_sector_nnz(1); nd=gpu_sector_nnz(2)
    allocate(v(dim),hv1(dim),hv2(dim),diag(dim),dg(dim),urp(DimUp+1),drp(DimDw+1),ucl(nu),dcl(nd),uv(nu),dv(nd))
    call gpu_get_sector_csr(1,urp,ucl,uv)
    call gpu_get_sector_csr(2,drp,dcl,dv)
    call gpu_get_sector_diag(diag)
    do i=1,dim
       v(i)=cmplx(sin(0.37d0*(i-1)+0.11d0),cos(0.23d0*(i-1)+0.05d0),8)
    enddo
    call gpuMatVec_main(dim,v,hv1)
    call gpu_delete_Hv_sector()
    dg=cmplx(diag,0d0,8)
    call gpu_build_Hv_sector_from_csr(DimUp,DimDw,urp,ucl,uv,drp,dcl,dv,dg,0,1)
    call gpuMatVec_main(dim,v,hv2)
    call gpu_delete_Hv_sector()
    write(*,"(A,2I10,A,ES12.4,A,ES12.4)")"stored matrices: nnz(H_up),nnz(H_dw)=",nu,nd," max|Hv(csr)-Hv(model)|=",maxval(abs(hv2-hv1))," max|Hv|=",maxval(abs(hv1))
    deallocate(v,hv1,hv2,diag,dg,urp,drp,ucl,dcl,uv,dv)
  end subroutine stored_matrices

  !> lowest eigenvalue of the Lanczos tridiagonal by bisection (Sturm count); blanc(1) unused
  function lowest_tridiag(a,b) result(e)
    real(8) :: a(:),b(:),e,lo,hi,mid,d
    integer :: n,k,it,cnt
    n=size(a)
    do while(n>1)          !trim trailing zeros left by an early exit
       if(b(n)/=0d0.or.a(n)/=0d0)exit
       n=n-1
    enddo
    lo=minval(a(1:n))-2d0*maxval(abs(b(1:n)))-1d0
    hi=maxval(a(1:n))+2d0*maxval(abs(b(1:n)))+1d0
    do it=1,200
       mid=0.5d0*(lo+hi)
       cnt=0
       d=a(1)-mid
       if(d<0d0)cnt=cnt+1
       do k=2,n
          if(d==0d0)d=1d-300
          d=a(k)-mid-b(k)*b(k)/d
          if(d<0d0)cnt=cnt+1
       enddo
       if(cnt>=1)then
          hi=mid
       else
          lo=mid
       endif
    enddo
    e=0.5d0*(lo+hi)
  end function lowest_tridiag

end program hxv_fortran_demo

!=======================================================================
! Drop-in replacement for the reference's boundary module (same module name and public
! entities as serial/ice_boundary.F90 and mpi/ice_boundary.F90: type ice_halo,
! ice_HaloCreate, the generic ice_HaloUpdate for 2-d/3-d/4-d R8/R4/I4 fields and
! ice_HaloExtrapolate), selected at link time like the reference's serial/ and mpi/
! variants.  Ghost cells are filled on the MI355X through libcice4_amd.so: on-rank copies
! by the halo kernel, off-rank rows by grouped RCCL send/recv over xGMI
! (include/cice4_amd.h: cice_domain_create, cice_halo_update_r8/i4).
!
! MPI builds (-DCICE4_AMD_MPI, with the reference's mpi/ice_communicate.F90): ice_HaloCreate also sets up
! the RCCL communicator (id broadcast over MPI_COMM_ICE); off-rank ghost cells then travel by RCCL.
!
! Scope: ghost width 1; cyclic / open / closed edges and the 'tripole' (U-fold) north boundary
! (serial/ice_boundary.F90:705-869: fieldLoc / fieldKind decide offsets and sign at the fold); ANY
! block distribution create_distribution produced, land-block elimination included -- the block->task
! map is handed to the device as it is, ghost cells facing an eliminated block take fillValue
! (mpi/ice_boundary.F90:5108-5111); 'tripoleT' (fold through T points) likewise.
! Ghost cells beyond an open or closed edge are left untouched, which is what the reference
! does for them (mpi/ice_boundary.F90: messages to a non-existent neighbour are never created).
!
! This generic entry point moves the host field to the device and back on every call: it
! is the functional drop-in for the model's set-up and state updates (ice_grid.F90,
! ice_state.F90:bound_state).  The dynamics do NOT come through here -- the EVP module
! keeps its fields on the device and updates their ghost cells inside its kernels.
!=======================================================================
module ice_boundary

   use iso_c_binding
   use ice_kinds_mod
   use ice_communicate, only: my_task
   use ice_fileunits, only: nu_diag
   use ice_domain_size, only: nx_global, ny_global, block_size_x, block_size_y
   use ice_blocks, only: nx_block, ny_block, nghost, nblocks_x, nblocks_y, nblocks_tot, block, get_block
   use ice_distribution, only: distrb, ice_distributionGet, ice_distributionGetBlockLoc, &
                               ice_distributionGetBlockID
   use ice_exit, only: abort_ice
   use cice4_amd_c

   implicit none
   private
   save

   type, public :: ice_halo
      integer (int_kind) :: communicator   ! kept for source compatibility (unused)
      integer (int_kind) :: numBlocks      ! local blocks the device domain was built for
      integer (int_kind) :: ewBnd, nsBnd   ! 0 open, 1 cyclic, 2 closed, 3 tripole, 4 tripoleT (north-south only)
   end type

   public :: ice_HaloCreate, ice_HaloUpdate, ice_HaloExtrapolate

   interface ice_HaloUpdate
      module procedure ice_HaloUpdate2DR8, ice_HaloUpdate2DR4, ice_HaloUpdate2DI4, &
                       ice_HaloUpdate3DR8, ice_HaloUpdate3DR4, ice_HaloUpdate3DI4, &
                       ice_HaloUpdate4DR8, ice_HaloUpdate4DR4, ice_HaloUpdate4DI4
   end interface

   interface ice_HaloExtrapolate
      module procedure ice_HaloExtrapolate2DR8
   end interface

contains

!=======================================================================
! Builds the device-side block topology and halo lists for the distribution `dist`
! and checks that the device's block->task map is the host's.
   function ice_HaloCreate(dist, nsBoundaryType, ewBoundaryType, nxGlobal) result(halo)
      type (distrb), intent(in) :: dist
      character (*), intent(in) :: nsBoundaryType, ewBoundaryType
      integer (int_kind), intent(in) :: nxGlobal
      type (ice_halo) :: halo

      integer (int_kind) :: nprocs, numBlocks, proc, lid, n, gid
      integer (c_int) :: info(9), binfo(10)
      integer (c_int), allocatable :: owner(:), local_id(:)

      if (nghost /= 1) call abort_ice('ice_HaloCreate: the GPU path needs nghost = 1')
      if (nxGlobal /= nx_global) call abort_ice('ice_HaloCreate: nxGlobal /= nx_global')
      halo%ewBnd = boundary_code(ewBoundaryType)
      halo%nsBnd = boundary_code(nsBoundaryType)
      if (halo%ewBnd >= 3) call abort_ice('ice_HaloCreate: tripole is a north-south boundary type')

      call ice_distributionGet(dist, nprocs=nprocs, communicator=halo%communicator, &
                               numLocalBlocks=numBlocks)
      ! the block -> task map exactly as create_distribution made it (cartesian, rake, space curve;
      ! proc = 0: eliminated land block), global block n = (jblock-1)*nblocks_x + iblock
      allocate(owner(nblocks_tot), local_id(nblocks_tot))
      do n = 1, nblocks_tot
         call ice_distributionGetBlockLoc(dist, n, proc, lid)
         owner(n) = proc - 1
         local_id(n) = lid - 1
      enddo

      ! one task = one GPU: task t takes device mod(t, visible devices)
      call cice_gpu_ensure(mod(my_task, max(1, cice_device_count())))
      call cice_gpu_check(cice_domain_create_map(cice_gpu_ctx, nx_global, ny_global, block_size_x, &
           block_size_y, halo%ewBnd, halo%nsBnd, my_task, nprocs, owner, local_id), 'cice_domain_create_map')
      deallocate(owner, local_id)
      call cice_gpu_check(cice_domain_info(cice_gpu_ctx, info), 'cice_domain_info')
      if (info(1) /= nx_block .or. info(2) /= ny_block .or. info(3) /= numBlocks) then
         write(nu_diag,*) 'ice_HaloCreate: device layout', info(1:3), ' host layout', &
                          nx_block, ny_block, numBlocks
         call abort_ice('ice_HaloCreate: device block layout differs from the host layout')
      endif
      do n = 1, numBlocks
         call ice_distributionGetBlockID(dist, n, gid)
         call cice_gpu_check(cice_domain_block(cice_gpu_ctx, n-1, binfo), 'cice_domain_block')
         if (binfo(7) + 1 /= gid) then
            write(nu_diag,*) 'ice_HaloCreate: local block', n, ' is global block', gid, &
                             ' on the host but', binfo(7) + 1, ' on the device'
            call abort_ice('ice_HaloCreate: device block order differs from the host order')
         endif
      enddo
      halo%numBlocks = numBlocks
      cice_gpu_domain_ready = .true.
#ifdef CICE4_AMD_MPI
      call cice_gpu_comm_setup(my_task, nprocs, halo%communicator)
#endif
   end function ice_HaloCreate

   integer (int_kind) function boundary_code(name)
      character (*), intent(in) :: name
      boundary_code = -1
      select case (trim(name))
      case ('open');   boundary_code = 0
      case ('cyclic'); boundary_code = 1
      case ('closed'); boundary_code = 2
      case ('tripole'); boundary_code = 3
      case ('tripoleT'); boundary_code = 4
      case default
         call abort_ice('ice_HaloCreate: boundary type not supported on the GPU path: '//trim(name))
      end select
   end function boundary_code

!=======================================================================
! Workers: the field goes to the device in the layout the caller has it in, (nx_block,ny_block,nz,max_blocks) with
! the task's blocks first; nz = 1 for 2-d fields, nz*nt for 4-d ones (cice_halo_update_blocked_*).
   subroutine update_r8(array, n1, n2, nz, nlast, halo, who, fieldLoc, fieldKind, fillValue)
      integer (int_kind), intent(in) :: n1, n2, nz, nlast, fieldLoc, fieldKind
      real (dbl_kind), intent(inout) :: array(*)
      type (ice_halo), intent(in) :: halo
      character (*), intent(in) :: who
      real (dbl_kind), intent(in), optional :: fillValue
      real (c_double) :: fill
      if (n1 /= nx_block .or. n2 /= ny_block) call abort_ice(who//': horizontal extent is not (nx_block,ny_block)')
      if (nlast < halo%numBlocks) call abort_ice(who//': fewer blocks in the array than on this task')
      if (nz < 1 .or. halo%numBlocks < 1) return
      fill = 0.0_c_double
      if (present(fillValue)) fill = fillValue
      call cice_gpu_check(cice_halo_update_blocked_r8(cice_gpu_ctx, array, nz, fieldLoc, fieldKind, fill), who)
   end subroutine update_r8

   subroutine update_r4(array, n1, n2, nz, nlast, halo, who, fieldLoc, fieldKind, fillValue)
      integer (int_kind), intent(in) :: n1, n2, nz, nlast, fieldLoc, fieldKind
      real (real_kind), intent(inout) :: array(*)
      type (ice_halo), intent(in) :: halo
      character (*), intent(in) :: who
      real (real_kind), intent(in), optional :: fillValue
      real (c_float) :: fill
      if (n1 /= nx_block .or. n2 /= ny_block) call abort_ice(who//': horizontal extent is not (nx_block,ny_block)')
      if (nlast < halo%numBlocks) call abort_ice(who//': fewer blocks in the array than on this task')
      if (nz < 1 .or. halo%numBlocks < 1) return
      fill = 0.0_c_float
      if (present(fillValue)) fill = fillValue
      call cice_gpu_check(cice_halo_update_blocked_r4(cice_gpu_ctx, array, nz, fieldLoc, fieldKind, fill), who)
   end subroutine update_r4

   subroutine update_i4(array, n1, n2, nz, nlast, halo, who, fieldLoc, fieldKind, fillValue)
      integer (int_kind), intent(in) :: n1, n2, nz, nlast, fieldLoc, fieldKind
      integer (int_kind), intent(inout) :: array(*)
      type (ice_halo), intent(in) :: halo
      character (*), intent(in) :: who
      integer (int_kind), intent(in), optional :: fillValue
      integer (c_int) :: fill
      if (n1 /= nx_block .or. n2 /= ny_block) call abort_ice(who//': horizontal extent is not (nx_block,ny_block)')
      if (nlast < halo%numBlocks) call abort_ice(who//': fewer blocks in the array than on this task')
      if (nz < 1 .or. halo%numBlocks < 1) return
      fill = 0
      if (present(fillValue)) fill = fillValue
      call cice_gpu_check(cice_halo_update_blocked_i4(cice_gpu_ctx, array, nz, fieldLoc, fieldKind, fill), who)
   end subroutine update_i4

!=======================================================================
! The nine specifics of the generic ice_HaloUpdate: 2-d (nx,ny,blocks), 3-d (nx,ny,nz,blocks), 4-d (nx,ny,nz,nt,blocks)
   subroutine ice_HaloUpdate2DR8(array, halo, fieldLoc, fieldKind, fillValue)
      real (dbl_kind), dimension(:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      real (dbl_kind), intent(in), optional :: fillValue
      call update_r8(array, size(array,1), size(array,2), 1, size(array,3), halo, 'ice_HaloUpdate2DR8', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate2DR8

   subroutine ice_HaloUpdate2DR4(array, halo, fieldLoc, fieldKind, fillValue)
      real (real_kind), dimension(:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      real (real_kind), intent(in), optional :: fillValue
      call update_r4(array, size(array,1), size(array,2), 1, size(array,3), halo, 'ice_HaloUpdate2DR4', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate2DR4

   subroutine ice_HaloUpdate2DI4(array, halo, fieldLoc, fieldKind, fillValue)
      integer (int_kind), dimension(:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      integer (int_kind), intent(in), optional :: fillValue
      call update_i4(array, size(array,1), size(array,2), 1, size(array,3), halo, 'ice_HaloUpdate2DI4', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate2DI4

   subroutine ice_HaloUpdate3DR8(array, halo, fieldLoc, fieldKind, fillValue)
      real (dbl_kind), dimension(:,:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      real (dbl_kind), intent(in), optional :: fillValue
      call update_r8(array, size(array,1), size(array,2), size(array,3), size(array,4), halo, 'ice_HaloUpdate3DR8', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate3DR8

   subroutine ice_HaloUpdate3DR4(array, halo, fieldLoc, fieldKind, fillValue)
      real (real_kind), dimension(:,:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      real (real_kind), intent(in), optional :: fillValue
      call update_r4(array, size(array,1), size(array,2), size(array,3), size(array,4), halo, 'ice_HaloUpdate3DR4', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate3DR4

   subroutine ice_HaloUpdate3DI4(array, halo, fieldLoc, fieldKind, fillValue)
      integer (int_kind), dimension(:,:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      integer (int_kind), intent(in), optional :: fillValue
      call update_i4(array, size(array,1), size(array,2), size(array,3), size(array,4), halo, 'ice_HaloUpdate3DI4', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate3DI4

   subroutine ice_HaloUpdate4DR8(array, halo, fieldLoc, fieldKind, fillValue)
      ! no `contiguous` here: bound_state passes the section trcrn(:,:,1:ntrcr,:,:) (ice_state.F90:206), and a
      ! contiguous dummy would make the compiler copy 25 planes in and out around every call.  Whole horizontal planes
      ! with strided levels / blocks go to the library as they lie; anything else through a contiguous copy.
      real (dbl_kind), dimension(:,:,:,:,:), intent(inout), target :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      real (dbl_kind), intent(in), optional :: fillValue
      real (dbl_kind), allocatable :: tmp(:,:,:,:,:)
      integer (c_long_long) :: s1, s2, sb
      integer (c_intptr_t) :: a0
      real (c_double) :: fill
      integer :: n1, n2, n3, n4, n5
      n1 = size(array,1); n2 = size(array,2); n3 = size(array,3); n4 = size(array,4); n5 = size(array,5)
      if (n1 /= nx_block .or. n2 /= ny_block) call abort_ice('ice_HaloUpdate4DR8: horizontal extent is not (nx_block,ny_block)')
      if (n5 < halo%numBlocks) call abort_ice('ice_HaloUpdate4DR8: fewer blocks in the array than on this task')
      if (n3*n4 < 1 .or. halo%numBlocks < 1) return
      if (is_contiguous(array)) then
         call update_r8(array, n1, n2, n3*n4, n5, halo, 'ice_HaloUpdate4DR8', fieldLoc, fieldKind, fillValue)
         return
      endif
      a0 = transfer(c_loc(array(1,1,1,1,1)), a0)
      if (transfer(c_loc(array(2,1,1,1,1)), a0) - a0 /= 8 .or. &
          transfer(c_loc(array(1,2,1,1,1)), a0) - a0 /= 8_c_intptr_t * n1) then   ! planes not whole: copy
         allocate(tmp(n1,n2,n3,n4,n5))
         tmp = array
         call update_r8(tmp, n1, n2, n3*n4, n5, halo, 'ice_HaloUpdate4DR8', fieldLoc, fieldKind, fillValue)
         array = tmp
         deallocate(tmp)
         return
      endif
      s1 = 0; s2 = 0; sb = 0
      if (n3 > 1) s1 = (transfer(c_loc(array(1,1,2,1,1)), a0) - a0) / 8
      if (n4 > 1) s2 = (transfer(c_loc(array(1,1,1,2,1)), a0) - a0) / 8
      if (n5 > 1) sb = (transfer(c_loc(array(1,1,1,1,2)), a0) - a0) / 8
      fill = 0.0_c_double
      if (present(fillValue)) fill = fillValue
      call cice_gpu_check(cice_halo_update_strided_r8(cice_gpu_ctx, c_loc(array(1,1,1,1,1)), n3, s1, n4, s2, sb, &
                          fieldLoc, fieldKind, fill), 'ice_HaloUpdate4DR8')
   end subroutine ice_HaloUpdate4DR8

   subroutine ice_HaloUpdate4DR4(array, halo, fieldLoc, fieldKind, fillValue)
      real (real_kind), dimension(:,:,:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      real (real_kind), intent(in), optional :: fillValue
      call update_r4(array, size(array,1), size(array,2), size(array,3)*size(array,4), size(array,5), halo, 'ice_HaloUpdate4DR4', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate4DR4

   subroutine ice_HaloUpdate4DI4(array, halo, fieldLoc, fieldKind, fillValue)
      integer (int_kind), dimension(:,:,:,:,:), intent(inout), contiguous :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldKind, fieldLoc
      integer (int_kind), intent(in), optional :: fillValue
      call update_i4(array, size(array,1), size(array,2), size(array,3)*size(array,4), size(array,5), halo, 'ice_HaloUpdate4DI4', &
                     fieldLoc, fieldKind, fillValue)
   end subroutine ice_HaloUpdate4DI4

!=======================================================================
! Ghost cells on non-cyclic domain edges by linear extrapolation from the two cells inside
! (used by ice_grid.F90 for HTN/HTE/dxhy/dyhx).  Host arithmetic: it runs a few times at
! set-up on grid arrays that live on the host.  West/east columns first, then south/north
! rows over the full width, so corner ghosts extrapolate the extrapolated columns.  The
! east/north target is the last column/row whose global index is non-zero, counted as the
! reference counts it (serial/ice_boundary.F90:4283-4310): padding has index 0 and so has the
! ghost line of a 'closed' edge, for which the target therefore is the last physical line.
   subroutine ice_HaloExtrapolate2DR8(ARRAY, dist, ew_bndy_type, ns_bndy_type)
      real (dbl_kind), dimension(:,:,:), intent(inout) :: ARRAY
      type (distrb), intent(in) :: dist
      character (char_len) :: ew_bndy_type, ns_bndy_type

      integer (int_kind) :: i, j, n, numBlocks, gid, ig, jg
      logical (log_kind) :: ew_edge, ns_edge
      type (block) :: blk
      real (dbl_kind), parameter :: two = 2.0_dbl_kind

      ew_edge = trim(ew_bndy_type) /= 'cyclic'
      ns_edge = trim(ns_bndy_type) /= 'cyclic'
      call ice_distributionGet(dist, numLocalBlocks=numBlocks)
      do n = 1, numBlocks
         call ice_distributionGetBlockID(dist, n, gid)
         blk = get_block(gid, gid)
         if (ew_edge .and. blk%iblock == 1) then
            do j = 1, ny_block
               ARRAY(1,j,n) = two*ARRAY(2,j,n) - ARRAY(3,j,n)
            enddo
         endif
         if (ew_edge .and. blk%iblock == nblocks_x) then
            ig = nx_block - count(blk%i_glob(nghost+1:nx_block) == 0)
            do j = 1, ny_block
               ARRAY(ig,j,n) = two*ARRAY(ig-1,j,n) - ARRAY(ig-2,j,n)
            enddo
         endif
         if (ns_edge .and. blk%jblock == 1) then
            do i = 1, nx_block
               ARRAY(i,1,n) = two*ARRAY(i,2,n) - ARRAY(i,3,n)
            enddo
         endif
         if (ns_edge .and. blk%jblock == nblocks_y .and. &
             trim(ns_bndy_type) /= 'tripole' .and. trim(ns_bndy_type) /= 'tripoleT') then
            jg = ny_block - count(blk%j_glob(nghost+1:ny_block) == 0)
            do i = 1, nx_block
               ARRAY(i,jg,n) = two*ARRAY(i,jg-1,n) - ARRAY(i,jg-2,n)
            enddo
         endif
      enddo
   end subroutine ice_HaloExtrapolate2DR8

end module ice_boundary

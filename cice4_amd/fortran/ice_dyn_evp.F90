!=======================================================================
! Drop-in replacement for the reference's source/ice_dyn_evp.F90.
!
! Same module name, same public entities (kdyn, ndte, evp_damping,
! yield_curve, the EVP scalars, fcor_blk, evp, init_evp,
! set_evp_parameters, principal_stress), so that ice_step_mod.F90:575
! (`call evp (dt)`), CICE_InitMod.F90:141 (`call init_evp`),
! ice_init.F90:91 (namelist) and ice_history.F90:1939
! (`principal_stress`) compile and link unchanged.  The work is done on
! the GPU by libcice4_amd.so through cice4_amd_c; nothing of the
! reference's implementation is kept here.
!
! Build: put this file in place of source/ice_dyn_evp.F90, add
! cice4_amd_c.F90 to the source list and link with -lcice4_amd.
! With -DAusCOM (bld/Macros.nci:56-57) the module has the coupled build's
! public entities as well -- the namelist variables cosw, sinw, dragio
! (ice_init.F90:97,156) -- writes the coupler's sicemass as the
! reference's evp does (:246-248), and links with -lcice4_amd_auscom.
!=======================================================================
      module ice_dyn_evp

      use ice_kinds_mod
      use ice_fileunits, only: nu_diag
#ifdef CICE4_AMD_MPI
      use ice_communicate, only: my_task, master_task, MPI_COMM_ICE, get_num_procs
#else
      use ice_communicate, only: my_task, master_task
#endif
      use ice_domain_size
      use ice_constants
      use iso_c_binding
      use cice4_amd_c

      implicit none
      save

      ! namelist parameters (ice_init.F90 reads them by these names)
      integer (kind=int_kind) :: kdyn, ndte
      logical (kind=log_kind) :: evp_damping
      character (len=char_len) :: yield_curve

#if defined(AusCOM)
      real (kind=dbl_kind), parameter :: &
         eyc = 0.36_dbl_kind, a_min = p001, m_min = p01
      real (kind=dbl_kind) :: dragio, cosw, sinw, dragw   ! namelist (dragw = dragio*rhow)
#else
      real (kind=dbl_kind), parameter :: &
         dragw = dragio * rhow, eyc = 0.36_dbl_kind, cosw = c1, sinw = c0, &
         a_min = p001, m_min = p01
#endif

      real (kind=dbl_kind) :: ecci, dtei, dte2T, denom1, denom2, rcon

      real (kind=dbl_kind), allocatable :: fcor_blk(:,:,:)
      logical, save, private :: fields_pinned = .false.
      logical, save, private :: stresses_on_device = .false.   ! CICE4_AMD_LAZY_STRESSES=1: evp_sync_stresses fetches them

      contains

!=======================================================================
      subroutine evp (dt)
      use ice_state
      use ice_flux
      use ice_timers
#if defined(AusCOM)
      use ice_domain, only: nblocks
      use ice_grid, only: tmask
      use cpl_parameters, only: use_ocnslope
      use cpl_arrays_setup, only: sicemass
#endif
      real (kind=dbl_kind), intent(in) :: dt
      type (cice_evp_fields) :: f
      character (len=8) :: env
      integer :: keep, ios
#if defined(AusCOM)
      integer (kind=int_kind) :: iblk
#endif

      call ice_timer_start(timer_dynamics)
#if defined(AusCOM)
      dragw = dragio * rhow
      call cice_gpu_check(cice_set_auscom(cice_gpu_ctx, cosw, sinw, dragio, &
                          merge(1_c_int, 0_c_int, use_ocnslope)), 'cice_set_auscom')
      ! the ice + snow mass handed to the ocean (:246-248: tmass of evp_prep1, :651-655)
      if (allocated(sicemass)) then
         do iblk = 1, nblocks
            sicemass(:,:,iblk) = merge(rhoi*vice(:,:,iblk) + rhos*vsno(:,:,iblk), c0, tmask(:,:,iblk))
         enddo
      endif
#endif
      f%aice = addr_r8(aice);  f%vice = addr_r8(vice);  f%vsno = addr_r8(vsno)
      f%aice0 = addr_r8(aice0); f%aicen = addr_r8(aicen); f%vicen = addr_r8(vicen)
      f%strairxT = addr_r8(strairxT); f%strairyT = addr_r8(strairyT)
      f%uocn = addr_r8(uocn); f%vocn = addr_r8(vocn)
      f%ss_tltx = addr_r8(ss_tltx); f%ss_tlty = addr_r8(ss_tlty)
      f%uvel = addr_r8(uvel); f%vvel = addr_r8(vvel)
      f%stressp_1 = addr_r8(stressp_1); f%stressp_2 = addr_r8(stressp_2)
      f%stressp_3 = addr_r8(stressp_3); f%stressp_4 = addr_r8(stressp_4)
      f%stressm_1 = addr_r8(stressm_1); f%stressm_2 = addr_r8(stressm_2)
      f%stressm_3 = addr_r8(stressm_3); f%stressm_4 = addr_r8(stressm_4)
      f%stress12_1 = addr_r8(stress12_1); f%stress12_2 = addr_r8(stress12_2)
      f%stress12_3 = addr_r8(stress12_3); f%stress12_4 = addr_r8(stress12_4)
      f%iceumask = addr_l4(iceumask)
      f%fm = addr_r8(fm); f%strtltx = addr_r8(strtltx); f%strtlty = addr_r8(strtlty)
      f%strocnx = addr_r8(strocnx); f%strocny = addr_r8(strocny)
      f%strintx = addr_r8(strintx); f%strinty = addr_r8(strinty)
      f%strairx = addr_r8(strairx); f%strairy = addr_r8(strairy)
      f%strength = addr_r8(strength); f%divu = addr_r8(divu); f%shear = addr_r8(shear)
      f%rdg_conv = addr_r8(rdg_conv); f%rdg_shear = addr_r8(rdg_shear)
      f%prs_sig = addr_r8(prs_sig)
      f%strocnxT = addr_r8(strocnxT); f%strocnyT = addr_r8(strocnyT)
      if (.not. fields_pinned) then   ! module arrays never move: page-lock them once (asynchronous DMA)
         call cice_gpu_check(cice_evp_pin_fields(cice_gpu_ctx, f), 'cice_evp_pin_fields')
         fields_pinned = .true.
         ! Two statements about the DRIVER, hence opt-in (include/cice4_amd.h, cice_evp):
         ! CICE4_AMD_KEEP_STATE=1: nothing but evp writes uvel, vvel, the stresses, iceumask between two steps (true of the
         !   reference; the restart reader runs before the first step) -- they are uploaded once;
         ! CICE4_AMD_KEEP_STATE=2: and init_history_dyn has zeroed fm, strtlt, strocn, strint before evp is called (true
         !   of CICE_RunMod.F90's ice_step) -- zeroed on the device instead of uploaded;
         ! CICE4_AMD_LAZY_STRESSES=1: the stresses stay on the device; whoever reads them on the host (ice_history's
         !   principal_stress, ice_restart's dumpfile) calls evp_sync_stresses first.
         call get_environment_variable('CICE4_AMD_KEEP_STATE', env)
         read(env, *, iostat=ios) keep
         if (ios == 0 .and. keep >= 1 .and. keep <= 2) then
            call cice_gpu_check(cice_evp_set_option(cice_gpu_ctx, 'keep_state'//c_null_char, int(keep, c_int)), 'keep_state')
            if (my_task == master_task) write(nu_diag,*) &
               'evp keeps uvel, vvel, the stresses and iceumask on the device (CICE4_AMD_KEEP_STATE)', keep
         endif
         call get_environment_variable('CICE4_AMD_LAZY_STRESSES', env)
         if (trim(env) == '1') then
            call cice_gpu_check(cice_evp_set_option(cice_gpu_ctx, 'lazy_stresses'//c_null_char, 1_c_int), 'lazy_stresses')
            stresses_on_device = .true.
            if (my_task == master_task) write(nu_diag,*) &
               'evp leaves the stresses on the device until evp_sync_stresses (CICE4_AMD_LAZY_STRESSES=1)'
         endif
      endif
      call cice_gpu_check(cice_evp(cice_gpu_ctx, dt, f), 'evp')
      call ice_timer_stop(timer_dynamics)
      end subroutine evp

!=======================================================================
! The 12 stresses of the device state into the module arrays (a no-op unless CICE4_AMD_LAZY_STRESSES=1): one line
! before `call principal_stress` in ice_history.F90:1939 and at the top of dumpfile (ice_restart.F90:74).
      subroutine evp_sync_stresses
      use ice_state
      use ice_flux
      type (cice_evp_fields) :: f
      if (.not. stresses_on_device) return
      f%stressp_1 = addr_r8(stressp_1); f%stressp_2 = addr_r8(stressp_2)
      f%stressp_3 = addr_r8(stressp_3); f%stressp_4 = addr_r8(stressp_4)
      f%stressm_1 = addr_r8(stressm_1); f%stressm_2 = addr_r8(stressm_2)
      f%stressm_3 = addr_r8(stressm_3); f%stressm_4 = addr_r8(stressm_4)
      f%stress12_1 = addr_r8(stress12_1); f%stress12_2 = addr_r8(stress12_2)
      f%stress12_3 = addr_r8(stress12_3); f%stress12_4 = addr_r8(stress12_4)
      call cice_gpu_check(cice_evp_download_stresses(cice_gpu_ctx, f), 'evp_sync_stresses')
      end subroutine evp_sync_stresses

!=======================================================================
      subroutine init_evp (dt)
      use ice_blocks, only: nx_block, ny_block
      use ice_domain, only: nblocks, ew_boundary_type, ns_boundary_type
      use ice_state
      use ice_flux
      use ice_grid, only: ULAT
      real (kind=dbl_kind), intent(in) :: dt
      integer (kind=int_kind) :: i, j, iblk

      call set_evp_parameters (dt)
      if (my_task == master_task) then
         write(nu_diag,*) 'dt  = ',dt
         write(nu_diag,*) 'dte = ',dt/real(ndte,kind=dbl_kind)
         write(nu_diag,*) 'tdamp =', eyc*dt
         write(nu_diag,*) 'EVP dynamics on the GPU (libcice4_amd)'
      endif
      if (.not. allocated(fcor_blk)) allocate(fcor_blk(nx_block,ny_block,max_blocks))
      do iblk = 1, nblocks
      do j = 1, ny_block
      do i = 1, nx_block
         uvel(i,j,iblk) = c0;  vvel(i,j,iblk) = c0
         divu(i,j,iblk) = c0;  shear(i,j,iblk) = c0
         rdg_conv(i,j,iblk) = c0;  rdg_shear(i,j,iblk) = c0
         fcor_blk(i,j,iblk) = c2*omega*sin(ULAT(i,j,iblk))
         stressp_1 (i,j,iblk) = c0; stressp_2 (i,j,iblk) = c0
         stressp_3 (i,j,iblk) = c0; stressp_4 (i,j,iblk) = c0
         stressm_1 (i,j,iblk) = c0; stressm_2 (i,j,iblk) = c0
         stressm_3 (i,j,iblk) = c0; stressm_4 (i,j,iblk) = c0
         stress12_1(i,j,iblk) = c0; stress12_2(i,j,iblk) = c0
         stress12_3(i,j,iblk) = c0; stress12_4(i,j,iblk) = c0
         iceumask(i,j,iblk) = .false.
      enddo
      enddo
      enddo

      call evp_gpu_setup
      end subroutine init_evp

!=======================================================================
! One-time device set-up: block topology, grid metrics, masks, EVP switches.
! (Serial build: one task.  An MPI build passes my_task and its process grid to
!  cice_domain_create, broadcasts cice_comm_unique_id from the master task and calls
!  cice_comm_init.)  Public so that a driver that changes the grid can call it again.
      subroutine evp_gpu_setup
      use ice_blocks, only: nx_block, ny_block
      use ice_domain, only: nblocks, ew_boundary_type, ns_boundary_type
      use ice_grid
      use ice_mechred, only: kstrength, krdg_partic, krdg_redist, mu_rdg
      type (cice_evp_grid) :: g
      type (cice_evp_config) :: cfg
      integer (c_int) :: info(9)

      call cice_gpu_ensure()
      ! ice_strength reads aicen, vicen with the library's category stride
      call cice_gpu_check(cice_check_sizes(cice_gpu_ctx, ncat, nilyr, nslyr, max_ntrcr), 'init_evp')
      ! with rccl/ice_boundary.F90 in the build the device topology (and, under MPI, the RCCL
      ! communicator) already exists for the model's block distribution; with the reference's own
      ! boundary module (serial build) it is created here
      if (.not. cice_gpu_domain_ready) then
         call cice_gpu_check(cice_domain_create(cice_gpu_ctx, nx_global, ny_global, block_size_x, &
              block_size_y, bnd_code(ew_boundary_type), bnd_code(ns_boundary_type), 0_c_int, &
              1_c_int, 1_c_int), 'cice_domain_create')
         cice_gpu_domain_ready = .true.     ! the transport module reuses it
      endif
      call cice_gpu_check(cice_domain_info(cice_gpu_ctx, info), 'cice_domain_info')
      if (info(1) /= nx_block .or. info(2) /= ny_block .or. info(3) /= nblocks .or. &
          nblocks > max_blocks) then
         write(nu_diag,*) 'init_evp: GPU block layout differs from the host layout', info(1:3), &
                          nx_block, ny_block, nblocks, max_blocks
         call cice_gpu_abort('init_evp: GPU block layout differs from the host layout')
      endif
      g%dxt = addr_r8(dxt); g%dyt = addr_r8(dyt); g%dxhy = addr_r8(dxhy); g%dyhx = addr_r8(dyhx)
      g%cxp = addr_r8(cxp); g%cyp = addr_r8(cyp); g%cxm = addr_r8(cxm); g%cym = addr_r8(cym)
      g%tarea = addr_r8(tarea); g%uarea = addr_r8(uarea)
      g%tarear = addr_r8(tarear); g%uarear = addr_r8(uarear); g%tinyarea = addr_r8(tinyarea)
      g%fcor = addr_r8(fcor_blk)
      g%tmask = addr_l4(tmask); g%umask = addr_l4(umask)
      g%HTN = addr_r8(HTN); g%HTE = addr_r8(HTE)
      cfg%ndte = ndte
      cfg%evp_damping = merge(1, 0, evp_damping)
      cfg%kstrength = kstrength; cfg%krdg_partic = krdg_partic; cfg%krdg_redist = krdg_redist
      cfg%mu_rdg = mu_rdg
      call cice_gpu_check(cice_evp_init(cice_gpu_ctx, cfg, g), 'cice_evp_init')
#ifdef CICE4_AMD_MPI
      ! one block per task (any cartesian layout): connect the neighbouring tasks' exchange buffers (the one-launch loop across tasks)
      call cice_gpu_peer_setup(my_task, get_num_procs(), MPI_COMM_ICE, nblocks)
#endif
      end subroutine evp_gpu_setup

      integer (c_int) function bnd_code(name)
      character (len=*), intent(in) :: name
      select case (trim(name))
      case ('cyclic'); bnd_code = 1
      case ('closed'); bnd_code = 2
      case ('open');   bnd_code = 0
      case ('tripole'); bnd_code = 3
      case ('tripoleT'); bnd_code = 4
      case default
         write(nu_diag,*) 'boundary type not supported on the GPU path: ', trim(name)
         call cice_gpu_abort('bnd_code: boundary type not supported on the GPU path')
      end select
      end function bnd_code

!=======================================================================
      subroutine set_evp_parameters (dt)
      real (kind=dbl_kind), intent(in) :: dt
      real (kind=dbl_kind) :: dte, ecc, tdamp2
      dte = dt/real(ndte,kind=dbl_kind)
      dtei = c1/dte
      ecc  = c4
      ecci = p25
      tdamp2 = c2*eyc*dt
      dte2T = dte/tdamp2
      denom1 = c1/(c1+dte2T)
      denom2 = c1/(c1+dte2T*ecc)
      rcon = 1230._dbl_kind*eyc*dt*dtei**2
      end subroutine set_evp_parameters

!=======================================================================
      subroutine principal_stress(nx_block, ny_block, stressp_1, stressm_1, stress12_1, &
                                  prs_sig, sig1, sig2)
      integer (kind=int_kind), intent(in) :: nx_block, ny_block
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(in) :: &
         stressp_1, stressm_1, stress12_1, prs_sig
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(out):: sig1, sig2
      integer (kind=int_kind) :: i, j
      real (kind=dbl_kind) :: root
      do j = 1, ny_block
      do i = 1, nx_block
         if (prs_sig(i,j) > puny) then
            root = sqrt(stressm_1(i,j)**2 + c4*stress12_1(i,j)**2)
            sig1(i,j) = (p5*(stressp_1(i,j) + root)) / prs_sig(i,j)
            sig2(i,j) = (p5*(stressp_1(i,j) - root)) / prs_sig(i,j)
         else
            sig1(i,j) = spval_dbl
            sig2(i,j) = spval_dbl
         endif
      enddo
      enddo
      end subroutine principal_stress

      end module ice_dyn_evp

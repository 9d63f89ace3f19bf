!=======================================================================
! Drop-in replacement for the reference's source/ice_transport_driver.F90.
!
! Same module name and the public entities other files use: `advection` (namelist, ice_init.F90:103,145),
! init_transport (CICE_InitMod.F90:152), transport_remap and transport_upwind (ice_step_mod.F90:581-585).
! The incremental remapping runs on the GPU (libcice4_amd.so: cice_transport_init / cice_transport_remap --
! state_to_tracers, horizontal_remap, tracers_to_state and bound_state in one call); nothing of the
! reference's implementation is kept here.  source/ice_transport_remap.F90 stays in the build unchanged
! (ice_step_mod and the drivers `use` it) but is no longer called.
! advection = 'upwind' runs on the GPU as well (cice_transport_upwind_init / cice_transport_upwind).
!=======================================================================
      module ice_transport_driver

      use ice_kinds_mod
      use ice_communicate, only: my_task, master_task
      use ice_domain_size
      use ice_constants
      use ice_fileunits, only: nu_diag
      use iso_c_binding
      use cice4_amd_c

      implicit none
      save

      character (len=char_len) :: advection   ! 'remap' or 'upwind' (ice_init.F90 reads it by this name)
      logical, private :: fields_pinned = .false., upwind_ready = .false., chain_asked = .false.

      contains

!=======================================================================
      subroutine init_transport
      use ice_state, only: ntrcr, trcr_depend
      use ice_grid, only: HTN, HTE, dxt, dyt, dxu, dyu, tarear, hm
      use ice_exit, only: abort_ice
      use ice_timers
      type (cice_transport_config) :: cfg
      type (cice_transport_grid) :: g

      call ice_timer_start(timer_advect)
      if (trim(advection) == 'remap') then
         if (.not. cice_gpu_domain_ready) &
            call abort_ice('init_transport: the device block topology does not exist yet (init_evp or ice_HaloCreate first)')
         ! the library's strides of the category / layer / tracer dimensions are compile-time sizes
         call cice_gpu_check(cice_check_sizes(cice_gpu_ctx, ncat, nilyr, nslyr, max_ntrcr), 'init_transport')
         if (ntrcr > size(cfg%trcr_depend)) &
            call abort_ice('init_transport: more tracers than the GPU transport module is built for')
         cfg%ntrcr = ntrcr
         cfg%trcr_depend = 0
         cfg%trcr_depend(1:ntrcr) = trcr_depend(1:ntrcr)
         g%HTN = addr_r8(HTN); g%HTE = addr_r8(HTE); g%dxt = addr_r8(dxt); g%dyt = addr_r8(dyt)
         g%dxu = addr_r8(dxu); g%dyu = addr_r8(dyu); g%tarear = addr_r8(tarear); g%hm = addr_r8(hm)
         call cice_gpu_check(cice_transport_init(cice_gpu_ctx, cfg, g), 'init_transport')
         if (my_task == master_task) write(nu_diag,*) 'Incremental remapping on the GPU (libcice4_amd)'
      endif

      call ice_timer_stop(timer_advect)
      end subroutine init_transport

!=======================================================================
      subroutine transport_remap (dt)
      use ice_state
      use ice_exit, only: abort_ice
      use ice_calendar, only: istep1
      use ice_timers
      real (kind=dbl_kind), intent(in) :: dt
      type (cice_transport_fields) :: f
      integer (c_int) :: l_stop, istop, jstop
      integer :: np
      character (len=8) :: chain_env

      call ice_timer_start(timer_advect)
      f%aice0 = addr_r8(aice0); f%aicen = addr_r8(aicen); f%trcrn = addr_r8(trcrn)
      f%vicen = addr_r8(vicen); f%vsnon = addr_r8(vsnon); f%eicen = addr_r8(eicen); f%esnon = addr_r8(esnon)
      f%uvel = addr_r8(uvel); f%vvel = addr_r8(vvel)
      if (.not. fields_pinned) then   ! module arrays never move: page-lock them once (asynchronous DMA)
         np = size(aice0)
         call cice_gpu_pin_r8(aice0, np); call cice_gpu_pin_r8(aicen, np*ncat)
         call cice_gpu_pin_r8(trcrn, np*ncat*max_ntrcr); call cice_gpu_pin_r8(vicen, np*ncat)
         call cice_gpu_pin_r8(vsnon, np*ncat); call cice_gpu_pin_r8(eicen, np*ntilyr)
         call cice_gpu_pin_r8(esnon, np*ntslyr)
         fields_pinned = .true.
      endif
      if (.not. chain_asked) then
         ! evp -> transport without a PCIe round trip (include/cice4_amd.h: cice_transport_chain): step_dynamics calls
         ! evp(dt) and transport_remap(dt) back to back (ice_step_mod.F90:575-584), nothing writes the state in between.
         ! A statement about the DRIVER, hence opt-in: CICE4_AMD_CHAIN=1 in the environment.
         chain_asked = .true.
         call get_environment_variable('CICE4_AMD_CHAIN', chain_env)
         if (trim(chain_env) == '1') then
            call cice_gpu_check(cice_transport_chain(cice_gpu_ctx, f), 'cice_transport_chain')
            if (my_task == master_task) write(nu_diag,*) &
               'transport_remap takes its state from the device after evp (CICE4_AMD_CHAIN=1)'
         endif
      endif
      call cice_gpu_check(cice_transport_remap(cice_gpu_ctx, dt, f, l_stop, istop, jstop), 'transport_remap')
      if (l_stop /= 0) then
         write (nu_diag,*) 'istep1, my_task =', istep1, my_task
         write (nu_diag,*) 'transport_remap (GPU): local i and j:', istop, jstop
         if (l_stop == 1) call abort_ice('remap transport: bad departure points')
         call abort_ice('ice remap_transport: negative area')
      endif
      call ice_timer_stop(timer_advect)
      end subroutine transport_remap

!=======================================================================
      subroutine transport_upwind (dt)
      use ice_state
      use ice_grid, only: HTE, HTN, tarea
      use ice_exit, only: abort_ice
      use ice_timers
      real (kind=dbl_kind), intent(in) :: dt
      type (cice_transport_fields) :: f
      type (cice_transport_config) :: cfg
      integer :: np

      call ice_timer_start(timer_advect)
      if (.not. upwind_ready) then   ! (init_transport does nothing for this scheme in the reference, :81-170)
         if (.not. cice_gpu_domain_ready) &
            call abort_ice('transport_upwind: the device block topology does not exist yet (init_evp or ice_HaloCreate first)')
         call cice_gpu_check(cice_check_sizes(cice_gpu_ctx, ncat, nilyr, nslyr, max_ntrcr), 'transport_upwind')
         if (ntrcr > size(cfg%trcr_depend)) &
            call abort_ice('transport_upwind: more tracers than the GPU transport module is built for')
         cfg%ntrcr = ntrcr
         cfg%trcr_depend = 0
         cfg%trcr_depend(1:ntrcr) = trcr_depend(1:ntrcr)
         call cice_gpu_check(cice_transport_upwind_init(cice_gpu_ctx, cfg, nt_Tsfc, addr_r8(HTE), addr_r8(HTN), &
                             addr_r8(tarea)), 'transport_upwind')
         upwind_ready = .true.
      endif
      f%aice0 = addr_r8(aice0); f%aicen = addr_r8(aicen); f%trcrn = addr_r8(trcrn)
      f%vicen = addr_r8(vicen); f%vsnon = addr_r8(vsnon); f%eicen = addr_r8(eicen); f%esnon = addr_r8(esnon)
      f%uvel = addr_r8(uvel); f%vvel = addr_r8(vvel)
      if (.not. fields_pinned) then
         np = size(aice0)
         call cice_gpu_pin_r8(aice0, np); call cice_gpu_pin_r8(aicen, np*ncat)
         call cice_gpu_pin_r8(trcrn, np*ncat*max_ntrcr); call cice_gpu_pin_r8(vicen, np*ncat)
         call cice_gpu_pin_r8(vsnon, np*ncat); call cice_gpu_pin_r8(eicen, np*ntilyr)
         call cice_gpu_pin_r8(esnon, np*ntslyr)
         fields_pinned = .true.
      endif
      call cice_gpu_check(cice_transport_upwind(cice_gpu_ctx, dt, f), 'transport_upwind')
      call ice_timer_stop(timer_advect)
      end subroutine transport_upwind

      end module ice_transport_driver

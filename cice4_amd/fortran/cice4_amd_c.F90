!=======================================================================
! ISO_C_BINDING shim: Fortran interfaces to libcice4_amd.so
! (include/cice4_amd.h).  Host code stays Fortran; these are the only
! declarations a CICE4 build needs in order to call the MI355X hot path.
! The derived types mirror the C structs member for member.
!=======================================================================
module cice4_amd_c
   use iso_c_binding
   implicit none
   public

   integer(c_int), parameter :: CICE_OK = 0

   type, bind(C) :: cice_evp_grid      ! source/ice_grid.F90:58-133 arrays
      type(c_ptr) :: dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, tarea, uarea, tarear, &
                     uarear, tinyarea, fcor, tmask, umask
      type(c_ptr) :: HTN = c_null_ptr, HTE = c_null_ptr   ! optional
   end type

   type, bind(C) :: cice_evp_config    ! ice_dyn_evp.F90:64-74, ice_mechred.F90:64-79
      integer(c_int) :: ndte, evp_damping, kstrength, krdg_partic, krdg_redist
      real(c_double) :: mu_rdg
   end type

   type, bind(C) :: cice_evp_fields    ! ice_state.F90:55-148, ice_flux.F90:42-99
      type(c_ptr) :: aice, vice, vsno, aice0, aicen, vicen, strairxT, strairyT, uocn, vocn, &
                     ss_tltx, ss_tlty
      type(c_ptr) :: uvel, vvel
      type(c_ptr) :: stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, &
                     stressm_3, stressm_4, stress12_1, stress12_2, stress12_3, stress12_4
      type(c_ptr) :: iceumask
      type(c_ptr) :: fm, strtltx, strtlty, strocnx, strocny, strintx, strinty
      type(c_ptr) :: strairx, strairy, strength, divu, shear, rdg_conv, rdg_shear, prs_sig, &
                     strocnxT, strocnyT
   end type

   type, bind(C) :: cice_thermo_config ! ice_therm_vertical.F90:56-79
      integer(c_int) :: heat_capacity, calc_Tsfc, conduct
      real(c_double) :: ustar_min
      integer(c_int) :: tr_iage, nt_Tsfc, nt_iage
   end type

   ! horizontal transport (cice_transport_init / cice_transport_remap)
   type, bind(C) :: cice_transport_config
      integer(c_int) :: ntrcr
      integer(c_int) :: trcr_depend(5)
   end type

   type, bind(C) :: cice_transport_grid
      type(c_ptr) :: HTN, HTE, dxt, dyt, dxu, dyu, tarear, hm
   end type

   type, bind(C) :: cice_transport_fields
      type(c_ptr) :: aice0, aicen, trcrn, vicen, vsnon, eicen, esnon, uvel, vvel
   end type

   ! the thermodynamic half-step in one call (cice_step_therm1): c_ptr to module arrays of shape
   ! (nx_block,ny_block[,k],[ncat,]max_blocks) used with nblocks = max_blocks; c_null_ptr = not wanted
   type, bind(C) :: cice_thermo_fields
      type(c_ptr) :: aicen, trcrn, vicen, vsnon, eicen, esnon
      type(c_ptr) :: flw, potT, Qa, rhoa, fsnow, fbot, Tbot
      type(c_ptr) :: lhcoef, shcoef
      type(c_ptr) :: fswsfc, fswint, fswthrun, Sswabs, Iswabs
      type(c_ptr) :: fsurfn, fcondtopn, fsensn, flatn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, &
                     meltt, melts, meltb, congel, snoice
      type(c_ptr) :: mlt_onset, frz_onset
   end type

   type, bind(C) :: cice_frzmlt_fields
      type(c_ptr) :: aice, frzmlt, sst, Tf, strocnxT, strocnyT
      type(c_ptr) :: Tbot = c_null_ptr, fbot = c_null_ptr, rside = c_null_ptr
   end type

   type, bind(C) :: cice_merge_fields
      type(c_ptr) :: aicen_init = c_null_ptr, strairxn, strairyn, Trefn, Qrefn
      type(c_ptr) :: acc(20)
   end type

   ! atmo_boundary_layer on the device inside the one-call half-step (cice_step_therm1_abl)
   type, bind(C) :: cice_atmo_fields
      type(c_ptr) :: uatm, vatm, wind, zlvl
      type(c_ptr) :: strax = c_null_ptr, stray = c_null_ptr
      integer(c_int) :: calc_strair = 1
      type(c_ptr) :: strairxn = c_null_ptr, strairyn = c_null_ptr, Trefn = c_null_ptr, Qrefn = c_null_ptr, &
                     lhcoef = c_null_ptr, shcoef = c_null_ptr
   end type

   interface
      integer(c_int) function cice_step_therm1_abl(ctx, dt, yday, st, fz, mg, atm, n_updates, l_stop, istop, &
            jstop, nstop, bstop) bind(C, name='cice_step_therm1_abl')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: dt, yday
         type(cice_thermo_fields), intent(in) :: st
         type(cice_frzmlt_fields), intent(in) :: fz
         type(cice_merge_fields), intent(in) :: mg
         type(cice_atmo_fields), intent(in) :: atm
         integer(c_long_long), intent(out) :: n_updates
         integer(c_int), intent(out) :: l_stop, istop, jstop, nstop, bstop
      end function
      ! atmo_boundary_layer (source/ice_atmo.F90:56), one block; sfctype 0 'ice', 1 'ocn'
      integer(c_int) function cice_atmo_boundary_layer(ctx, nx_block, ny_block, sfctype, icells, indxi, indxj, &
            Tsf, potT, uatm, vatm, wind, zlvl, Qa, rhoa, calc_strair, strx, stry, Tref, Qref, delt, delq, &
            lhcoef, shcoef) bind(C, name='cice_atmo_boundary_layer')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: nx_block, ny_block, sfctype, icells, calc_strair
         integer(c_int), intent(in) :: indxi(*), indxj(*)
         real(c_double), intent(in) :: Tsf(*), potT(*), uatm(*), vatm(*), wind(*), zlvl(*), Qa(*), rhoa(*)
         real(c_double), intent(inout) :: strx(*), stry(*)
         real(c_double), intent(out) :: Tref(*), Qref(*), delt(*), delq(*), lhcoef(*), shcoef(*)
      end function
      integer(c_int) function cice_step_therm1(ctx, dt, yday, st, fz, mg, n_updates, l_stop, istop, jstop, &
            nstop, bstop) bind(C, name='cice_step_therm1')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: dt, yday
         type(cice_thermo_fields), intent(in) :: st
         type(cice_frzmlt_fields), intent(in) :: fz
         type(cice_merge_fields), intent(in) :: mg
         integer(c_long_long), intent(out) :: n_updates
         integer(c_int), intent(out) :: l_stop, istop, jstop, nstop, bstop
      end function
      integer(c_int) function cice_transport_init(ctx, cfg, grid) bind(C, name='cice_transport_init')
         import
         type(c_ptr), value :: ctx
         type(cice_transport_config), intent(in) :: cfg
         type(cice_transport_grid), intent(in) :: grid
      end function
      integer(c_int) function cice_transport_upwind_init(ctx, cfg, nt_Tsfc, HTE, HTN, tarea) &
            bind(C, name='cice_transport_upwind_init')
         import
         type(c_ptr), value :: ctx
         type(cice_transport_config), intent(in) :: cfg
         integer(c_int), value :: nt_Tsfc
         type(c_ptr), value :: HTE, HTN, tarea
      end function
      integer(c_int) function cice_transport_upwind(ctx, dt, f) bind(C, name='cice_transport_upwind')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: dt
         type(cice_transport_fields), intent(in) :: f
      end function
      integer(c_int) function cice_transport_chain(ctx, f) bind(C, name='cice_transport_chain')
         import
         type(c_ptr), value :: ctx
         type(cice_transport_fields), intent(in) :: f
      end function
      integer(c_int) function cice_transport_remap(ctx, dt, f, l_stop, istop, jstop) &
            bind(C, name='cice_transport_remap')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: dt
         type(cice_transport_fields), intent(in) :: f
         integer(c_int), intent(out) :: l_stop, istop, jstop
      end function
      integer(c_int) function cice_thermo_batch_alloc(ctx, nx_block, ny_block, nblocks) &
            bind(C, name='cice_thermo_batch_alloc')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: nx_block, ny_block, nblocks
      end function
      integer(c_int) function cice_create(ctx, device) bind(C, name='cice_create')
         import
         type(c_ptr), intent(out) :: ctx
         integer(c_int), value :: device
      end function
      integer(c_int) function cice_device_count() bind(C, name='cice_device_count')
         import
      end function
      integer(c_int) function cice_host_register(ctx, host, bytes) bind(C, name='cice_host_register')
         import
         type(c_ptr), value :: ctx, host
         integer(c_size_t), value :: bytes
      end function
      integer(c_int) function cice_destroy(ctx) bind(C, name='cice_destroy')
         import
         type(c_ptr), value :: ctx
      end function
      type(c_ptr) function cice_last_error(ctx) bind(C, name='cice_last_error')
         import
         type(c_ptr), value :: ctx
      end function
      integer(c_int) function cice_domain_create(ctx, nx_global, ny_global, block_size_x, &
            block_size_y, ew_boundary, ns_boundary, rank, npx, npy) bind(C, name='cice_domain_create')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: nx_global, ny_global, block_size_x, block_size_y, ew_boundary, &
                                  ns_boundary, rank, npx, npy
      end function
      integer(c_int) function cice_domain_create_map(ctx, nx_global, ny_global, block_size_x, &
            block_size_y, ew_boundary, ns_boundary, rank, nranks, owner, local_id) &
            bind(C, name='cice_domain_create_map')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: nx_global, ny_global, block_size_x, block_size_y, ew_boundary, &
                                  ns_boundary, rank, nranks
         integer(c_int), intent(in) :: owner(*), local_id(*)
      end function
      integer(c_int) function cice_domain_info(ctx, info) bind(C, name='cice_domain_info')
         import
         type(c_ptr), value :: ctx
         integer(c_int), intent(out) :: info(9)
      end function
      integer(c_int) function cice_domain_block(ctx, local_block, info) bind(C, name='cice_domain_block')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: local_block
         integer(c_int), intent(out) :: info(10)
      end function
      integer(c_int) function cice_halo_update_r8(ctx, field, nlev) bind(C, name='cice_halo_update_r8')
         import
         type(c_ptr), value :: ctx
         real(c_double) :: field(*)
         integer(c_int), value :: nlev
      end function
      integer(c_int) function cice_halo_update_i4(ctx, field, nlev) bind(C, name='cice_halo_update_i4')
         import
         type(c_ptr), value :: ctx
         integer(c_int) :: field(*)
         integer(c_int), value :: nlev
      end function
      integer(c_int) function cice_halo_update_ex_r8(ctx, field, nlev, loc, kind, fill) &
            bind(C, name='cice_halo_update_ex_r8')
         import
         type(c_ptr), value :: ctx
         real(c_double) :: field(*)
         integer(c_int), value :: nlev, loc, kind
         real(c_double), value :: fill
      end function
      integer(c_int) function cice_halo_update_ex_r4(ctx, field, nlev, loc, kind, fill) &
            bind(C, name='cice_halo_update_ex_r4')
         import
         type(c_ptr), value :: ctx
         real(c_float) :: field(*)
         integer(c_int), value :: nlev, loc, kind
         real(c_float), value :: fill
      end function
      integer(c_int) function cice_halo_update_ex_i4(ctx, field, nlev, loc, kind, fill) &
            bind(C, name='cice_halo_update_ex_i4')
         import
         type(c_ptr), value :: ctx
         integer(c_int) :: field(*)
         integer(c_int), value :: nlev, loc, kind, fill
      end function
      integer(c_int) function cice_halo_update_blocked_r8(ctx, field, nz, loc, kind, fill) &
            bind(C, name='cice_halo_update_blocked_r8')
         import
         type(c_ptr), value :: ctx
         real(c_double) :: field(*)
         integer(c_int), value :: nz, loc, kind
         real(c_double), value :: fill
      end function
      integer(c_int) function cice_halo_update_blocked_r4(ctx, field, nz, loc, kind, fill) &
            bind(C, name='cice_halo_update_blocked_r4')
         import
         type(c_ptr), value :: ctx
         real(c_float) :: field(*)
         integer(c_int), value :: nz, loc, kind
         real(c_float), value :: fill
      end function
      integer(c_int) function cice_halo_update_blocked_i4(ctx, field, nz, loc, kind, fill) &
            bind(C, name='cice_halo_update_blocked_i4')
         import
         type(c_ptr), value :: ctx
         integer(c_int) :: field(*)
         integer(c_int), value :: nz, loc, kind, fill
      end function
      integer(c_int) function cice_halo_update_strided_r8(ctx, field, nz1, stride1, nz2, stride2, stride_block, &
            loc, kind, fill) bind(C, name='cice_halo_update_strided_r8')
         import
         type(c_ptr), value :: ctx, field
         integer(c_int), value :: nz1, nz2, loc, kind
         integer(c_long_long), value :: stride1, stride2, stride_block
         real(c_double), value :: fill
      end function
      integer(c_int) function cice_comm_unique_id(uid) bind(C, name='cice_comm_unique_id')
         import
         character(kind=c_char), intent(out) :: uid(128)
      end function
      integer(c_int) function cice_comm_init(ctx, uid, rank, nranks) bind(C, name='cice_comm_init')
         import
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: uid(128)
         integer(c_int), value :: rank, nranks
      end function
      integer(c_int) function cice_check_sizes(ctx, ncat, nilyr, nslyr, max_ntrcr) bind(C, name='cice_check_sizes')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: ncat, nilyr, nslyr, max_ntrcr
      end function
      ! libcice4_amd_auscom.so only (the library for a reference built -DAusCOM -Dcoupled)
      integer(c_int) function cice_set_auscom(ctx, cosw, sinw, dragio, use_ocnslope) bind(C, name='cice_set_auscom')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: cosw, sinw, dragio
         integer(c_int), value :: use_ocnslope
      end function
      integer(c_int) function cice_thermo_set_chio(ctx, chio) bind(C, name='cice_thermo_set_chio')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: chio
      end function
      integer(c_int) function cice_comm_init_shm(ctx, name, rank, nranks, box_bytes) bind(C, name='cice_comm_init_shm')
         import
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: name(*)
         integer(c_int), value :: rank, nranks
         integer(c_long_long), value :: box_bytes
      end function
      integer(c_int) function cice_getpid() bind(C, name='getpid')
         import
      end function
      integer(c_int) function cice_comm_init_local(ctx, link_id, rank, nranks) bind(C, name='cice_comm_init_local')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: link_id, rank, nranks
      end function
      ! the one-launch subcycle loop across tasks: exchange copies / progress words of the neighbours (DESIGN.md section 7)
      integer(c_int) function cice_evp_set_option(ctx, key, value) bind(C, name='cice_evp_set_option')
         import
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: key(*)
         integer(c_int), value :: value
      end function
      integer(c_int) function cice_evp_get_info(ctx, key, value) bind(C, name='cice_evp_get_info')
         import
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: key(*)
         integer(c_int), intent(out) :: value
      end function
      integer(c_int) function cice_evp_peer_export_ipc(ctx, handles, plane) bind(C, name='cice_evp_peer_export_ipc')
         import
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(out) :: handles(64,3)
         integer(c_long_long), intent(out) :: plane
      end function
      integer(c_int) function cice_evp_peer_connect_ipc(ctx, side, handles, plane) bind(C, name='cice_evp_peer_connect_ipc')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: side
         character(kind=c_char), intent(in) :: handles(64,3)
         integer(c_long_long), value :: plane
      end function
      integer(c_int) function cice_evp_peer_ranks(ctx, n, ranks) bind(C, name='cice_evp_peer_ranks')
         import
         type(c_ptr), value :: ctx
         integer(c_int), intent(out) :: n
         integer(c_int), intent(out) :: ranks(8)
      end function
      integer(c_int) function cice_evp_peer_connect_rank_ipc(ctx, rank, handles, plane) bind(C, name='cice_evp_peer_connect_rank_ipc')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: rank
         character(kind=c_char), intent(in) :: handles(64,3)
         integer(c_long_long), value :: plane
      end function
      integer(c_int) function cice_evp_peer_export(ctx, bufs, plane) bind(C, name='cice_evp_peer_export')
         import
         type(c_ptr), value :: ctx
         type(c_ptr), intent(out) :: bufs(3)
         integer(c_long_long), intent(out) :: plane
      end function
      integer(c_int) function cice_evp_peer_connect(ctx, side, xu0, xu1, rprog, plane) bind(C, name='cice_evp_peer_connect')
         import
         type(c_ptr), value :: ctx, xu0, xu1, rprog
         integer(c_int), value :: side
         integer(c_long_long), value :: plane
      end function
      ! the thermodynamic state on the device becomes the dynamics' input there (SURVEY section 8 f1)
      integer(c_int) function cice_evp_adopt_thermo_state(ctx) bind(C, name='cice_evp_adopt_thermo_state')
         import
         type(c_ptr), value :: ctx
      end function
      integer(c_int) function cice_thermo_set_option(ctx, key, value) bind(C, name='cice_thermo_set_option')
         import
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: key(*)
         integer(c_int), value :: value
      end function
      integer(c_int) function cice_comm_count(ctx, nranks) bind(C, name='cice_comm_count')
         import
         type(c_ptr), value :: ctx
         integer(c_int), intent(out) :: nranks
      end function
      integer(c_int) function cice_evp_init(ctx, cfg, grid) bind(C, name='cice_evp_init')
         import
         type(c_ptr), value :: ctx
         type(cice_evp_config), intent(in) :: cfg
         type(cice_evp_grid), intent(in) :: grid
      end function
      integer(c_int) function cice_evp(ctx, dt, f) bind(C, name='cice_evp')
         import
         type(c_ptr), value :: ctx
         real(c_double), value :: dt
         type(cice_evp_fields), intent(in) :: f
      end function
      integer(c_int) function cice_evp_pin_fields(ctx, f) bind(C, name='cice_evp_pin_fields')
         import
         type(c_ptr), value :: ctx
         type(cice_evp_fields), intent(in) :: f
      end function
      integer(c_int) function cice_evp_download_stresses(ctx, f) bind(C, name='cice_evp_download_stresses')
         import
         type(c_ptr), value :: ctx
         type(cice_evp_fields), intent(in) :: f
      end function
      integer(c_int) function cice_thermo_init(ctx, cfg, salin, Tmlt) bind(C, name='cice_thermo_init')
         import
         type(c_ptr), value :: ctx
         type(cice_thermo_config), intent(in) :: cfg
         real(c_double), intent(out) :: salin(*), Tmlt(*)
      end function
      integer(c_int) function cice_thermo_vertical(ctx, nx_block, ny_block, dt, icells, indxi, &
            indxj, aicen, trcrn, vicen, vsnon, eicen, esnon, flw, potT, Qa, rhoa, fsnow, fbot, &
            Tbot, lhcoef, shcoef, fswsfc, fswint, fswthrun, Sswabs, Iswabs, fsurfn, fcondtopn, &
            fsensn, flatn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, &
            congel, snoice, mlt_onset, frz_onset, yday, l_stop, istop, jstop) &
            bind(C, name='cice_thermo_vertical')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: nx_block, ny_block, icells
         real(c_double), value :: dt, yday
         integer(c_int), intent(in) :: indxi(*), indxj(*)
         real(c_double) :: aicen(*), trcrn(*), vicen(*), vsnon(*), eicen(*), esnon(*)
         real(c_double), intent(in) :: flw(*), potT(*), Qa(*), rhoa(*), fsnow(*), fbot(*), Tbot(*), &
                                       lhcoef(*), shcoef(*)
         real(c_double) :: fswsfc(*), fswint(*), fswthrun(*), Sswabs(*), Iswabs(*), fsurfn(*), &
                           fcondtopn(*), fsensn(*), flatn(*), fswabsn(*), flwoutn(*), evapn(*), &
                           freshn(*), fsaltn(*), fhocnn(*), meltt(*), melts(*), meltb(*), congel(*), &
                           snoice(*), mlt_onset(*), frz_onset(*)
         integer(c_int), intent(out) :: l_stop, istop, jstop
      end function
      integer(c_int) function cice_frzmlt_bottom_lateral(ctx, nx_block, ny_block, ilo, ihi, jlo, jhi, &
            dt, aice, frzmlt, eicen, esnon, sst, Tf, strocnxT, strocnyT, Tbot, fbot, rside) &
            bind(C, name='cice_frzmlt_bottom_lateral')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: nx_block, ny_block, ilo, ihi, jlo, jhi
         real(c_double), value :: dt
         real(c_double), intent(in) :: aice(*), frzmlt(*), eicen(*), esnon(*), sst(*), Tf(*), &
                                       strocnxT(*), strocnyT(*)
         real(c_double), intent(out) :: Tbot(*), fbot(*), rside(*)
      end function
   end interface

   ! one context per MPI task (= per GPU), shared by the drop-in modules
   type(c_ptr), save :: cice_gpu_ctx = c_null_ptr
   ! set by the boundary module (rccl/ice_boundary.F90) once it has built the device topology for the
   ! model's own block distribution; the dynamics module then reuses it instead of building a serial one
   logical, save :: cice_gpu_domain_ready = .false.

contains

   ! address of a (contiguous) array that is not declared TARGET in its home module
   function addr_r8(a) result(p)
      real(c_double), target, intent(in) :: a(*)
      type(c_ptr) :: p
      p = c_loc(a)
   end function addr_r8

   function addr_i4(a) result(p)
      integer(c_int), target, intent(in) :: a(*)
      type(c_ptr) :: p
      p = c_loc(a)
   end function addr_i4

   function addr_l4(a) result(p)       ! default logical is 4 bytes (ice_kinds_mod.F90:31)
      logical(c_int), target, intent(in) :: a(*)
      type(c_ptr) :: p
      p = c_loc(a)
   end function addr_l4

   ! page-lock a module array (its address does not change during the run): asynchronous DMA afterwards
   subroutine cice_gpu_pin_r8(a, n)
      real(c_double), target, intent(in) :: a(*)
      integer, intent(in) :: n
      integer(c_int) :: rc
      rc = cice_host_register(cice_gpu_ctx, c_loc(a), int(n, c_size_t) * 8_c_size_t)
   end subroutine cice_gpu_pin_r8

   ! A failed library call ends the run.  In an MPI job it has to end the JOB: the reference aborts through abort_ice ->
   ! MPI_ABORT (mpi/ice_exit.F90:41-80); an `error stop` of one task alone would leave the others waiting in their next
   ! exchange.  (The serial build has no MPI: error stop.)
   subroutine cice_gpu_check(rc, where)
      integer(c_int), intent(in) :: rc
      character(len=*), intent(in) :: where
      character(kind=c_char), pointer :: msg(:)
      integer :: n
      if (rc == CICE_OK) return
      write(*,*) 'cice4_amd error ', rc, ' in ', where
      call c_f_pointer(cice_last_error(cice_gpu_ctx), msg, [512])
      n = 1
      do while (n < 512 .and. msg(n) /= c_null_char)
         n = n + 1
      enddo
      write(*,*) msg(1:n-1)
      call cice_gpu_abort('cice4_amd')
   end subroutine cice_gpu_check

   subroutine cice_gpu_abort(what)
      character(len=*), intent(in) :: what
#ifdef CICE4_AMD_MPI
      include 'mpif.h'
      integer :: ierr
      logical :: up
      write(0,*) 'cice4_amd: aborting the MPI job: ', what
      flush(6)
      flush(0)
      call MPI_INITIALIZED(up, ierr)
      if (up) call MPI_ABORT(MPI_COMM_WORLD, 1, ierr)
#endif
      error stop 'cice4_amd'
   end subroutine cice_gpu_abort

   subroutine cice_gpu_ensure(device)
      integer(c_int), intent(in), optional :: device
      integer(c_int) :: dev
      if (c_associated(cice_gpu_ctx)) return
      dev = -1
      if (present(device)) dev = device
      call cice_gpu_check(cice_create(cice_gpu_ctx, dev), 'cice_create')
   end subroutine cice_gpu_ensure

#ifdef CICE4_AMD_MPI
   ! MPI builds: the control plane stays MPI (the RCCL id travels by MPI_BCAST on the model's own
   ! communicator), the data plane of the halo exchange becomes RCCL over xGMI.  Called by every task
   ! right after cice_domain_create.
   subroutine cice_gpu_comm_setup(my_task, nprocs, comm)
      integer, intent(in) :: my_task, nprocs, comm
      include 'mpif.h'
      character(kind=c_char) :: uid(128)
      integer :: ierr, pid0, k
      character(len=32) :: link
      character(len=48) :: txt
      character(kind=c_char) :: cname(49)
      ! CICE4_AMD_LINK=shm: the tasks are processes of ONE host and talk through a file under /dev/shm instead of RCCL --
      ! the way to run an MPI job of the model on a box with fewer GPUs than tasks (RCCL refuses two ranks on one device)
      call get_environment_variable('CICE4_AMD_LINK', link)
      if (trim(link) == 'shm') then
         pid0 = 0
         if (my_task == 0) pid0 = cice_getpid()
         call MPI_BCAST(pid0, 1, MPI_INTEGER, 0, comm, ierr)
         write(txt,'(a,i0)') '/cice4_amd_mpi_', pid0
         cname = c_null_char
         do k = 1, len_trim(txt)
            cname(k) = txt(k:k)
         enddo
         call cice_gpu_check(cice_comm_init_shm(cice_gpu_ctx, cname, my_task, nprocs, 67108864_c_long_long), &
                             'cice_comm_init_shm')
         return
      endif
      uid = c_null_char
      if (my_task == 0) call cice_gpu_check(cice_comm_unique_id(uid), 'cice_comm_unique_id')
      call MPI_BCAST(uid, 128, MPI_CHARACTER, 0, comm, ierr)
      call cice_gpu_check(cice_comm_init(cice_gpu_ctx, uid, my_task, nprocs), 'cice_comm_init')
   end subroutine cice_gpu_comm_setup

   ! The one-launch subcycle loop ACROSS tasks (DESIGN.md section 7): the tasks (any cartesian layout of blocks: j-slabs,
   ! i-slabs, 2 x 2 ..., one block per task or several) hand each other the IPC handles of their exchange copies / progress words
   ! (control plane: MPI on the model's communicator) and the library maps those of the tasks it exchanges ghost cells with
   ! (cice_evp_peer_ranks: up to eight, the diagonal ones included) -- after that the two ice_HaloUpdate calls per subcycle of
   ! evp need no message.  Called after cice_evp_init by the drop-in ice_dyn_evp; where the decomposition is another one the
   ! library simply keeps the message path.  CICE4_AMD_PEER_LOOP=0 in the environment skips it; CICE4_AMD_PEER_SHARE=n tells
   ! the library that n tasks share one device (tests on a one-GPU box, with CICE4_AMD_LINK=shm).
   subroutine cice_gpu_peer_setup(my_task, nprocs, comm, nblocks_local)
      integer, intent(in) :: my_task, nprocs, comm, nblocks_local
      include 'mpif.h'
      character(kind=c_char) :: mine(64,3)
      character(kind=c_char), allocatable :: every(:,:,:)
      integer(c_long_long) :: plane(1)
      integer(c_long_long), allocatable :: planes(:)
      integer(c_int) :: nn, ranks(8)
      integer :: ierr, ok, allok, share, k
      character(len=16) :: txt
      call get_environment_variable('CICE4_AMD_PEER_LOOP', txt)
      if (trim(txt) == '0') return
      ok = 0
      if (nprocs > 1 .and. nblocks_local >= 1) ok = 1     ! (one block per task or several: the library says whether the layout qualifies)
      call MPI_ALLREDUCE(ok, allok, 1, MPI_INTEGER, MPI_MIN, comm, ierr)
      if (allok /= 1) return
      call get_environment_variable('CICE4_AMD_PEER_SHARE', txt)
      if (len_trim(txt) > 0) then
         read(txt,*) share
         call cice_gpu_check(cice_evp_set_option(cice_gpu_ctx, 'resident_peer_share'//c_null_char, share), 'resident_peer_share')
      endif
      call cice_gpu_check(cice_evp_peer_export_ipc(cice_gpu_ctx, mine, plane(1)), 'cice_evp_peer_export_ipc')
      allocate(every(64,3,nprocs), planes(nprocs))
      call MPI_ALLGATHER(mine, 192, MPI_CHARACTER, every, 192, MPI_CHARACTER, comm, ierr)
      call MPI_ALLGATHER(plane, 1, MPI_INTEGER8, planes, 1, MPI_INTEGER8, comm, ierr)
      call cice_gpu_check(cice_evp_peer_ranks(cice_gpu_ctx, nn, ranks), 'cice_evp_peer_ranks')
      do k = 1, nn
         call cice_gpu_check(cice_evp_peer_connect_rank_ipc(cice_gpu_ctx, ranks(k), every(:,:,ranks(k)+1), planes(ranks(k)+1)), &
                             'cice_evp_peer_connect_rank_ipc')
      enddo
      deallocate(every, planes)
      call MPI_BARRIER(comm, ierr)
      call cice_gpu_check(cice_evp_get_info(cice_gpu_ctx, 'resident_peer'//c_null_char, share), 'resident_peer')
      if (my_task == 0) then
         if (share == 1) then
            write(*,*) 'EVP subcycling as one launch per task: exchange buffers of the neighbouring tasks connected'
         else     ! e.g. a task boundary through a tripole fold, more tiles than compute units
            write(*,*) 'EVP subcycling keeps its message exchange (the cross-task loop does not cover this decomposition)'
         endif
      endif
   end subroutine cice_gpu_peer_setup
#endif

end module cice4_amd_c

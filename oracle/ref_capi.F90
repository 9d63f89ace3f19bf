!=======================================================================
! TEST INFRASTRUCTURE ONLY.  C-ABI capture wrapper around the COMPILED
! REFERENCE (COSIMA/cice4 Fortran modules built by oracle/build_ref.sh from
! /root/reference).  It lets tests/ and bench.py's cpu_baseline leg call the
! reference's own public procedures on arbitrary inputs through ctypes:
!   stress, stepu, evp_prep1/2, evp_finish, evp         (source/ice_dyn_evp.F90)
!   ice_strength                                         (source/ice_mechred.F90)
!   thermo_vertical, frzmlt_bottom_lateral               (source/ice_therm_vertical.F90)
!   ice_HaloUpdate, to_ugrid                             (serial/ice_boundary.F90, ice_grid.F90)
! This file is our own code; it contains no reference source.  It is never
! linked into the product library.
!=======================================================================
module ref_capi
   use iso_c_binding
   use ice_kinds_mod
   use ice_constants
   use ice_domain_size
   implicit none
   logical, save :: booted = .false.
   logical, save :: domain_ready = .false.
contains

   subroutine cstr(c, f)
      character(kind=c_char), intent(in) :: c(*)
      character(len=*), intent(out) :: f
      integer :: i
      f = ' '
      do i = 1, len(f)
         if (c(i) == c_null_char) exit
         f(i:i) = c(i)
      enddo
   end subroutine cstr

   subroutine ref_boot() bind(C, name='ref_boot')
      use ice_communicate, only: init_communicate
      use ice_fileunits, only: init_fileunits
      use ice_state, only: nt_Tsfc, nt_iage, ntrcr
      use ice_age, only: tr_iage
      if (booted) return
      call init_communicate
      call init_fileunits
      nt_Tsfc = 1
      nt_iage = 2
      ntrcr   = 2
      tr_iage = .true.
      booted = .true.
#ifdef REF_AUSCOM
      call ref_set_auscom(1.0_c_double, 0.0_c_double, 0.00536_c_double, 0.006_c_double, 0_c_int)   ! ice_init.F90:258-267
#endif
   end subroutine ref_boot

#ifdef REF_AUSCOM
   ! AusCOM / coupled build of ice_dyn_evp and ice_therm_vertical: the namelist variables ice_init would read, and the
   ! coupler array evp writes (sicemass, ice_dyn_evp.F90:246-248: allocated by the access-om driver's set-up otherwise)
   subroutine ref_set_auscom(cosw_in, sinw_in, dragio_in, chio_in, ocnslope) bind(C, name='ref_set_auscom')
      use ice_dyn_evp, only: cosw, sinw, dragio
      use ice_therm_vertical, only: chio
      use cpl_parameters, only: use_ocnslope
      use cpl_arrays_setup, only: sicemass
      use ice_blocks, only: nx_block, ny_block
      use ice_constants, only: Tocnfrz
      use ice_atmo, only: iceruf
      real(c_double), value :: cosw_in, sinw_in, dragio_in, chio_in
      integer(c_int), value :: ocnslope
      Tocnfrz = -1.8_dbl_kind; iceruf = 0.0005_dbl_kind       ! ice_init.F90 defaults of the namelist variables
      cosw = cosw_in; sinw = sinw_in; dragio = dragio_in; chio = chio_in
      use_ocnslope = (ocnslope /= 0)
      if (.not. allocated(sicemass)) allocate(sicemass(nx_block, ny_block, max_blocks))
   end subroutine ref_set_auscom
#endif

   subroutine ref_dims(d) bind(C, name='ref_dims')
      use ice_blocks, only: nx_block, ny_block
      integer(c_int), intent(out) :: d(10)
      d(1) = nx_block; d(2) = ny_block; d(3) = max_blocks
      d(4) = nx_global; d(5) = ny_global
      d(6) = ncat; d(7) = nilyr; d(8) = nslyr; d(9) = max_ntrcr
      d(10) = 0
   end subroutine ref_dims

   !--------------------------------------------------------------------
   ! EVP scalar parameters (set_evp_parameters, ice_dyn_evp.F90:535)
   !--------------------------------------------------------------------
   subroutine ref_set_evp_parameters(dt, ndte_in, damping, out) &
         bind(C, name='ref_set_evp_parameters')
      use ice_dyn_evp
      real(c_double), value :: dt
      integer(c_int), value :: ndte_in, damping
      real(c_double), intent(out) :: out(6)
      call ref_boot
      ndte = ndte_in
      kdyn = 1
      evp_damping = (damping /= 0)
      yield_curve = 'ellipse'
      call set_evp_parameters(dt)
      out(1) = dtei; out(2) = dte2T; out(3) = denom1; out(4) = denom2
      out(5) = rcon; out(6) = ecci
   end subroutine ref_set_evp_parameters

   subroutine ref_set_strength_parameters(kstr, kpartic, kredist, mu) &
         bind(C, name='ref_set_strength_parameters')
      use ice_mechred
      integer(c_int), value :: kstr, kpartic, kredist
      real(c_double), value :: mu
      kstrength = kstr; krdg_partic = kpartic; krdg_redist = kredist
      mu_rdg = mu
   end subroutine ref_set_strength_parameters

   !--------------------------------------------------------------------
   ! per-routine pass-throughs, any (nx,ny)
   !--------------------------------------------------------------------
#ifndef DROPIN
   ! routines internal to the reference's ice_dyn_evp (absent from the drop-in module)
   subroutine ref_stress(nx, ny, ksub, icellt, indxti, indxtj, uvel, vvel, &
         dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, tarear, tinyarea, strength, &
         sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124, &
         shear, divu, prs_sig, rdg_conv, rdg_shear, str) bind(C, name='ref_stress')
      use ice_dyn_evp, only: stress
      integer(c_int), value :: nx, ny, ksub, icellt
      integer(c_int), intent(in) :: indxti(nx*ny), indxtj(nx*ny)
      real(c_double), dimension(nx,ny), intent(in) :: uvel, vvel, dxt, dyt, dxhy, &
         dyhx, cxp, cyp, cxm, cym, tarear, tinyarea, strength
      real(c_double), dimension(nx,ny), intent(inout) :: sp1, sp2, sp3, sp4, sm1, &
         sm2, sm3, sm4, s121, s122, s123, s124, shear, divu, prs_sig, rdg_conv, rdg_shear
      real(c_double), intent(out) :: str(nx,ny,8)
      call stress(nx, ny, ksub, icellt, indxti, indxtj, uvel, vvel, dxt, dyt, &
         dxhy, dyhx, cxp, cyp, cxm, cym, tarear, tinyarea, strength, &
         sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124, &
         shear, divu, prs_sig, rdg_conv, rdg_shear, str)
   end subroutine ref_stress

   subroutine ref_stepu(nx, ny, icellu, indxui, indxuj, aiu, str, uocn, vocn, &
         waterx, watery, forcex, forcey, umassdtei, fm, uarear, strocnx, strocny, &
         strintx, strinty, uvel, vvel) bind(C, name='ref_stepu')
      use ice_dyn_evp, only: stepu
      integer(c_int), value :: nx, ny, icellu
      integer(c_int), intent(in) :: indxui(nx*ny), indxuj(nx*ny)
      real(c_double), dimension(nx,ny), intent(in) :: aiu, uocn, vocn, waterx, &
         watery, forcex, forcey, umassdtei, fm, uarear
      real(c_double), intent(in) :: str(nx,ny,8)
      real(c_double), dimension(nx,ny), intent(inout) :: strocnx, strocny, strintx, &
         strinty, uvel, vvel
      call stepu(nx, ny, icellu, indxui, indxuj, aiu, str, uocn, vocn, waterx, &
         watery, forcex, forcey, umassdtei, fm, uarear, strocnx, strocny, &
         strintx, strinty, uvel, vvel)
   end subroutine ref_stepu

   subroutine ref_evp_prep1(nx, ny, ilo, ihi, jlo, jhi, aice, vice, vsno, tmask, &
         strairxT, strairyT, strairx, strairy, tmass, icetmask) bind(C, name='ref_evp_prep1')
      use ice_dyn_evp, only: evp_prep1
      integer(c_int), value :: nx, ny, ilo, ihi, jlo, jhi
      real(c_double), dimension(nx,ny), intent(in) :: aice, vice, vsno, strairxT, strairyT
      integer(c_int), intent(in) :: tmask(nx,ny)
      real(c_double), dimension(nx,ny), intent(out) :: strairx, strairy, tmass
      integer(c_int), intent(out) :: icetmask(nx,ny)
      logical(log_kind) :: ltm(nx,ny)
      ltm = (tmask /= 0)
      call evp_prep1(nx, ny, ilo, ihi, jlo, jhi, aice, vice, vsno, ltm, &
         strairxT, strairyT, strairx, strairy, tmass, icetmask)
   end subroutine ref_evp_prep1

   subroutine ref_evp_prep2(nx, ny, ilo, ihi, jlo, jhi, icellt, icellu, &
         indxti, indxtj, indxui, indxuj, aiu, umass, umassdtei, fcor, umask, &
         uocn, vocn, strairx, strairy, ss_tltx, ss_tlty, icetmask, iceumask, fm, &
         strtltx, strtlty, strocnx, strocny, strintx, strinty, waterx, watery, &
         forcex, forcey, sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, &
         s124, uvel, vvel) bind(C, name='ref_evp_prep2')
      use ice_dyn_evp, only: evp_prep2
      integer(c_int), value :: nx, ny, ilo, ihi, jlo, jhi
      integer(c_int), intent(out) :: icellt, icellu
      integer(c_int), dimension(nx*ny), intent(out) :: indxti, indxtj, indxui, indxuj
      real(c_double), dimension(nx,ny), intent(in) :: aiu, umass, fcor, uocn, vocn, &
         strairx, strairy, ss_tltx, ss_tlty
      integer(c_int), intent(in) :: umask(nx,ny), icetmask(nx,ny)
      integer(c_int), intent(inout) :: iceumask(nx,ny)
      real(c_double), dimension(nx,ny), intent(out) :: umassdtei, waterx, watery, &
         forcex, forcey
      real(c_double), dimension(nx,ny), intent(inout) :: fm, strtltx, strtlty, &
         strocnx, strocny, strintx, strinty, sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, &
         s121, s122, s123, s124, uvel, vvel
      logical(log_kind) :: lum(nx,ny), lium(nx,ny)
      lum = (umask /= 0)
      lium = (iceumask /= 0)
      call evp_prep2(nx, ny, ilo, ihi, jlo, jhi, icellt, icellu, indxti, indxtj, &
         indxui, indxuj, aiu, umass, umassdtei, fcor, lum, uocn, vocn, strairx, &
         strairy, ss_tltx, ss_tlty, icetmask, lium, fm, strtltx, strtlty, strocnx, &
         strocny, strintx, strinty, waterx, watery, forcex, forcey, sp1, sp2, sp3, &
         sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124, uvel, vvel)
      where (lium)
         iceumask = 1
      elsewhere
         iceumask = 0
      end where
   end subroutine ref_evp_prep2

   subroutine ref_evp_finish(nx, ny, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, &
         aiu, strocnx, strocny, strocnxT, strocnyT) bind(C, name='ref_evp_finish')
      use ice_dyn_evp, only: evp_finish
      integer(c_int), value :: nx, ny, icellu
      integer(c_int), intent(in) :: indxui(nx*ny), indxuj(nx*ny)
      real(c_double), dimension(nx,ny), intent(in) :: uvel, vvel, uocn, vocn, aiu
      real(c_double), dimension(nx,ny), intent(inout) :: strocnx, strocny, strocnxT, strocnyT
#ifdef REF_AUSCOM
      stop 'ref_evp_finish: the AusCOM build takes fm (ref_evp_finish_fm)'
#else
      call evp_finish(nx, ny, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, aiu, &
         strocnx, strocny, strocnxT, strocnyT)
#endif
   end subroutine ref_evp_finish

#ifdef REF_AUSCOM
   subroutine ref_evp_finish_fm(nx, ny, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, &
         aiu, fm, strocnx, strocny, strocnxT, strocnyT) bind(C, name='ref_evp_finish_fm')
      use ice_dyn_evp, only: evp_finish
      integer(c_int), value :: nx, ny, icellu
      integer(c_int), intent(in) :: indxui(nx*ny), indxuj(nx*ny)
      real(c_double), dimension(nx,ny), intent(in) :: uvel, vvel, uocn, vocn, aiu, fm
      real(c_double), dimension(nx,ny), intent(inout) :: strocnx, strocny, strocnxT, strocnyT
      call evp_finish(nx, ny, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, aiu, fm, &
         strocnx, strocny, strocnxT, strocnyT)
   end subroutine ref_evp_finish_fm
#endif

#endif
   subroutine ref_ice_strength(nx, ny, ilo, ihi, jlo, jhi, icells, indxi, indxj, &
         aice, vice, aice0, aicen, vicen, strength) bind(C, name='ref_ice_strength')
      use ice_mechred, only: ice_strength
      integer(c_int), value :: nx, ny, ilo, ihi, jlo, jhi, icells
      integer(c_int), intent(in) :: indxi(nx*ny), indxj(nx*ny)
      real(c_double), dimension(nx,ny), intent(in) :: aice, vice, aice0
      real(c_double), dimension(nx,ny,ncat), intent(in) :: aicen, vicen
      real(c_double), intent(out) :: strength(nx,ny)
      call ice_strength(nx, ny, ilo, ihi, jlo, jhi, icells, indxi, indxj, aice, &
         vice, aice0, aicen, vicen, strength)
   end subroutine ref_ice_strength

   !--------------------------------------------------------------------
   ! thermodynamics
   !--------------------------------------------------------------------
   subroutine ref_init_thermo(heatcap, calcts, conduct_id, ustarmin, salin_out, &
         tmlt_out) bind(C, name='ref_init_thermo')
      use ice_therm_vertical
      use ice_itd, only: ilyr1, ilyrn, slyr1, slyrn
      integer(c_int), value :: heatcap, calcts, conduct_id
      real(c_double), value :: ustarmin
      real(c_double), intent(out) :: salin_out(nilyr+1), tmlt_out(nilyr+1)
      integer :: n
      call ref_boot
      heat_capacity = (heatcap /= 0)
      calc_Tsfc = (calcts /= 0)
      if (conduct_id == 0) then
         conduct = 'MU71'
      else
         conduct = 'bubbly'
      endif
      ustar_min = ustarmin
      call init_thermo_vertical
      salin_out = salin
      tmlt_out = Tmlt
      ! layer index maps exactly as init_itd sets them (ice_itd.F90:240-260)
      ilyr1(1) = 1; ilyrn(1) = nilyr; slyr1(1) = 1; slyrn(1) = nslyr
      do n = 2, ncat
         ilyr1(n) = ilyrn(n-1) + 1; ilyrn(n) = ilyrn(n-1) + nilyr
         slyr1(n) = slyrn(n-1) + 1; slyrn(n) = slyrn(n-1) + nslyr
      enddo
   end subroutine ref_init_thermo

   subroutine ref_thermo_vertical(nx, ny, dt, icells, indxi, indxj, aicen, trcrn, &
         vicen, vsnon, eicen, esnon, flw, potT, Qa, rhoa, fsnow, fbot, Tbot, lhcoef, &
         shcoef, fswsfc, fswint, fswthrun, Sswabs, Iswabs, fsurfn, fcondtopn, fsensn, &
         flatn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, &
         congel, snoice, mlt_onset, frz_onset, yday, l_stop, istop, jstop) &
         bind(C, name='ref_thermo_vertical')
      use ice_therm_vertical, only: thermo_vertical
      integer(c_int), value :: nx, ny, icells
      real(c_double), value :: dt, yday
      integer(c_int), intent(in) :: indxi(nx*ny), indxj(nx*ny)
      real(c_double), dimension(nx,ny), intent(inout) :: aicen, vicen, vsnon
      real(c_double), intent(inout) :: trcrn(nx,ny,max_ntrcr), eicen(nx,ny,nilyr), &
         esnon(nx,ny,nslyr)
      real(c_double), dimension(nx,ny), intent(in) :: flw, potT, Qa, rhoa, fsnow, &
         fbot, Tbot, lhcoef, shcoef
      real(c_double), dimension(nx,ny), intent(inout) :: fswsfc, fswint, fswthrun
      real(c_double), intent(inout) :: Sswabs(nx,ny,nslyr), Iswabs(nx,ny,nilyr)
      real(c_double), dimension(nx,ny), intent(inout) :: fsurfn, fcondtopn, fsensn, &
         flatn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, &
         congel, snoice, mlt_onset, frz_onset
      integer(c_int), intent(out) :: l_stop, istop, jstop
      logical(log_kind) :: ls
      call thermo_vertical(nx, ny, dt, icells, indxi, indxj, aicen, trcrn, vicen, &
         vsnon, eicen, esnon, flw, potT, Qa, rhoa, fsnow, fbot, Tbot, lhcoef, shcoef, &
         fswsfc, fswint, fswthrun, Sswabs, Iswabs, fsurfn, fcondtopn, fsensn, flatn, &
         fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, congel, &
         snoice, mlt_onset, frz_onset, yday, ls, istop, jstop)
      l_stop = 0
      if (ls) l_stop = 1
   end subroutine ref_thermo_vertical

   subroutine ref_merge_fluxes(nx, ny, icells, indxi, indxj, aicen, flw, c, a) bind(C, name='ref_merge_fluxes')
      ! c(:,:,k), a(:,:,k): the 20 per-category / cumulative fields in the order strairx, strairy,
      ! fsurf, fcondtop, fsens, flat, fswabs, flwout, evap, Tref, Qref, fresh, fsalt, fhocn, fswthru,
      ! meltt, meltb, melts, congel, snoice
      use ice_flux, only: merge_fluxes
      integer(c_int), value :: nx, ny, icells
      integer(c_int), intent(in) :: indxi(nx*ny), indxj(nx*ny)
      real(c_double), intent(in) :: aicen(nx,ny), flw(nx,ny), c(nx,ny,20)
      real(c_double), intent(inout) :: a(nx,ny,20)
      real(c_double) :: coszn(nx,ny)
      coszn = 0.0d0
      call merge_fluxes(nx, ny, icells, indxi, indxj, aicen, flw, coszn, c(:,:,1), c(:,:,2), c(:,:,3), &
         c(:,:,4), c(:,:,5), c(:,:,6), c(:,:,7), c(:,:,8), c(:,:,9), c(:,:,10), c(:,:,11), c(:,:,12), &
         c(:,:,13), c(:,:,14), c(:,:,15), a(:,:,1), a(:,:,2), a(:,:,3), a(:,:,4), a(:,:,5), a(:,:,6), &
         a(:,:,7), a(:,:,8), a(:,:,9), a(:,:,10), a(:,:,11), a(:,:,12), a(:,:,13), a(:,:,14), a(:,:,15), &
         c(:,:,16), c(:,:,18), c(:,:,17), c(:,:,19), c(:,:,20), a(:,:,16), a(:,:,18), a(:,:,17), &
         a(:,:,19), a(:,:,20))
   end subroutine ref_merge_fluxes

   subroutine ref_frzmlt_bottom_lateral(nx, ny, ilo, ihi, jlo, jhi, dt, aice, frzmlt, &
         eicen, esnon, sst, Tf, strocnxT, strocnyT, Tbot, fbot, rside) &
         bind(C, name='ref_frzmlt_bottom_lateral')
      use ice_therm_vertical, only: frzmlt_bottom_lateral
      integer(c_int), value :: nx, ny, ilo, ihi, jlo, jhi
      real(c_double), value :: dt
      real(c_double), dimension(nx,ny), intent(in) :: aice, frzmlt, sst, Tf, strocnxT, strocnyT
      real(c_double), intent(in) :: eicen(nx,ny,ntilyr), esnon(nx,ny,ntslyr)
      real(c_double), dimension(nx,ny), intent(out) :: Tbot, fbot, rside
      call frzmlt_bottom_lateral(nx, ny, ilo, ihi, jlo, jhi, dt, aice, frzmlt, eicen, &
         esnon, sst, Tf, strocnxT, strocnyT, Tbot, fbot, rside)
   end subroutine ref_frzmlt_bottom_lateral

   ! atmo_boundary_layer (source/ice_atmo.F90:56), ocn = 0 'ice' / 1 'ocn'; calc_strair is the module variable
   subroutine ref_atmo_boundary_layer(nx, ny, ocn, icells, indxi, indxj, Tsf, potT, uatm, vatm, wind, zlvl, &
         Qa, rhoa, calc_strair_in, strx, stry, Tref, Qref, delt, delq, lhcoef, shcoef) &
         bind(C, name='ref_atmo_boundary_layer')
      use ice_atmo, only: atmo_boundary_layer, calc_strair
      integer(c_int), value :: nx, ny, ocn, icells, calc_strair_in
      integer(c_int), intent(in) :: indxi(nx*ny), indxj(nx*ny)
      real(c_double), dimension(nx,ny), intent(in) :: Tsf, potT, uatm, vatm, wind, zlvl, Qa, rhoa
      real(c_double), dimension(nx,ny), intent(inout) :: strx, stry
      real(c_double), dimension(nx,ny), intent(out) :: Tref, Qref, delt, delq, lhcoef, shcoef
      character(len=3) :: sfctype
      sfctype = 'ice'
      if (ocn /= 0) sfctype = 'ocn'
      calc_strair = (calc_strair_in /= 0)
      call atmo_boundary_layer(nx, ny, sfctype, icells, indxi, indxj, Tsf, potT, uatm, vatm, wind, zlvl, &
         Qa, rhoa, strx, stry, Tref, Qref, delt, delq, lhcoef, shcoef)
      calc_strair = .true.
   end subroutine ref_atmo_boundary_layer

   !--------------------------------------------------------------------
   ! whole-domain set-up (cice_init subset, CICE_InitMod.F90:124-150) so that
   ! evp(dt) can run on the reference's own module arrays, blocks and halos.
   ! The working directory must hold an `ice_in` with a domain_nml.
   !--------------------------------------------------------------------
   integer(c_int) function ref_init_domain(grid_kind, gridfile, kmtfile, dt, ndte_in, &
         damping) bind(C, name='ref_init_domain')
      use ice_work, only: init_work
      use ice_domain, only: init_domain_blocks, nblocks
      use ice_grid
      use ice_timers, only: init_ice_timers
      use ice_dyn_evp
      use ice_flux, only: init_coupler_flux
      use ice_itd, only: init_itd, kitd, kcatbound
      integer(c_int), value :: grid_kind, ndte_in, damping
      character(kind=c_char), intent(in) :: gridfile(*), kmtfile(*)
      real(c_double), value :: dt
      call ref_boot
      if (domain_ready) then
         ref_init_domain = nblocks
         return
      endif
      if (grid_kind == 1) then
         grid_type = 'displaced_pole'
         grid_format = 'bin'
         call cstr(gridfile, grid_file)
         call cstr(kmtfile, kmt_file)
      else
         grid_type = 'rectangular'
         grid_format = 'bin'
      endif
      call init_work
      call init_domain_blocks
      call init_grid1
      call init_ice_timers
      call init_grid2
      ndte = ndte_in
      kdyn = 1
      evp_damping = (damping /= 0)
      yield_curve = 'ellipse'
      call init_evp(dt)
      call init_coupler_flux
      kitd = 1
      kcatbound = 0
      call init_itd
      domain_ready = .true.
      ref_init_domain = nblocks
   end function ref_init_domain

   ! block decomposition and distribution only (init_domain_blocks + init_domain_distribution with an
   ! all-ocean mask): what decides which task owns which block.  Works under mpiexec with the MPI build.
   integer(c_int) function ref_init_topology(info) bind(C, name='ref_init_topology')
      use ice_domain, only: init_domain_blocks, init_domain_distribution, nblocks
      use ice_communicate, only: my_task
      use ice_distribution, only: nprocsX, nprocsY
      integer(c_int), intent(out) :: info(4)
      real(dbl_kind), allocatable :: kmtg(:,:), ulatg(:,:)
      call ref_boot
      call init_domain_blocks
      allocate(kmtg(nx_global,ny_global), ulatg(nx_global,ny_global))
      kmtg = c1
      ulatg = 1.3_dbl_kind
      call init_domain_distribution(kmtg, ulatg)
      deallocate(kmtg, ulatg)
      info(1) = my_task; info(2) = nprocsX; info(3) = nprocsY; info(4) = nblocks
      ref_init_topology = nblocks
   end function ref_init_topology

   subroutine ref_end_run() bind(C, name='ref_end_run')
      use ice_exit, only: end_run
      call end_run
   end subroutine ref_end_run

   subroutine ref_block_info(iblk, info, iglob, jglob) bind(C, name='ref_block_info')
      use ice_blocks
      use ice_domain, only: blocks_ice
      integer(c_int), value :: iblk
      integer(c_int), intent(out) :: info(6), iglob(nx_block), jglob(ny_block)
      type(block) :: b
      b = get_block(blocks_ice(iblk), iblk)
      info(1) = b%ilo; info(2) = b%ihi; info(3) = b%jlo; info(4) = b%jhi
      info(5) = b%block_id; info(6) = b%local_id
      iglob = b%i_glob
      jglob = b%j_glob
   end subroutine ref_block_info

#ifdef DROPIN
   ! drop-in build only: push the (possibly injected) host grid to the device again
   subroutine ref_evp_gpu_setup() bind(C, name='ref_evp_gpu_setup')
      use ice_dyn_evp, only: evp_gpu_setup
      call evp_gpu_setup
   end subroutine ref_evp_gpu_setup
#endif

#ifdef DROPIN
   ! drop-in build only (tests/mpi_evp_case.py `badsize`): a library call that fails on THIS task -- a model built with
   ! another ncat than the library -- must end the whole MPI job (cice_gpu_check -> MPI_ABORT), not just this task
   subroutine ref_gpu_bad_size() bind(C, name='ref_gpu_bad_size')
      use cice4_amd_c, only: cice_check_sizes, cice_gpu_check, cice_gpu_ctx
      use ice_domain_size, only: ncat, nilyr, nslyr, max_ntrcr
      call cice_gpu_check(cice_check_sizes(cice_gpu_ctx, ncat + 1, nilyr, nslyr, max_ntrcr), 'ref_gpu_bad_size (test)')
   end subroutine ref_gpu_bad_size

   integer(c_int) function ref_evp_info(ckey) bind(C, name='ref_evp_info')
      use cice4_amd_c, only: cice_evp_get_info, cice_gpu_ctx
      character(kind=c_char), intent(in) :: ckey(*)
      integer(c_int) :: v, rc
      v = -1
      rc = cice_evp_get_info(cice_gpu_ctx, ckey, v)
      ref_evp_info = v
   end function ref_evp_info
#endif

   subroutine ref_evp(dt) bind(C, name='ref_evp')
      use ice_dyn_evp, only: evp
      real(c_double), value :: dt
      call evp(dt)
   end subroutine ref_evp

   subroutine ref_halo_r8(a, loc, kind) bind(C, name='ref_halo_r8')
      use ice_blocks, only: nx_block, ny_block
      use ice_boundary
      use ice_domain, only: halo_info
      real(c_double), intent(inout) :: a(nx_block,ny_block,max_blocks)
      integer(c_int), value :: loc, kind
      call ice_HaloUpdate(a, halo_info, loc, kind)
   end subroutine ref_halo_r8

   subroutine ref_halo_i4(a, loc, kind) bind(C, name='ref_halo_i4')
      use ice_blocks, only: nx_block, ny_block
      use ice_boundary
      use ice_domain, only: halo_info
      integer(c_int), intent(inout) :: a(nx_block,ny_block,max_blocks)
      integer(c_int), value :: loc, kind
      call ice_HaloUpdate(a, halo_info, loc, kind)
   end subroutine ref_halo_i4

   ! generic n-d update: typ 0 = R8, 1 = R4, 2 = I4; nz = nt = 0 -> (nx,ny,nblk);
   ! nt = 0 -> (nx,ny,nz,nblk); else (nx,ny,nz,nt,nblk)
   subroutine ref_halo_nd(buf, typ, nz, nt, loc, kind) bind(C, name='ref_halo_nd')
      use ice_blocks, only: nx_block, ny_block
      use ice_boundary
      use ice_domain, only: halo_info
      type(c_ptr), value :: buf
      integer(c_int), value :: typ, nz, nt, loc, kind
      real(c_double), pointer :: d2(:,:,:), d3(:,:,:,:), d4(:,:,:,:,:)
      real(c_float), pointer :: f2(:,:,:), f3(:,:,:,:), f4(:,:,:,:,:)
      integer(c_int), pointer :: i2(:,:,:), i3(:,:,:,:), i4(:,:,:,:,:)
      if (nz == 0) then
         select case (typ)
         case (0); call c_f_pointer(buf, d2, [nx_block,ny_block,max_blocks])
                   call ice_HaloUpdate(d2, halo_info, loc, kind)
         case (1); call c_f_pointer(buf, f2, [nx_block,ny_block,max_blocks])
                   call ice_HaloUpdate(f2, halo_info, loc, kind)
         case (2); call c_f_pointer(buf, i2, [nx_block,ny_block,max_blocks])
                   call ice_HaloUpdate(i2, halo_info, loc, kind)
         end select
      else if (nt == 0) then
         select case (typ)
         case (0); call c_f_pointer(buf, d3, [nx_block,ny_block,nz,max_blocks])
                   call ice_HaloUpdate(d3, halo_info, loc, kind)
         case (1); call c_f_pointer(buf, f3, [nx_block,ny_block,nz,max_blocks])
                   call ice_HaloUpdate(f3, halo_info, loc, kind)
         case (2); call c_f_pointer(buf, i3, [nx_block,ny_block,nz,max_blocks])
                   call ice_HaloUpdate(i3, halo_info, loc, kind)
         end select
      else
         select case (typ)
         case (0); call c_f_pointer(buf, d4, [nx_block,ny_block,nz,nt,max_blocks])
                   call ice_HaloUpdate(d4, halo_info, loc, kind)
         case (1); call c_f_pointer(buf, f4, [nx_block,ny_block,nz,nt,max_blocks])
                   call ice_HaloUpdate(f4, halo_info, loc, kind)
         case (2); call c_f_pointer(buf, i4, [nx_block,ny_block,nz,nt,max_blocks])
                   call ice_HaloUpdate(i4, halo_info, loc, kind)
         end select
      endif
   end subroutine ref_halo_nd

   subroutine ref_halo_extrapolate(a) bind(C, name='ref_halo_extrapolate')
      use ice_blocks, only: nx_block, ny_block
      use ice_boundary
      use ice_domain, only: distrb_info, ew_boundary_type, ns_boundary_type
      real(c_double), intent(inout) :: a(nx_block,ny_block,max_blocks)
      call ice_HaloExtrapolate(a, distrb_info, ew_boundary_type, ns_boundary_type)
   end subroutine ref_halo_extrapolate

   ! the state-variable ghost update of ice_state.F90:bound_state on caller-supplied arrays
   subroutine ref_bound_state(aicen, trcrn, vicen, vsnon, eicen, esnon) bind(C, name='ref_bound_state')
      use ice_blocks, only: nx_block, ny_block
      use ice_state, only: bound_state
      real(c_double), intent(inout) :: aicen(nx_block,ny_block,ncat,max_blocks), &
         trcrn(nx_block,ny_block,max_ntrcr,ncat,max_blocks), vicen(nx_block,ny_block,ncat,max_blocks), &
         vsnon(nx_block,ny_block,ncat,max_blocks), eicen(nx_block,ny_block,ntilyr,max_blocks), &
         esnon(nx_block,ny_block,ntslyr,max_blocks)
      call bound_state(aicen, trcrn, vicen, vsnon, eicen, esnon)
   end subroutine ref_bound_state

   subroutine ref_to_ugrid(w1, w2) bind(C, name='ref_to_ugrid')
      use ice_blocks, only: nx_block, ny_block
      use ice_grid, only: to_ugrid
      real(c_double), intent(in) :: w1(nx_block,ny_block,max_blocks)
      real(c_double), intent(out) :: w2(nx_block,ny_block,max_blocks)
      call to_ugrid(w1, w2)
   end subroutine ref_to_ugrid

   !--------------------------------------------------------------------
   ! field access by name: dir = 0 get (module -> buf), 1 set (buf -> module).
   ! nlev = number of (nx_block,ny_block,max_blocks) slabs in buf; returns
   ! nlev actually moved, or -1 for an unknown name.  Logical masks travel
   ! as 0.0 / 1.0.
   !--------------------------------------------------------------------
   integer(c_int) function ref_field(cname, dir, buf) bind(C, name='ref_field')
      use ice_blocks, only: nx_block, ny_block
      use ice_state
      use ice_flux
      use ice_grid
      use ice_dyn_evp, only: fcor_blk
#ifdef REF_AUSCOM
      use cpl_arrays_setup, only: sicemass
#endif
      character(kind=c_char), intent(in) :: cname(*)
      integer(c_int), value :: dir
      real(c_double), intent(inout) :: buf(nx_block,ny_block,*)
      character(len=32) :: name
      integer :: n, k, m, iblk
      call cstr(cname, name)
      n = 1
#define F2(nm) if (dir == 0) then; buf(:,:,1:max_blocks) = nm; else; nm = buf(:,:,1:max_blocks); endif
      select case (trim(name))
      case ('aice');  F2(aice)
      case ('vice');  F2(vice)
      case ('vsno');  F2(vsno)
      case ('aice0'); F2(aice0)
      case ('uvel');  F2(uvel)
      case ('vvel');  F2(vvel)
      case ('divu');  F2(divu)
      case ('shear'); F2(shear)
      case ('strength'); F2(strength)
      case ('rdg_conv'); F2(rdg_conv)
      case ('rdg_shear'); F2(rdg_shear)
      case ('prs_sig'); F2(prs_sig)
      case ('strairxT'); F2(strairxT)
      case ('strairyT'); F2(strairyT)
      case ('strairx'); F2(strairx)
      case ('strairy'); F2(strairy)
      case ('uocn'); F2(uocn)
      case ('vocn'); F2(vocn)
      case ('ss_tltx'); F2(ss_tltx)
      case ('ss_tlty'); F2(ss_tlty)
      case ('strtltx'); F2(strtltx)
      case ('strtlty'); F2(strtlty)
      case ('strocnx'); F2(strocnx)
      case ('strocny'); F2(strocny)
      case ('strocnxT'); F2(strocnxT)
      case ('strocnyT'); F2(strocnyT)
      case ('strintx'); F2(strintx)
      case ('strinty'); F2(strinty)
      case ('fm'); F2(fm)
#ifdef REF_AUSCOM
      case ('sicemass'); F2(sicemass)
#endif
      case ('stressp_1'); F2(stressp_1)
      case ('stressp_2'); F2(stressp_2)
      case ('stressp_3'); F2(stressp_3)
      case ('stressp_4'); F2(stressp_4)
      case ('stressm_1'); F2(stressm_1)
      case ('stressm_2'); F2(stressm_2)
      case ('stressm_3'); F2(stressm_3)
      case ('stressm_4'); F2(stressm_4)
      case ('stress12_1'); F2(stress12_1)
      case ('stress12_2'); F2(stress12_2)
      case ('stress12_3'); F2(stress12_3)
      case ('stress12_4'); F2(stress12_4)
      case ('fcor'); F2(fcor_blk)
      case ('dxt'); F2(dxt)
      case ('dyt'); F2(dyt)
      case ('dxu'); F2(dxu)
      case ('dyu'); F2(dyu)
      case ('HTE'); F2(HTE)
      case ('HTN'); F2(HTN)
      case ('tarea'); F2(tarea)
      case ('uarea'); F2(uarea)
      case ('tarear'); F2(tarear)
      case ('uarear'); F2(uarear)
      case ('tinyarea'); F2(tinyarea)
      case ('dxhy'); F2(dxhy)
      case ('dyhx'); F2(dyhx)
      case ('cxp'); F2(cxp)
      case ('cyp'); F2(cyp)
      case ('cxm'); F2(cxm)
      case ('cym'); F2(cym)
      case ('ULAT'); F2(ULAT)
      case ('ULON'); F2(ULON)
      case ('TLAT'); F2(TLAT)
      case ('TLON'); F2(TLON)
      case ('ANGLE'); F2(ANGLE)
      case ('hm'); F2(hm)
      case ('uvm'); F2(uvm)
      case ('tmask')
         if (dir == 0) then
            buf(:,:,1:max_blocks) = merge(1.0d0, 0.0d0, tmask)
         else
            tmask = (buf(:,:,1:max_blocks) /= 0.0d0)
         endif
      case ('umask')
         if (dir == 0) then
            buf(:,:,1:max_blocks) = merge(1.0d0, 0.0d0, umask)
         else
            umask = (buf(:,:,1:max_blocks) /= 0.0d0)
         endif
      case ('iceumask')
         if (dir == 0) then
            buf(:,:,1:max_blocks) = merge(1.0d0, 0.0d0, iceumask)
         else
            iceumask = (buf(:,:,1:max_blocks) /= 0.0d0)
         endif
      case ('aicen')
         n = ncat*max_blocks
         do iblk = 1, max_blocks
            do k = 1, ncat
               m = (iblk-1)*ncat + k
               if (dir == 0) then
                  buf(:,:,m) = aicen(:,:,k,iblk)
               else
                  aicen(:,:,k,iblk) = buf(:,:,m)
               endif
            enddo
         enddo
      case ('vicen')
         n = ncat*max_blocks
         do iblk = 1, max_blocks
            do k = 1, ncat
               m = (iblk-1)*ncat + k
               if (dir == 0) then
                  buf(:,:,m) = vicen(:,:,k,iblk)
               else
                  vicen(:,:,k,iblk) = buf(:,:,m)
               endif
            enddo
         enddo
      case ('vsnon')
         n = ncat*max_blocks
         do iblk = 1, max_blocks
            do k = 1, ncat
               m = (iblk-1)*ncat + k
               if (dir == 0) then
                  buf(:,:,m) = vsnon(:,:,k,iblk)
               else
                  vsnon(:,:,k,iblk) = buf(:,:,m)
               endif
            enddo
         enddo
      case ('trcrn')      ! (nx,ny,max_ntrcr,ncat,max_blocks) as max_ntrcr*ncat*max_blocks planes
         n = max_ntrcr*ncat*max_blocks
         do iblk = 1, max_blocks
            do k = 1, ncat
               do m = 1, max_ntrcr
                  if (dir == 0) then
                     buf(:,:,((iblk-1)*ncat + k-1)*max_ntrcr + m) = trcrn(:,:,m,k,iblk)
                  else
                     trcrn(:,:,m,k,iblk) = buf(:,:,((iblk-1)*ncat + k-1)*max_ntrcr + m)
                  endif
               enddo
            enddo
         enddo
      case ('eicen')
         n = ntilyr*max_blocks
         do iblk = 1, max_blocks
            do k = 1, ntilyr
               if (dir == 0) then
                  buf(:,:,(iblk-1)*ntilyr + k) = eicen(:,:,k,iblk)
               else
                  eicen(:,:,k,iblk) = buf(:,:,(iblk-1)*ntilyr + k)
               endif
            enddo
         enddo
      case ('esnon')
         n = ntslyr*max_blocks
         do iblk = 1, max_blocks
            do k = 1, ntslyr
               if (dir == 0) then
                  buf(:,:,(iblk-1)*ntslyr + k) = esnon(:,:,k,iblk)
               else
                  esnon(:,:,k,iblk) = buf(:,:,(iblk-1)*ntslyr + k)
               endif
            enddo
         enddo
      case default
         n = -1
      end select
      ref_field = n
   end function ref_field

   !--------------------------------------------------------------------
   ! horizontal transport: init_transport (ice_transport_driver.F90:81) with advection = 'remap' and the
   ! tracer set of the gx3 namelist (Tsfc, iage), then transport_remap(dt) (:179) on the module state
   !--------------------------------------------------------------------
   subroutine ref_init_transport() bind(C, name='ref_init_transport')
      use ice_transport_driver, only: init_transport, advection
      use ice_state, only: ntrcr, nt_Tsfc, nt_iage, trcr_depend
      logical, save :: done = .false.
      if (done) return
      advection = 'remap'
      ntrcr = 2
      nt_Tsfc = 1
      nt_iage = 2
      trcr_depend(:) = 0
      trcr_depend(nt_iage) = 1     ! ice_init.F90:848-849
      call init_transport
      done = .true.
   end subroutine ref_init_transport

   ! pieces of transport_remap for stage-by-stage comparisons: state_to_tracers for every local block, and
   ! horizontal_remap on caller-supplied mean fields (module uvel, vvel); edgearea_* come back (l_fixed_area = F)
   subroutine ref_state_to_tracers(aim, trm) bind(C, name='ref_state_to_tracers')
      use ice_transport_driver, only: state_to_tracers, ntrace
      use ice_blocks, only: nx_block, ny_block
      use ice_domain, only: nblocks
      use ice_state
      real(c_double), intent(out) :: aim(nx_block,ny_block,0:ncat,max_blocks)
      real(c_double), intent(out) :: trm(nx_block,ny_block,ntrace,ncat,max_blocks)
      integer :: iblk
      aim = 0; trm = 0
      do iblk = 1, nblocks
         call state_to_tracers(nx_block, ny_block, ntrcr, ntrace, aice0(:,:,iblk), aicen(:,:,:,iblk), &
                               trcrn(:,:,1:ntrcr,:,iblk), vicen(:,:,:,iblk), vsnon(:,:,:,iblk), &
                               eicen(:,:,:,iblk), esnon(:,:,:,iblk), aim(:,:,:,iblk), trm(:,:,:,:,iblk))
      enddo
   end subroutine ref_state_to_tracers

   subroutine ref_horizontal_remap(dt, aim, trm, ee, en) bind(C, name='ref_horizontal_remap')
      use ice_transport_driver, only: ntrace, tracer_type, depend, has_dependents, integral_order, l_dp_midpt, &
                                      l_fixed_area
      use ice_transport_remap, only: horizontal_remap
      use ice_blocks, only: nx_block, ny_block
      use ice_state, only: uvel, vvel
      real(c_double), value :: dt
      real(c_double), intent(inout) :: aim(nx_block,ny_block,0:ncat,max_blocks)
      real(c_double), intent(inout) :: trm(nx_block,ny_block,ntrace,ncat,max_blocks)
      real(c_double), intent(out) :: ee(nx_block,ny_block,max_blocks), en(nx_block,ny_block,max_blocks)
      ee = 0; en = 0
      call horizontal_remap(dt, ntrace, uvel, vvel, aim, trm, l_fixed_area, ee, en, tracer_type, depend, &
                            has_dependents, integral_order, l_dp_midpt)
   end subroutine ref_horizontal_remap

   subroutine ref_transport_remap(dt) bind(C, name='ref_transport_remap')
      use ice_transport_driver, only: transport_remap
      real(c_double), value :: dt
      call transport_remap(dt)
   end subroutine ref_transport_remap

   subroutine ref_transport_upwind(dt) bind(C, name='ref_transport_upwind')
      use ice_transport_driver, only: transport_upwind
      real(c_double), value :: dt
      call transport_upwind(dt)
   end subroutine ref_transport_upwind

end module ref_capi

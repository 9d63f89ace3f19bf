!=======================================================================
! Drop-in replacement for the reference's source/ice_therm_vertical.F90.
!
! Same module name and the public entities other files use: saltmax, hs_min,
! betak, kimin, salin, Tmlt, ustar_min, conduct, l_brine, heat_capacity,
! calc_Tsfc (ice_init.F90:107,941; ice_shortwave.F90:990; ice_diagnostics.F90:128;
! CICE_RunMod.F90:1029), thermo_vertical (CICE_RunMod.F90:502),
! surface_fluxes (CICE_RunMod.F90:1216), init_thermo_vertical (CICE_InitMod.F90), frzmlt_bottom_lateral
! (CICE_RunMod.F90:363), calculate_Tin_from_qin (ice_history.F90:1731).
! The column physics runs on the GPU (libcice4_amd.so) through cice4_amd_c;
! nothing of the reference's implementation is kept here.
!=======================================================================
      module ice_therm_vertical

      use ice_kinds_mod
      use ice_domain_size, only: ncat, nilyr, nslyr, ntilyr, ntslyr, max_ntrcr
      use ice_constants
      use ice_fileunits, only: nu_diag
      use iso_c_binding
      use cice4_amd_c

      implicit none
      save

      real (kind=dbl_kind), parameter :: &
         saltmax = 3.2_dbl_kind, hs_min = 1.e-4_dbl_kind, betak = 0.13_dbl_kind, &
         kimin = 0.10_dbl_kind

      real (kind=dbl_kind), dimension(nilyr+1) :: salin, Tmlt
      real (kind=dbl_kind) :: ustar_min
#if defined(AusCOM)
      real (kind=dbl_kind) :: chio   ! namelist (ice_init.F90:99,156): basal heat transfer, link with -lcice4_amd_auscom
#endif
      character (char_len) :: conduct
      logical (kind=log_kind) :: l_brine, heat_capacity, calc_Tsfc

      contains

!=======================================================================
      subroutine init_thermo_vertical
      use ice_state, only: nt_Tsfc, nt_iage
      use ice_age, only: tr_iage
      type (cice_thermo_config) :: cfg
      real (kind=dbl_kind), parameter :: min_salin = 0.1_dbl_kind

      l_brine = (saltmax > min_salin .and. heat_capacity)
      cfg%heat_capacity = merge(1, 0, heat_capacity)
      cfg%calc_Tsfc = merge(1, 0, calc_Tsfc)
      cfg%conduct = merge(0, 1, trim(conduct) == 'MU71')
      cfg%ustar_min = ustar_min
      cfg%tr_iage = merge(1, 0, tr_iage)
      cfg%nt_Tsfc = nt_Tsfc
      cfg%nt_iage = nt_iage
      call cice_gpu_ensure()
      call cice_gpu_check(cice_check_sizes(cice_gpu_ctx, ncat, nilyr, nslyr, max_ntrcr), 'init_thermo_vertical')
      ! the device side computes the salinity / melting-temperature profile and hands it back,
      ! so that the host copies used by ice_init / ice_history stay consistent with it
      call cice_gpu_check(cice_thermo_init(cice_gpu_ctx, cfg, salin, Tmlt), 'init_thermo_vertical')
      end subroutine init_thermo_vertical

!=======================================================================
      subroutine thermo_vertical (nx_block, ny_block, dt, icells, indxi, indxj, &
                                  aicen, trcrn, vicen, vsnon, eicen, esnon, &
                                  flw, potT, Qa, rhoa, fsnow, fbot, Tbot, lhcoef, shcoef, &
                                  fswsfc, fswint, fswthrun, Sswabs, Iswabs, &
                                  fsurfn, fcondtopn, fsensn, flatn, fswabsn, flwoutn, &
                                  evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, &
                                  congel, snoice, mlt_onset, frz_onset, yday, &
                                  l_stop, istop, jstop)
      integer (kind=int_kind), intent(in) :: nx_block, ny_block, icells
      integer (kind=int_kind), dimension (nx_block*ny_block), intent(in) :: indxi, indxj
      real (kind=dbl_kind), intent(in) :: dt
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(inout) :: aicen, vicen, vsnon
      real (kind=dbl_kind), dimension (nx_block,ny_block,max_ntrcr), intent(inout) :: trcrn
      real (kind=dbl_kind), dimension (nx_block,ny_block,nilyr), intent(inout) :: eicen
      real (kind=dbl_kind), dimension (nx_block,ny_block,nslyr), intent(inout) :: esnon
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(in) :: &
         flw, potT, Qa, rhoa, fsnow, shcoef, lhcoef, fbot, Tbot
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(inout) :: fswsfc, fswint, fswthrun
      real (kind=dbl_kind), dimension (nx_block,ny_block,nslyr), intent(inout) :: Sswabs
      real (kind=dbl_kind), dimension (nx_block,ny_block,nilyr), intent(inout) :: Iswabs
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(out) :: &
         fsensn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(inout) :: &
         flatn, fsurfn, fcondtopn, meltt, melts, meltb, congel, snoice, mlt_onset, frz_onset
      real (kind=dbl_kind), intent(in) :: yday
      logical (kind=log_kind), intent(out) :: l_stop
      integer (kind=int_kind), intent(out) :: istop, jstop
      integer (c_int) :: ls

      call cice_gpu_check(cice_thermo_vertical(cice_gpu_ctx, nx_block, ny_block, dt, icells, &
         indxi, indxj, aicen, trcrn, vicen, vsnon, eicen, esnon, flw, potT, Qa, rhoa, fsnow, &
         fbot, Tbot, lhcoef, shcoef, fswsfc, fswint, fswthrun, Sswabs, Iswabs, fsurfn, fcondtopn, &
         fsensn, flatn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, &
         congel, snoice, mlt_onset, frz_onset, yday, ls, istop, jstop), 'thermo_vertical')
      l_stop = (ls /= 0)
      if (l_stop) write(nu_diag,*) 'thermo_vertical (GPU): column failed at i, j =', istop, jstop
      end subroutine thermo_vertical

!=======================================================================
      subroutine frzmlt_bottom_lateral (nx_block, ny_block, ilo, ihi, jlo, jhi, dt, &
                                        aice, frzmlt, eicen, esnon, sst, Tf, &
                                        strocnxT, strocnyT, Tbot, fbot, rside)
      integer (kind=int_kind), intent(in) :: nx_block, ny_block, ilo, ihi, jlo, jhi
      real (kind=dbl_kind), intent(in) :: dt
      real (kind=dbl_kind), dimension(nx_block,ny_block), intent(in) :: &
         aice, frzmlt, sst, Tf, strocnxT, strocnyT
      real (kind=dbl_kind), dimension(nx_block,ny_block,ntilyr), intent(in) :: eicen
      real (kind=dbl_kind), dimension(nx_block,ny_block,ntslyr), intent(in) :: esnon
      real (kind=dbl_kind), dimension(nx_block,ny_block), intent(out) :: Tbot, fbot, rside

#if defined(AusCOM)
      call cice_gpu_check(cice_thermo_set_chio(cice_gpu_ctx, chio), 'cice_thermo_set_chio')
#endif
      call cice_gpu_check(cice_frzmlt_bottom_lateral(cice_gpu_ctx, nx_block, ny_block, ilo, ihi, &
         jlo, jhi, dt, aice, frzmlt, eicen, esnon, sst, Tf, strocnxT, strocnyT, Tbot, fbot, rside), &
         'frzmlt_bottom_lateral')
      end subroutine frzmlt_bottom_lateral

!=======================================================================
! Host-side helper kept because the stand-alone driver's explicit_calc_Tsfc (CICE_RunMod.F90:1216,
! only reached with calc_Tsfc = F) calls it by name with this argument list: radiative and turbulent
! surface fluxes and their Tsf derivatives for the first `isolve` entries of the (i, j, m) lists.
      subroutine surface_fluxes (nx_block, ny_block, isolve, icells, indxii, indxjj, indxij, &
                                 Tsf, fswsfc, rhoa, flw, potT, Qa, shcoef, lhcoef, &
                                 flwoutn, fsensn, flatn, fsurfn, &
                                 dflwout_dT, dfsens_dT, dflat_dT, dfsurf_dT)
      integer (kind=int_kind), intent(in) :: nx_block, ny_block, isolve, icells
      integer (kind=int_kind), dimension(icells), intent(in) :: indxii, indxjj
      integer (kind=int_kind), dimension(icells) :: indxij
      real (kind=dbl_kind), dimension(icells), intent(in) :: Tsf
      real (kind=dbl_kind), dimension(nx_block,ny_block), intent(in) :: &
         fswsfc, rhoa, flw, potT, Qa, shcoef, lhcoef
      real (kind=dbl_kind), dimension(nx_block,ny_block), intent(inout) :: &
         fsensn, flatn, flwoutn, fsurfn
      real (kind=dbl_kind), dimension(icells), intent(inout) :: dfsens_dT, dflat_dT, dflwout_dT
      real (kind=dbl_kind), dimension(isolve), intent(inout) :: dfsurf_dT
      integer (kind=int_kind) :: n, i, j, m
      real (kind=dbl_kind) :: TK, rTK, Qs, es

      es = emissivity*stefan_boltzmann
      do n = 1, isolve
         i = indxii(n);  j = indxjj(n);  m = indxij(n)
         TK  = Tsf(m) + Tffresh
         rTK = c1/TK
         Qs  = qqqice*exp(-TTTice*rTK) / rhoa(i,j)
         flwoutn(i,j) = -es * TK**4
         fsensn(i,j)  = shcoef(i,j) * (potT(i,j) - TK)
         flatn(i,j)   = lhcoef(i,j) * (Qa(i,j) - Qs)
         dflwout_dT(m) = -es * c4*TK**3
         dfsens_dT(m)  = -shcoef(i,j)
         dflat_dT(m)   = -lhcoef(i,j) * (TTTice*rTK*rTK*Qs)
         fsurfn(i,j)   = fswsfc(i,j) + emissivity*flw(i,j) + flwoutn(i,j) + fsensn(i,j) + flatn(i,j)
         dfsurf_dT(n)  = dflwout_dT(m) + dfsens_dT(m) + dflat_dT(m)
      enddo
      end subroutine surface_fluxes

!=======================================================================
! Host-side helper kept for ice_history (enthalpy -> temperature, quadratic formula).
      function calculate_Tin_from_qin (qin, Tmltk) result(Tin)
      real (kind=dbl_kind), intent(in) :: qin, Tmltk
      real (kind=dbl_kind) :: Tin, aa1, bb1, cc1
      if (l_brine) then
         aa1 = cp_ice
         bb1 = (cp_ocn-cp_ice)*Tmltk - qin/rhoi - Lfresh
         cc1 = Lfresh * Tmltk
         Tin = (-bb1 - sqrt(bb1*bb1 - c4*aa1*cc1)) / (c2*aa1)
      else
         Tin = (Lfresh + qin/rhoi) / cp_ice
      endif
      end function calculate_Tin_from_qin

      end module ice_therm_vertical

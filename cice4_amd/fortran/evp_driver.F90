!=======================================================================
! Stand-alone Fortran host driver for the GPU EVP path (own code; needs no
! reference module).  Shows the calling sequence a Fortran sea-ice model
! uses: cice_create -> cice_domain_create -> cice_evp_init (once) ->
! cice_evp (every dynamics step), all through the ISO_C_BINDING shim.
!
!   evp_driver <input.bin> <output.bin> [nsteps]
!
! input (stream, native endian): nxg nyg bsx bsy ew ns ndte (int32), dt (real64),
!   then for the local blocks, each (nx_block,ny_block,nblocks):
!   14 grid arrays (cice_evp_grid order), tmask, umask (int32),
!   12 "in" arrays (aicen, vicen with ncat levels), uvel, vvel, 12 stresses,
!   iceumask (int32), fm, strtltx, strtlty, strocnx, strocny, strintx, strinty.
! output: uvel, vvel, 12 stresses, iceumask, the 7 io and the 10 out arrays.
!=======================================================================
program evp_driver
   use iso_c_binding
   use cice4_amd_c
   implicit none
   integer(c_int) :: nxg, nyg, bsx, bsy, ew, ns, ndte, info(9), nsteps, k, istep
   real(c_double) :: dt
   integer :: nx, ny, nb, n, u
   character(len=512) :: fin, fout, arg
   real(c_double), allocatable, target :: g(:,:), fin2(:,:), aicen(:), vicen(:), io(:,:), outp(:,:)
   integer(c_int), allocatable, target :: tmask(:), umask(:), iceumask(:)
   type(cice_evp_grid) :: grid
   type(cice_evp_config) :: cfg
   type(cice_evp_fields) :: f
   integer, parameter :: ncat = 5

   call get_command_argument(1, fin)
   call get_command_argument(2, fout)
   nsteps = 1
   if (command_argument_count() >= 3) then
      call get_command_argument(3, arg)
      read(arg, *) nsteps
   endif
   open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
   read(u) nxg, nyg, bsx, bsy, ew, ns, ndte
   read(u) dt

   call cice_gpu_ensure()
   call cice_gpu_check(cice_domain_create(cice_gpu_ctx, nxg, nyg, bsx, bsy, ew, ns, 0_c_int, 1_c_int, 1_c_int), &
                       'cice_domain_create')
   call cice_gpu_check(cice_domain_info(cice_gpu_ctx, info), 'cice_domain_info')
   nx = info(1); ny = info(2); nb = info(3); n = nx*ny*nb

   allocate(g(n,14), tmask(n), umask(n), fin2(n,10), aicen(n*ncat), vicen(n*ncat), io(n,21), &
            iceumask(n), outp(n,10))
   read(u) g
   read(u) tmask, umask
   read(u) fin2(:,1:4)            ! aice vice vsno aice0
   read(u) aicen, vicen
   read(u) fin2(:,5:10)           ! strairxT strairyT uocn vocn ss_tltx ss_tlty
   read(u) io(:,1:14)             ! uvel vvel + 12 stresses
   read(u) iceumask
   read(u) io(:,15:21)            ! fm strtltx strtlty strocnx strocny strintx strinty
   close(u)
   outp = 0.0d0

   grid%dxt = c_loc(g(1,1)); grid%dyt = c_loc(g(1,2)); grid%dxhy = c_loc(g(1,3)); grid%dyhx = c_loc(g(1,4))
   grid%cxp = c_loc(g(1,5)); grid%cyp = c_loc(g(1,6)); grid%cxm = c_loc(g(1,7)); grid%cym = c_loc(g(1,8))
   grid%tarea = c_loc(g(1,9)); grid%uarea = c_loc(g(1,10)); grid%tarear = c_loc(g(1,11))
   grid%uarear = c_loc(g(1,12)); grid%tinyarea = c_loc(g(1,13)); grid%fcor = c_loc(g(1,14))
   grid%tmask = c_loc(tmask); grid%umask = c_loc(umask)
   grid%HTN = c_null_ptr; grid%HTE = c_null_ptr
   cfg%ndte = ndte; cfg%evp_damping = 0
   cfg%kstrength = 1; cfg%krdg_partic = 1; cfg%krdg_redist = 1; cfg%mu_rdg = 4.0d0
   call cice_gpu_check(cice_evp_init(cice_gpu_ctx, cfg, grid), 'cice_evp_init')

   f%aice = c_loc(fin2(1,1)); f%vice = c_loc(fin2(1,2)); f%vsno = c_loc(fin2(1,3)); f%aice0 = c_loc(fin2(1,4))
   f%aicen = c_loc(aicen); f%vicen = c_loc(vicen)
   f%strairxT = c_loc(fin2(1,5)); f%strairyT = c_loc(fin2(1,6)); f%uocn = c_loc(fin2(1,7))
   f%vocn = c_loc(fin2(1,8)); f%ss_tltx = c_loc(fin2(1,9)); f%ss_tlty = c_loc(fin2(1,10))
   f%uvel = c_loc(io(1,1)); f%vvel = c_loc(io(1,2))
   f%stressp_1 = c_loc(io(1,3)); f%stressp_2 = c_loc(io(1,4)); f%stressp_3 = c_loc(io(1,5))
   f%stressp_4 = c_loc(io(1,6)); f%stressm_1 = c_loc(io(1,7)); f%stressm_2 = c_loc(io(1,8))
   f%stressm_3 = c_loc(io(1,9)); f%stressm_4 = c_loc(io(1,10)); f%stress12_1 = c_loc(io(1,11))
   f%stress12_2 = c_loc(io(1,12)); f%stress12_3 = c_loc(io(1,13)); f%stress12_4 = c_loc(io(1,14))
   f%iceumask = c_loc(iceumask)
   f%fm = c_loc(io(1,15)); f%strtltx = c_loc(io(1,16)); f%strtlty = c_loc(io(1,17))
   f%strocnx = c_loc(io(1,18)); f%strocny = c_loc(io(1,19)); f%strintx = c_loc(io(1,20))
   f%strinty = c_loc(io(1,21))
   f%strairx = c_loc(outp(1,1)); f%strairy = c_loc(outp(1,2)); f%strength = c_loc(outp(1,3))
   f%divu = c_loc(outp(1,4)); f%shear = c_loc(outp(1,5)); f%rdg_conv = c_loc(outp(1,6))
   f%rdg_shear = c_loc(outp(1,7)); f%prs_sig = c_loc(outp(1,8)); f%strocnxT = c_loc(outp(1,9))
   f%strocnyT = c_loc(outp(1,10))

   do istep = 1, nsteps
      call cice_gpu_check(cice_evp(cice_gpu_ctx, dt, f), 'cice_evp')
   enddo

   open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
   write(u) io(:,1:14)
   write(u) iceumask
   write(u) io(:,15:21)
   write(u) outp
   close(u)
   write(*,'(a,i0,a,3i6,a,es22.14)') 'evp_driver: ', nsteps, ' step(s) on blocks ', nx, ny, nb, &
        '  max|uvel| = ', maxval(abs(io(:,1)))
   k = cice_destroy(cice_gpu_ctx)
end program evp_driver

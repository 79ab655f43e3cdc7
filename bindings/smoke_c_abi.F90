!> Stand-alone Fortran smoke driver of the C ABI (no reference sources needed): builds a config by
!> hand, uploads an rxff.bin record block read from a file, runs QEq + FORCE + 2 MD steps.
!>   amdflang -DSTANDALONE -cpp bindings/smoke_c_abi.F90 -o smoke -L rxmd_amd -lrxmd_hip -Wl,-rpath,$PWD/rxmd_amd
program smoke_c_abi
  use iso_c_binding
  implicit none
  type, bind(c) :: rxmd_config
     type(c_ptr) :: ffield_path
     real(c_double) :: lattice(6)
     integer(c_int) :: vprocs(3), myid, isQEq, NMAXQEq
     real(c_double) :: QEq_tol
     integer(c_int) :: qstep
     real(c_double) :: dt_fs, Lex_fqs, Lex_k
     integer(c_int) :: nbuffer, maxneighbs, maxneighbs10, device, qeq_mode, lg
     type(c_ptr) :: pqeq_path
     integer(c_int) :: efield_dir, reserved1
     real(c_double) :: efield_strength
  end type
  interface
     subroutine rxmd_hip_default_config(cfg) bind(c, name='rxmd_hip_default_config')
       import; type(rxmd_config), intent(out) :: cfg
     end subroutine
     integer(c_int) function rxmd_hip_create(cfg, h) bind(c, name='rxmd_hip_create')
       import; type(rxmd_config), intent(in) :: cfg; type(c_ptr), intent(out) :: h
     end function
     integer(c_int) function rxmd_hip_set_atoms_rxff(h, n, rec) bind(c, name='rxmd_hip_set_atoms_rxff')
       import; type(c_ptr), value :: h; integer(c_int), value :: n; real(c_double), intent(in) :: rec(*)
     end function
     integer(c_int) function rxmd_hip_qeq(h, it, est) bind(c, name='rxmd_hip_qeq')
       import; type(c_ptr), value :: h; integer(c_int), intent(out) :: it; real(c_double), intent(out) :: est
     end function
     integer(c_int) function rxmd_hip_force(h, pe) bind(c, name='rxmd_hip_force')
       import; type(c_ptr), value :: h; real(c_double), intent(out) :: pe(0:13)
     end function
     integer(c_int) function rxmd_hip_step(h, n) bind(c, name='rxmd_hip_step')
       import; type(c_ptr), value :: h; integer(c_int), value :: n
     end function
     integer(c_int) function rxmd_hip_destroy(h) bind(c, name='rxmd_hip_destroy')
       import; type(c_ptr), value :: h
     end function
  end interface
  type(rxmd_config) :: cfg
  type(c_ptr) :: h
  character(kind=c_char, len=:), allocatable, target :: ffp
  character(len=512) :: ffarg, binarg
  integer(c_int) :: rc, it, np, vp(3), nat(1), cur
  real(c_double) :: est, pe(0:13), lat(6)
  real(c_double), allocatable :: rec(:)
  call get_command_argument(1, ffarg); call get_command_argument(2, binarg)
  open(11, file=trim(binarg), form='unformatted', access='stream', status='old')
  read(11) np, vp, nat, cur, lat           ! rxff.bin header of a 1-rank file (src/fileio.F90:465-494)
  allocate(rec(10*nat(1))); read(11) rec; close(11)
  call rxmd_hip_default_config(cfg)
  ffp = trim(ffarg)//c_null_char
  cfg%ffield_path = c_loc(ffp); cfg%lattice = lat; cfg%QEq_tol = 1d-12; cfg%NMAXQEq = 2000
  rc = rxmd_hip_create(cfg, h); if (rc /= 0) stop 2
  rc = rxmd_hip_set_atoms_rxff(h, nat(1), rec); if (rc /= 0) stop 3
  rc = rxmd_hip_qeq(h, it, est); if (rc /= 0) stop 4
  rc = rxmd_hip_force(h, pe); if (rc /= 0) stop 5
  write(6,'(a,i6,a,i5,a,es25.16,a,es25.16)') 'fortran-abi natoms', nat(1), ' qeq_iters', it, ' Est', est, ' PE', pe(0)
  rc = rxmd_hip_step(h, 2); if (rc /= 0) stop 6
  rc = rxmd_hip_destroy(h)
  write(6,'(a)') 'fortran-abi ok'
end program

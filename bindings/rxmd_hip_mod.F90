!> rxmd_hip_mod -- iso_c_binding interface of the MI355X ReaxFF+QEq engine (include/rxmd_hip.h) and
!> drop-in replacements for the reference's three entry points of the hot path:
!>
!>     QEq(atype,pos,q)      USCCACS/RXMD src/qeq.F90:2      ->  QEq_hip
!>     FORCE(atype,pos,f,q)  USCCACS/RXMD src/pot.F90:2      ->  FORCE_hip
!>     MD loop body          USCCACS/RXMD src/main.F90:64-98 ->  rxmd_hip_step (device resident)
!>
!> The wrappers take the SAME arguments as the reference subroutines (atype(NBUFFER) packed
!> type+gid*1e-13, pos/f(NBUFFER,3) column major, q(NBUFFER)), so `call QEq(atype,pos,q)` in
!> main.F90:30,81 becomes `call QEq_hip(atype,pos,q)` and nothing else changes in the driver.
!> Multi-rank (vprocs > 1 under the reference's MPI driver, examples/2-reaxff-dc: `mpirun -np 2 rxmd`, comm.F90:291-364):
!> rxmd_hip_init installs a transport for the engine's six-stage exchange and its CG all-reduces --
!>   * one GPU per rank: the engine's own RCCL communicator (ncclSend/ncclRecv/ncclAllReduce on its stream); MPI only broadcasts the
!>     128-byte unique id (rxmd_hip_rccl_unique_id -> MPI_BCAST -> rxmd_hip_comm_init_rccl);
!>   * fewer GPUs than ranks, or RXMD_HIP_TRANSPORT=mpi: MPI_SENDRECV / MPI_ALLREDUCE callbacks bound to rxmd_comm_ops, every message
!>     staged through host memory (no GPU-aware MPI needed).
!> Compiled with -DNOMPI (the reference's serial build) the transport is left out.
!> Build:  amdflang -cpp [-DNOMPI | -I<mpi include>] -c bindings/rxmd_hip_mod.F90 ; link with -L rxmd_amd -lrxmd_hip
module rxmd_hip_mod
  use iso_c_binding
  implicit none
#ifndef NOMPI
  include 'mpif.h'
#endif
  private
  public :: rxmd_config, rxmd_hip_init, rxmd_hip_finalize, QEq_hip, PQEq_hip, FORCE_hip, rxmd_hip_handle
  public :: rxmd_hip_create, rxmd_hip_destroy, rxmd_hip_qeq, rxmd_hip_force, rxmd_hip_step, rxmd_hip_set_atoms_rxff, &
            rxmd_hip_get_atoms_rxff, rxmd_hip_last_error, rxmd_hip_default_config, rxmd_hip_qeq_arrays, rxmd_hip_force_arrays

  !> mirrors `struct rxmd_config` (include/rxmd_hip.h)
  type, bind(c) :: rxmd_config
     type(c_ptr)    :: ffield_path
     real(c_double) :: lattice(6)
     integer(c_int) :: vprocs(3)
     integer(c_int) :: myid
     integer(c_int) :: isQEq
     integer(c_int) :: NMAXQEq
     real(c_double) :: QEq_tol
     integer(c_int) :: qstep
     real(c_double) :: dt_fs
     real(c_double) :: Lex_fqs, Lex_k
     integer(c_int) :: nbuffer, maxneighbs, maxneighbs10, device, qeq_mode
     integer(c_int) :: lg
     type(c_ptr)    :: pqeq_path
     integer(c_int) :: efield_dir, reserved1
     real(c_double) :: efield_strength
  end type

  type(c_ptr), save :: rxmd_hip_handle = c_null_ptr

  !> mirrors `struct rxmd_comm_ops` (include/rxmd_hip.h)
  type, bind(c) :: rxmd_comm_ops
     type(c_ptr)    :: ctx
     type(c_funptr) :: exchange, allreduce_sum, exchange_known
  end type
  type(rxmd_comm_ops), save, target :: mpi_ops
  real(c_double), allocatable, save, target :: mpi_sbuf(:), mpi_rbuf(:)     ! host staging of one message each way

  interface
     subroutine rxmd_hip_default_config(cfg) bind(c, name='rxmd_hip_default_config')
       import :: rxmd_config
       type(rxmd_config), intent(out) :: cfg
     end subroutine
     integer(c_int) function rxmd_hip_create(cfg, h) bind(c, name='rxmd_hip_create')
       import :: rxmd_config, c_ptr, c_int
       type(rxmd_config), intent(in) :: cfg
       type(c_ptr), intent(out) :: h
     end function
     integer(c_int) function rxmd_hip_destroy(h) bind(c, name='rxmd_hip_destroy')
       import :: c_ptr, c_int
       type(c_ptr), value :: h
     end function
     type(c_ptr) function rxmd_hip_last_error(h) bind(c, name='rxmd_hip_last_error')
       import :: c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function rxmd_hip_set_atoms_rxff(h, natoms, rec10) bind(c, name='rxmd_hip_set_atoms_rxff')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: natoms
       real(c_double), intent(in) :: rec10(*)
     end function
     integer(c_int) function rxmd_hip_get_atoms_rxff(h, rec10, capacity) bind(c, name='rxmd_hip_get_atoms_rxff')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(out) :: rec10(*)
       integer(c_int), value :: capacity
     end function
     integer(c_int) function rxmd_hip_qeq(h, iters, est) bind(c, name='rxmd_hip_qeq')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), intent(out) :: iters
       real(c_double), intent(out) :: est
     end function
     integer(c_int) function rxmd_hip_force(h, pe) bind(c, name='rxmd_hip_force')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(out) :: pe(0:13)
     end function
     integer(c_int) function rxmd_hip_step(h, nsteps) bind(c, name='rxmd_hip_step')
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: nsteps
     end function
     integer(c_int) function rxmd_hip_qeq_arrays(h, nbuffer, natoms, atype, pos, q) bind(c, name='rxmd_hip_QEq')   ! Fortran is case-blind
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: nbuffer, natoms
       real(c_double), intent(in) :: atype(*), pos(*)
       real(c_double), intent(inout) :: q(*)
     end function
     integer(c_int) function rxmd_hip_force_arrays(h, nbuffer, natoms, atype, pos, f, q, pe) bind(c, name='rxmd_hip_FORCE')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: nbuffer, natoms
       real(c_double), intent(in) :: atype(*), pos(*), q(*)
       real(c_double), intent(out) :: f(*), pe(0:13)
     end function
     integer(c_int) function rxmd_hip_pqeq_arrays(h, nbuffer, natoms, atype, pos, q, spos) bind(c, name='rxmd_hip_PQEq')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: nbuffer, natoms
       real(c_double), intent(in) :: atype(*), pos(*)
       real(c_double), intent(inout) :: q(*), spos(*)
     end function
     integer(c_int) function rxmd_hip_force_pqeq_arrays(h, nbuffer, natoms, atype, pos, f, q, spos, pe) bind(c, name='rxmd_hip_FORCE_pqeq')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: nbuffer, natoms
       real(c_double), intent(in) :: atype(*), pos(*), q(*), spos(*)
       real(c_double), intent(out) :: f(*), pe(0:13)
     end function
     integer(c_int) function rxmd_hip_set_comm(h, ops) bind(c, name='rxmd_hip_set_comm')
       import :: c_ptr, c_int, rxmd_comm_ops
       type(c_ptr), value :: h
       type(rxmd_comm_ops), intent(in) :: ops
     end function
     integer(c_int) function rxmd_hip_rccl_unique_id(id128) bind(c, name='rxmd_hip_rccl_unique_id')
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id128(128)
     end function
     integer(c_int) function rxmd_hip_comm_init_rccl(h, id128, rank, world) bind(c, name='rxmd_hip_comm_init_rccl')
       import :: c_ptr, c_int, c_char
       type(c_ptr), value :: h
       character(kind=c_char), intent(in) :: id128(128)
       integer(c_int), value :: rank, world
     end function
     integer(c_int) function rxmd_hip_copy_to_host(dev, host, n) bind(c, name='rxmd_hip_copy_to_host')
       import :: c_ptr, c_int, c_double, c_long_long
       type(c_ptr), value :: dev
       real(c_double), intent(out) :: host(*)
       integer(c_long_long), value :: n
     end function
     integer(c_int) function rxmd_hip_copy_to_device(dev, host, n) bind(c, name='rxmd_hip_copy_to_device')
       import :: c_ptr, c_int, c_double, c_long_long
       type(c_ptr), value :: dev
       real(c_double), intent(in) :: host(*)
       integer(c_long_long), value :: n
     end function
     integer(c_int) function rxmd_hip_device_count() bind(c, name='rxmd_hip_device_count')
       import :: c_int
     end function
     integer(c_int) function rxmd_hip_put_lex(h, natoms, qsfp, qsfv) bind(c, name='rxmd_hip_put_lex')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: natoms
       real(c_double), intent(in) :: qsfp(*), qsfv(*)
     end function
     integer(c_int) function rxmd_hip_get_lex(h, natoms, qsfp, qsfv) bind(c, name='rxmd_hip_get_lex')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: natoms
       real(c_double), intent(out) :: qsfp(*), qsfv(*)
     end function
     integer(c_int) function rxmd_hip_last_qeq_iters(h) bind(c, name='rxmd_hip_last_qeq_iters')
       import :: c_ptr, c_int
       type(c_ptr), value :: h
     end function
     integer(c_int) function rxmd_hip_get_energy(h, ke, qsum, pe, astr6) bind(c, name='rxmd_hip_get_energy')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(out) :: ke, qsum, pe(0:13), astr6(6)
     end function
  end interface

contains

  !> call once after GETPARAMS/INITSYSTEM (reference src/main.F90:20-23) with the module-global run parameters
  subroutine rxmd_hip_init(ffpath, lata, latb, latc, lalpha, lbeta, lgamma, vprocs, myid, isQEq, NMAXQEq, QEq_tol, qstep, dt_fs, device, pqeqpath)
    use atoms, only: isLG, isEfield, eFieldDir, eFieldStrength   ! --lg (cmdline.F90:148-151); rxmd.in `efield` (cmdline.F90:131-137,286-289)
    character(len=*), intent(in) :: ffpath
    character(len=*), intent(in), optional :: pqeqpath          ! PQEqParmPath when isPQEq (cmdline.F90:112-128)
    character(kind=c_char, len=:), allocatable, target, save :: cpq
    real(8), intent(in) :: lata, latb, latc, lalpha, lbeta, lgamma, QEq_tol, dt_fs
    integer, intent(in) :: vprocs(3), myid, isQEq, NMAXQEq, qstep, device
    type(rxmd_config) :: cfg
    character(kind=c_char, len=:), allocatable, target, save :: cpath
    integer(c_int) :: rc
    call rxmd_hip_default_config(cfg)
    cpath = trim(ffpath)//c_null_char
    cfg%ffield_path = c_loc(cpath)
    cfg%lattice = (/lata, latb, latc, lalpha, lbeta, lgamma/)
    cfg%vprocs = vprocs; cfg%myid = myid
    cfg%isQEq = isQEq; cfg%NMAXQEq = NMAXQEq; cfg%QEq_tol = QEq_tol; cfg%qstep = qstep; cfg%dt_fs = dt_fs
    cfg%device = device
    if (device < 0) cfg%device = mod(myid, max(1, rxmd_hip_device_count()))     ! one GPU per rank of the node, round robin
    if (isLG) cfg%lg = 1
    if (isEfield) then                                          ! field force on cores and shells inside FORCE / PQEq (pot.F90:61, pqeq.F90:205)
       cfg%efield_dir = eFieldDir; cfg%efield_strength = eFieldStrength
    endif
    if (present(pqeqpath)) then
       if (len_trim(pqeqpath) > 0) then
          cpq = trim(pqeqpath)//c_null_char
          cfg%pqeq_path = c_loc(cpq)
       endif
    endif
    rc = rxmd_hip_create(cfg, rxmd_hip_handle)
    if (rc /= 0) call die('rxmd_hip_create', rc)
#ifndef NOMPI
    if (vprocs(1)*vprocs(2)*vprocs(3) > 1) call install_transport(myid, vprocs(1)*vprocs(2)*vprocs(3))
#endif
  end subroutine

#ifndef NOMPI
  !> the transport of a vprocs > 1 run (see the module header)
  subroutine install_transport(myid, nprocs)
    integer, intent(in) :: myid, nprocs
    character(kind=c_char) :: id(128)
    character(len=16) :: want
    integer :: ierr, stat
    integer(c_int) :: rc
    logical :: use_rccl
    call get_environment_variable('RXMD_HIP_TRANSPORT', want, status=stat)
    if (stat /= 0) want = ' '
    use_rccl = rxmd_hip_device_count() >= nprocs                ! RCCL wants one device per rank
    if (trim(want) == 'mpi') use_rccl = .false.
    if (trim(want) == 'rccl') use_rccl = .true.
    if (use_rccl) then
       id = c_null_char
       if (myid == 0) then
          rc = rxmd_hip_rccl_unique_id(id)
          if (rc /= 0) call die('rxmd_hip_rccl_unique_id', rc)
       endif
       call MPI_BCAST(id, 128, MPI_CHARACTER, 0, MPI_COMM_WORLD, ierr)
       rc = rxmd_hip_comm_init_rccl(rxmd_hip_handle, id, int(myid, c_int), int(nprocs, c_int))
       if (rc /= 0) call die('rxmd_hip_comm_init_rccl', rc)
       if (myid == 0) write(6,'(a,i4,a)') 'rxmd_hip: RCCL transport over ', nprocs, ' ranks (one GPU each)'
    else
       mpi_ops%ctx = c_null_ptr
       mpi_ops%exchange = c_funloc(mpi_exchange)
       mpi_ops%allreduce_sum = c_funloc(mpi_allreduce_sum)
       mpi_ops%exchange_known = c_funloc(mpi_exchange_known)
       rc = rxmd_hip_set_comm(rxmd_hip_handle, mpi_ops)
       if (rc /= 0) call die('rxmd_hip_set_comm', rc)
       if (myid == 0) write(6,'(a,i4,a)') 'rxmd_hip: host-staged MPI transport over ', nprocs, ' ranks'
    endif
  end subroutine

  subroutine need_host_buffers(ns, nr)
    integer(c_long_long), intent(in) :: ns, nr
    if (.not. allocated(mpi_sbuf)) allocate(mpi_sbuf(max(ns, 65536_c_long_long)))
    if (size(mpi_sbuf, kind=c_long_long) < ns) then
       deallocate(mpi_sbuf); allocate(mpi_sbuf(ns + ns/4))
    endif
    if (.not. allocated(mpi_rbuf)) allocate(mpi_rbuf(max(nr, 65536_c_long_long)))
    if (size(mpi_rbuf, kind=c_long_long) < nr) then
       deallocate(mpi_rbuf); allocate(mpi_rbuf(nr + nr/4))
    endif
  end subroutine

  !> rxmd_comm_ops.exchange: nsend doubles at device pointer `send` go to rank `to`; whatever rank `from` sends lands at device
  !> pointer `recv` (capacity cap); returns the number received.  One send_recv of comm.F90:291-364: size first, then payload.
  function mpi_exchange(ctx, to, send, nsend, from, recv, cap) bind(c) result(nrecv)
    type(c_ptr), value :: ctx, send, recv
    integer(c_int), value :: to, from
    integer(c_long_long), value :: nsend, cap
    integer(c_long_long) :: nrecv
    integer(c_long_long) :: ns8(1), nr8(1)
    integer :: ierr, stat(MPI_STATUS_SIZE)
    ns8(1) = nsend
    call MPI_SENDRECV(ns8, 1, MPI_INTEGER8, to, 20, nr8, 1, MPI_INTEGER8, from, 20, MPI_COMM_WORLD, stat, ierr)
    nrecv = nr8(1)
    if (ierr /= MPI_SUCCESS .or. nrecv > cap) then
       nrecv = -1
       return
    endif
    nrecv = mpi_exchange_known(ctx, to, send, nsend, from, recv, nrecv)
  end function

  !> rxmd_comm_ops.exchange_known: the same when the receiver already knows the count (vector halos, force return)
  function mpi_exchange_known(ctx, to, send, nsend, from, recv, nrecv_in) bind(c) result(nrecv)
    type(c_ptr), value :: ctx, send, recv
    integer(c_int), value :: to, from
    integer(c_long_long), value :: nsend, nrecv_in
    integer(c_long_long) :: nrecv
    integer :: ierr, stat(MPI_STATUS_SIZE)
    integer(c_int) :: rc
    nrecv = nrecv_in
    call need_host_buffers(max(nsend, 1_c_long_long), max(nrecv, 1_c_long_long))
    rc = rxmd_hip_copy_to_host(send, mpi_sbuf, nsend)
    if (rc /= 0) then
       nrecv = -1
       return
    endif
    call MPI_SENDRECV(mpi_sbuf, int(nsend), MPI_DOUBLE_PRECISION, to, 21, mpi_rbuf, int(nrecv), MPI_DOUBLE_PRECISION, from, 21, &
                      MPI_COMM_WORLD, stat, ierr)
    if (ierr /= MPI_SUCCESS) then
       nrecv = -1
       return
    endif
    rc = rxmd_hip_copy_to_device(recv, mpi_rbuf, nrecv)
    if (rc /= 0) nrecv = -1
  end function

  !> rxmd_comm_ops.allreduce_sum: n host doubles summed in place over all ranks (MPI_ALLREDUCE of qeq.F90:107,129,144,357)
  function mpi_allreduce_sum(ctx, buf, n) bind(c) result(rc)
    type(c_ptr), value :: ctx
    integer(c_int), value :: n
    real(c_double), intent(inout) :: buf(n)
    integer(c_int) :: rc
    integer :: ierr
    call MPI_ALLREDUCE(MPI_IN_PLACE, buf, int(n), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
    rc = merge(0_c_int, 1_c_int, ierr == MPI_SUCCESS)
  end function
#endif

  subroutine rxmd_hip_finalize()
    integer(c_int) :: rc
    if (c_associated(rxmd_hip_handle)) rc = rxmd_hip_destroy(rxmd_hip_handle)
    rxmd_hip_handle = c_null_ptr
  end subroutine

  !> same arguments as the reference's QEq(atype,pos,q), src/qeq.F90:2,15-16
  subroutine QEq_hip(atype, pos, q)
    use atoms, only: NBUFFER, NATOMS, nstep_qeq, qsfp, qsfv
    real(8), intent(in) :: atype(NBUFFER), pos(NBUFFER,3)
    real(8), intent(inout) :: q(NBUFFER)
    integer(c_int) :: rc
    ! the fictitious charges of module atoms travel with the call (read with isQEq = 2, written with isQEq = 1: qeq.F90:41-42,51-52)
    rc = rxmd_hip_put_lex(rxmd_hip_handle, int(NATOMS, c_int), qsfp, qsfv)
    if (rc /= 0) call die('QEq (put_lex)', rc)
    rc = rxmd_hip_qeq_arrays(rxmd_hip_handle, int(NBUFFER, c_int), int(NATOMS, c_int), atype, pos, q)
    if (rc /= 0) call die('QEq', rc)
    rc = rxmd_hip_get_lex(rxmd_hip_handle, int(NATOMS, c_int), qsfp, qsfv)
    if (rc /= 0) call die('QEq (get_lex)', rc)
    nstep_qeq = rxmd_hip_last_qeq_iters(rxmd_hip_handle)        ! printed by PRINTE, src/main.F90:261
  end subroutine

  !> same arguments as the reference's PQEq(atype,pos,q), src/pqeq.F90:2; the shell displacements are module atoms' spos
  subroutine PQEq_hip(atype, pos, q)
    use atoms, only: NBUFFER, NATOMS, nstep_qeq, spos, qsfp, qsfv
    real(8), intent(in) :: atype(NBUFFER), pos(NBUFFER,3)
    real(8), intent(inout) :: q(NBUFFER)
    integer(c_int) :: rc
    rc = rxmd_hip_put_lex(rxmd_hip_handle, int(NATOMS, c_int), qsfp, qsfv)
    if (rc /= 0) call die('PQEq (put_lex)', rc)
    rc = rxmd_hip_pqeq_arrays(rxmd_hip_handle, int(NBUFFER, c_int), int(NATOMS, c_int), atype, pos, q, spos)
    if (rc /= 0) call die('PQEq', rc)
    rc = rxmd_hip_get_lex(rxmd_hip_handle, int(NATOMS, c_int), qsfp, qsfv)
    if (rc /= 0) call die('PQEq (get_lex)', rc)
    nstep_qeq = rxmd_hip_last_qeq_iters(rxmd_hip_handle)
  end subroutine

  !> same arguments as the reference's FORCE(atype,pos,f,q), src/pot.F90:2,9-11; fills PE(0:13) of module atoms
  subroutine FORCE_hip(atype, pos, f, q)
    use atoms, only: NBUFFER, NATOMS, PE, astr, isPQEq, spos
    real(8), intent(in) :: atype(NBUFFER), q(NBUFFER), pos(NBUFFER,3)
    real(8), intent(inout) :: f(NBUFFER,3)
    integer(c_int) :: rc
    real(c_double) :: ke, qsum, pe2(0:13), a6(6)
    if (isPQEq) then
       rc = rxmd_hip_force_pqeq_arrays(rxmd_hip_handle, int(NBUFFER, c_int), int(NATOMS, c_int), atype, pos, f, q, spos, PE)
    else
       rc = rxmd_hip_force_arrays(rxmd_hip_handle, int(NBUFFER, c_int), int(NATOMS, c_int), atype, pos, f, q, PE)
    endif
    if (rc /= 0) call die('FORCE', rc)
    rc = rxmd_hip_get_energy(rxmd_hip_handle, ke, qsum, pe2, a6)   ! the virial this FORCE call added (src/pot.F90:65-72); reading resets it
    astr(1:6) = astr(1:6) + a6(1:6)
  end subroutine

  !> the reference's error behaviour: message on unit 6, then stop (src/main.F90:402-407)
  subroutine die(where, rc)
    character(len=*), intent(in) :: where
    integer(c_int), intent(in) :: rc
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: n
    p = rxmd_hip_last_error(rxmd_hip_handle)
    if (c_associated(p)) then
       call c_f_pointer(p, msg, (/512/))
       n = 1
       do while (n < 512 .and. msg(n) /= c_null_char); n = n + 1; end do
       write(6,'(a,a,a,i4,a,512a1)') 'ERROR in ', where, ' (rxmd_hip code', rc, '): ', msg(1:n-1)
    else
       write(6,'(a,a,a,i4,a)') 'ERROR in ', where, ' (rxmd_hip code', rc, ')'
    endif
    stop 1
  end subroutine

end module rxmd_hip_mod

!  bsp_atom_host.f90 -- Fortran host of libbspatom (ISO_C_BINDING over the C ABI in
!  include/bspatom.h).  Drop-in for `Bsp_Atom_omp.x < bsp_0.inp` up to the end of SOLVE_SYSTEM:
!  reads the three NAMELIST groups from stdin exactly as READ_INPUTS does (reference
!  src/ReadInputs.f90:15-21,27-37,75-85,155-184), solves all l-channels on the MI355X, and
!  writes stdout, Enl.dat and wf_n0.dat in the reference's formats
!  (src/matrices.f90:239-240,256-265,388-391; src/Bsp_Atom.f90:118-146).  KIND_PI = 0 ends there;
!  KIND_PI >= 3 also reproduces the state limits and Eigenvec_All.dat (src/matrices.f90:290-378);
!  KIND_PI = 1, 2 (one-photon, length / velocity gauge) continue through TRANS_AMP and CROSS_SECTIONS
!  (src/PhotoIon.f90:50-107, 274-468) to CSs/CrossSection_Len.dat / CSs/CrossSection_Vel.dat, the dipole
!  matrix elements coming from bspatom_dipole_elements.  Two variables CROSS_SECTIONS reads are set by the
!  reference for KIND_PI >= 3 only (Enl(n0,l0) -> E_ini(n0) here; the loop bound n1_max -> n1_fin), and its
!  T_fi(nf,il) is addressed with the channel l_fin it was allocated for; see DESIGN.md.
!  The Gaussian / LG-beam branches that follow SOLVE_SYSTEM for KIND_PI >= 3 are not part of this host.
!  With RANK / LOCAL_RANK / WORLD_SIZE in the environment (one process per GPU) the l-loop of SOLVE_SYSTEM
!  (src/matrices.f90:242-248) is sharded over the ranks and rank 0 writes the outputs (KIND_PI = 0; see below).
      MODULE BSPATOM_C
      USE ISO_C_BINDING
      IMPLICIT NONE
      TYPE, BIND(C) :: bspatom_input
        INTEGER(C_INT32_T) :: kind_grid, k, ka, nfun, kind_bc1, kind_bc2
        REAL(C_DOUBLE) :: ra, rb, rmax
        INTEGER(C_INT32_T) :: n0_ini, l_ini, m_ini, l_fin, lmax, kind_pot
        REAL(C_DOUBLE) :: emax_fin, zatom
      END TYPE
      TYPE, BIND(C) :: bspatom_sizes
        INTEGER(C_INT32_T) :: nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax, nintv_exp, nintv_lin, npad
      END TYPE
      INTERFACE
        SUBROUTINE bspatom_input_defaults(inp) BIND(C, NAME='bspatom_input_defaults')
          IMPORT :: bspatom_input
          TYPE(bspatom_input), INTENT(OUT) :: inp
        END SUBROUTINE
        INTEGER(C_INT) FUNCTION bspatom_problem_create(inp, device, prob) BIND(C, NAME='bspatom_problem_create')
          IMPORT :: bspatom_input, C_INT, C_PTR
          TYPE(bspatom_input), INTENT(IN) :: inp
          INTEGER(C_INT), VALUE :: device
          TYPE(C_PTR), INTENT(OUT) :: prob
        END FUNCTION
        SUBROUTINE bspatom_problem_destroy(prob) BIND(C, NAME='bspatom_problem_destroy')
          IMPORT :: C_PTR
          TYPE(C_PTR), VALUE :: prob
        END SUBROUTINE
        INTEGER(C_INT) FUNCTION bspatom_problem_sizes(prob, s) BIND(C, NAME='bspatom_problem_sizes')
          IMPORT :: bspatom_sizes, C_INT, C_PTR
          TYPE(C_PTR), VALUE :: prob
          TYPE(bspatom_sizes), INTENT(OUT) :: s
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_solve(prob, l0, nl, E, info) BIND(C, NAME='bspatom_solve')
          IMPORT :: C_INT, C_PTR, C_DOUBLE, C_INT32_T
          TYPE(C_PTR), VALUE :: prob
          INTEGER(C_INT), VALUE :: l0, nl
          REAL(C_DOUBLE), INTENT(OUT) :: E(*)
          INTEGER(C_INT32_T), INTENT(OUT) :: info(*)
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_eigvec(prob, l, n0, c) BIND(C, NAME='bspatom_eigvec')
          IMPORT :: C_INT, C_PTR, C_DOUBLE
          TYPE(C_PTR), VALUE :: prob
          INTEGER(C_INT), VALUE :: l, n0
          REAL(C_DOUBLE), INTENT(OUT) :: c(*)
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_eigvecs(prob, l, n0, cnt, Z) BIND(C, NAME='bspatom_eigvecs')
          IMPORT :: C_INT, C_PTR, C_DOUBLE
          TYPE(C_PTR), VALUE :: prob
          INTEGER(C_INT), VALUE :: l, n0, cnt
          REAL(C_DOUBLE), INTENT(OUT) :: Z(*)
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_dipole_elements(prob, l_ini, n0_ini, l_fin, n0_fin, cnt, a, D)                    &
     &                  BIND(C, NAME='bspatom_dipole_elements')
          IMPORT :: C_INT, C_PTR, C_DOUBLE
          TYPE(C_PTR), VALUE :: prob
          INTEGER(C_INT), VALUE :: l_ini, n0_ini, l_fin, n0_fin, cnt
          REAL(C_DOUBLE), INTENT(IN) :: a(3)
          REAL(C_DOUBLE), INTENT(OUT) :: D(*)
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_write_wf(prob, c, npts, r, u) BIND(C, NAME='bspatom_write_wf')
          IMPORT :: C_INT, C_PTR, C_DOUBLE
          TYPE(C_PTR), VALUE :: prob
          REAL(C_DOUBLE), INTENT(IN) :: c(*)
          INTEGER(C_INT), VALUE :: npts
          REAL(C_DOUBLE), INTENT(OUT) :: r(*), u(*)
        END FUNCTION
!       the one exchange of the sharded run (include/bspatom.h, csrc/comm.hip): an RCCL all-gather of the ranks' records
        INTEGER(C_INT) FUNCTION bspatom_run_token(buf, cap) BIND(C, NAME='bspatom_run_token')
          IMPORT :: C_INT, C_CHAR
          CHARACTER(KIND=C_CHAR) :: buf(*)
          INTEGER(C_INT), VALUE :: cap
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_comm_create(rank, world, dir, comm) BIND(C, NAME='bspatom_comm_create')
          IMPORT :: C_INT, C_CHAR, C_PTR
          INTEGER(C_INT), VALUE :: rank, world
          CHARACTER(KIND=C_CHAR), INTENT(IN) :: dir(*)
          TYPE(C_PTR), INTENT(OUT) :: comm
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_comm_allgather(comm, send, recv, cnt) BIND(C, NAME='bspatom_comm_allgather')
          IMPORT :: C_INT, C_PTR, C_DOUBLE, C_LONG
          TYPE(C_PTR), VALUE :: comm
          REAL(C_DOUBLE), INTENT(IN) :: send(*)
          REAL(C_DOUBLE), INTENT(OUT) :: recv(*)
          INTEGER(C_LONG), VALUE :: cnt
        END FUNCTION
        INTEGER(C_INT) FUNCTION bspatom_comm_collectives(comm) BIND(C, NAME='bspatom_comm_collectives')
          IMPORT :: C_INT, C_PTR
          TYPE(C_PTR), VALUE :: comm
        END FUNCTION
        SUBROUTINE bspatom_comm_destroy(comm) BIND(C, NAME='bspatom_comm_destroy')
          IMPORT :: C_PTR
          TYPE(C_PTR), VALUE :: comm
        END SUBROUTINE
!       libc, for the file exchange (the fallback when ranks share a GPU): no shell is forked from a process that holds a GPU
        INTEGER(C_INT) FUNCTION c_rename(old, new) BIND(C, NAME='rename')
          IMPORT :: C_INT, C_CHAR
          CHARACTER(KIND=C_CHAR), INTENT(IN) :: old(*), new(*)
        END FUNCTION
        INTEGER(C_INT) FUNCTION c_mkdir(path, mode) BIND(C, NAME='mkdir')
          IMPORT :: C_INT, C_CHAR
          CHARACTER(KIND=C_CHAR), INTENT(IN) :: path(*)
          INTEGER(C_INT), VALUE :: mode
        END FUNCTION
        INTEGER(C_INT) FUNCTION c_usleep(us) BIND(C, NAME='usleep')
          IMPORT :: C_INT
          INTEGER(C_INT), VALUE :: us
        END FUNCTION
      END INTERFACE
      END MODULE BSPATOM_C

      PROGRAM BSP_ATOM_MI355X
      USE ISO_C_BINDING
      USE BSPATOM_C
      IMPLICIT NONE
      INTEGER, PARAMETER :: DP = KIND(1.0D0)
!     VARS_BSP / VARS_TISE / VARS_FIELD, same names and defaults as the reference
      INTEGER :: KIND_GRID, k, ka, nfun, KIND_BC1, KIND_BC2, nfib
      REAL(DP) :: ra, rb, rmax
      INTEGER :: n0_ini, l_ini, m_ini, l_fin, lmax, KIND_POT, KIND_EGR, KIND_NLM
      REAL(DP) :: Emax_fin, Zatom
      INTEGER :: KIND_PI, KIND_SCP, KIND_TD, KIND_ENV, KIND_RK, KIND_VEC, ncyc, ncyc2, moam, mph
      INTEGER :: nEpts, nthpts, nphpts
      REAL(DP) :: A0, w0, Eph, Eph2, I0, I01, b0, afocus, Eref, bx, B0z, A01, t_delay, A0x, A0y, A0z
      NAMELIST / VARS_BSP / KIND_GRID, ra, rb, rmax, k, ka, nfun, KIND_BC1, KIND_BC2, nfib
      NAMELIST / VARS_TISE / n0_ini, l_ini, m_ini, l_fin, lmax, Emax_fin, Zatom, &
     &                        KIND_POT, KIND_EGR, KIND_NLM
      NAMELIST / VARS_FIELD / KIND_PI, KIND_SCP, KIND_TD, KIND_ENV, KIND_RK,  &
     &                         KIND_VEC, A0, w0, Eph, ncyc, Eph2, ncyc2, moam, &
     &                         mph, I0, I01, b0, afocus, nEpts, nthpts, nphpts,&
     &                         Eref, bx, B0z, A01, t_delay, A0x, A0y, A0z
      TYPE(bspatom_input) :: inp
      TYPE(bspatom_sizes) :: sz
      TYPE(C_PTR) :: prob
      INTEGER(C_INT) :: rc
      INTEGER :: l, i, npts
!     KIND_PI >= 3 bookkeeping (names as in SOLVE_SYSTEM)
      INTEGER :: n0_fin, n1_fin, nlim, nbds, nbold, ntemp, nE0, n1_max, ni, ntemp0
      INTEGER, ALLOCATABLE :: n01(:,:)
      REAL(DP) :: Elim, Ei
      REAL(DP), ALLOCATABLE :: Zl(:,:)
      REAL(DP), ALLOCATABLE :: En(:), ci(:), r(:), u(:)
      INTEGER(C_INT32_T), ALLOCATABLE :: info(:)
!     KIND_PI = 1, 2: TRANS_AMP / CROSS_SECTIONS (names as in PhotoIon.f90)
      INTEGER :: lf, mf, l0, m0, nf
      REAL(DP) :: T3ja, T3jb, c0, c1, c2, An, d1, d2, csl, M_au, cc0, cc1, Ef, avec(3)
      REAL(DP), ALLOCATABLE :: dip(:), T_fi(:)
      REAL(DP), PARAMETER :: PI = 3.141592653589793238462643D0, c_au = 137.03599913815D0, a_au = 5.29177249D-9
      REAL(DP), PARAMETER :: I0_au = 3.50944758D16
      REAL(DP), EXTERNAL :: W3J
!     stdout echo of READ_INPUTS / SEL_LM (the numbers a run prints before MATRIX_SVT starts)
      INTEGER :: nshell(3), ntot_el, ifib, fa, fb, fc, nsel, il, im, la
      REAL(DP) :: rog(3,0:3), xn, ssum, Epump, Eprobe, kph
!     l-channels sharded over the GPUs of a node: one process per GPU, started by any launcher that sets RANK, LOCAL_RANK and
!     WORLD_SIZE (python -m torch.distributed.run --no-python, mpirun with a wrapper, a shell loop).  The channels are independent
!     (matrices.f90:242-248), so the ranks exchange nothing while they solve; the spectra reach rank 0, which writes the
!     reference's outputs, through files of a scratch directory (BSPATOM_XCHG, unique per run) -- the file boundary this
!     program has anyway.  KIND_PI = 0 only, like the Python host's run_sharded.
      INTEGER :: rank, world, lrank, devid, l0s, nls, nbase, nrem, rr, owner, r0s, nlr, ios, tries
      LOGICAL :: have_wf, ex, launched, use_rccl
      CHARACTER(LEN=512) :: envv, xdir, fname
      INTEGER(C_INT32_T), ALLOCATABLE :: infoall(:)
      INTEGER(C_INT32_T) :: hdr(5)
      INTEGER(C_INT32_T), PARAMETER :: XMAGIC = 1112756312                     ! 'BSPX'
      CHARACTER(KIND=C_CHAR) :: tokc(128)
      CHARACTER(LEN=128) :: token
      TYPE(C_PTR) :: comm
      INTEGER :: nmax, o
      INTEGER(C_LONG) :: reclen
      REAL(C_DOUBLE), ALLOCATABLE :: sbuf(:), rbuf(:)

      rank = 0; world = 1; lrank = 0
      CALL GET_ENVIRONMENT_VARIABLE('WORLD_SIZE', envv, STATUS=ios)
      IF( ios == 0 ) READ(envv,*,IOSTAT=ios) world
      IF( world < 1 ) world = 1
      CALL GET_ENVIRONMENT_VARIABLE('RANK', envv, STATUS=ios)
      launched = ( ios == 0 .AND. LEN_TRIM(envv) > 0 )                         ! under a launcher, also at world size 1
      IF( world > 1 ) THEN
        IF( ios == 0 ) READ(envv,*,IOSTAT=ios) rank
        CALL GET_ENVIRONMENT_VARIABLE('LOCAL_RANK', envv, STATUS=ios)
        IF( ios == 0 ) READ(envv,*,IOSTAT=ios) lrank
        IF( rank < 0 .OR. rank >= world ) THEN
          WRITE(0,*) 'bsp_atom_host: RANK outside 0 .. WORLD_SIZE - 1'
          STOP 2
        END IF
!       every rank but the first is silent: rank 0's stdout is the program's
        IF( rank > 0 ) OPEN( UNIT=6, FILE='/dev/null', ACTION='WRITE' )
      END IF
!     launchers forward stdin to one rank at most: the namelists can come from the file BSPATOM_INPUT names
      CALL GET_ENVIRONMENT_VARIABLE('BSPATOM_INPUT', envv, STATUS=ios)
      IF( ios == 0 .AND. LEN_TRIM(envv) > 0 ) OPEN( UNIT=5, FILE=TRIM(envv), ACTION='READ', STATUS='OLD' )
      devid = lrank
      CALL GET_ENVIRONMENT_VARIABLE('BSPATOM_DEVICE', envv, STATUS=ios)
      IF( ios == 0 .AND. LEN_TRIM(envv) > 0 ) READ(envv,*,IOSTAT=ios) devid

      WRITE(6,'(A64)') 'PROGRAM TO CALCULATE ELECTRONIC STRUCTURE AND PI CROSS SECTIONS,'
      WRITE(6,'(A17,/)') '  USING B-SPLINES'

      KIND_GRID = 0; ra = 0.D0; rb = 0.D0; rmax = 0.D0; k = 0; ka = 0; nfun = 0
      KIND_BC1 = 0; KIND_BC2 = 0; nfib = 1
      READ(5,VARS_BSP)
      KIND_POT = 0; n0_ini = 1; l_ini = 0; m_ini = 0; l_fin = 0; lmax = 0
      Emax_fin = -1.D0; Zatom = 1.D0; KIND_EGR = 0; KIND_NLM = 0
      READ(5,VARS_TISE)
      KIND_PI = 0; KIND_SCP = 0; KIND_TD = 0; KIND_ENV = 0; KIND_RK = 6; KIND_VEC = 0
      A0 = 0.D0; I0 = 0.D0; A01 = 0.D0; I01 = 0.D0; w0 = 0.D0; Eph = 0.D0; mph = 0; moam = 0
      b0 = 0.D0; afocus = 0.D0; nEpts = 10; Eref = 0.D0; nthpts = 1; nphpts = 1; ncyc = 0
      bx = 0.D0; B0z = 0.D0; t_delay = 0.D0; ncyc2 = 0; Eph2 = 0.D0; A0x = 0.D0; A0y = 0.D0; A0z = 1.D0
      READ(5,VARS_FIELD)
      lmax = MAX(lmax, l_fin)                          ! ReadInputs.f90:87
      IF( KIND_PI == 1 .OR. KIND_PI == 2 ) THEN
!       SEL_LM, dipolar case (grid.f90:128-143): final channels l0-1 (if it can carry m0) and l0+1; TRANS_AMP takes the last
        lf = l_ini + 1; mf = m_ini
        IF( lf > lmax ) THEN
          WRITE(6,*) 'bsp_atom_host: l_fin = l_ini + 1 is beyond lmax of the input (the reference reads an unallocated ci_fin)'
          STOP 2
        END IF
      END IF
!     Bsp_Atom.f90:59-60: the output directory of the cross-section files exists from the start
      CALL EXECUTE_COMMAND_LINE('mkdir -p CSs')

      CALL bspatom_input_defaults(inp)
      inp%kind_grid = KIND_GRID; inp%k = k; inp%ka = ka; inp%nfun = nfun
      inp%kind_bc1 = KIND_BC1; inp%kind_bc2 = KIND_BC2
      inp%ra = ra; inp%rb = rb; inp%rmax = rmax
      inp%n0_ini = n0_ini; inp%l_ini = l_ini; inp%m_ini = m_ini; inp%l_fin = l_fin; inp%lmax = lmax
      inp%kind_pot = KIND_POT; inp%emax_fin = Emax_fin; inp%zatom = Zatom
      IF( world > 1 .AND. KIND_PI /= 0 ) THEN
        WRITE(0,*) 'bsp_atom_host: WORLD_SIZE > 1 shards the l-channels of a KIND_PI = 0 run; the other branches run on one GPU'
        STOP 2
      END IF
      rc = bspatom_problem_create(inp, INT(devid,C_INT), prob)
      IF( rc /= 0 ) THEN
        WRITE(6,*) 'bsp_atom_host: bspatom_problem_create failed, code ', rc
        STOP 1
      END IF
!     Sharded run (or any run under a launcher): the spectra travel ONCE, at the end, as an RCCL all-gather of one record per rank
!     (csrc/comm.hip).  The communicator is created now, before anything is solved.  Ranks that share a GPU (more ranks than
!     devices: a test box) and BSPATOM_XCHG_MODE=files use stream files in the exchange directory instead; their names carry the
!     token of this launch (launcher pid + start time), so a file another run left behind is never taken for this run's.
      use_rccl = .FALSE.; comm = C_NULL_PTR; token = ' '
      IF( world > 1 .OR. ( launched .AND. KIND_PI == 0 ) ) THEN
        xdir = '.bspatom_xchg'
        CALL GET_ENVIRONMENT_VARIABLE('BSPATOM_XCHG', envv, STATUS=ios)
        IF( ios == 0 .AND. LEN_TRIM(envv) > 0 ) xdir = envv
        rc = c_mkdir(TRIM(xdir)//C_NULL_CHAR, INT(O'777',C_INT))               ! exists already: fine
        rc = bspatom_run_token(tokc, 128_C_INT)
        IF( rc /= 0 ) THEN
          WRITE(0,*) 'bsp_atom_host: no run token, code ', rc
          STOP 1
        END IF
        DO i = 1, 128
          IF( tokc(i) == C_NULL_CHAR ) EXIT
          token(i:i) = tokc(i)
        END DO
        CALL GET_ENVIRONMENT_VARIABLE('BSPATOM_XCHG_MODE', envv, STATUS=ios)
        IF( .NOT. ( ios == 0 .AND. TRIM(envv) == 'files' ) ) THEN
          rc = bspatom_comm_create(INT(rank,C_INT), INT(world,C_INT), TRIM(xdir)//C_NULL_CHAR, comm)
          IF( rc == 0 ) THEN
            use_rccl = .TRUE.
          ELSE IF( rc /= -5 ) THEN                                             ! -5: ranks share a GPU, or no librccl: files
            WRITE(0,*) 'bsp_atom_host: bspatom_comm_create failed, code ', rc
            STOP 1
          END IF
        END IF
      END IF
      rc = bspatom_problem_sizes(prob, sz)
!     ---- what READ_INPUTS prints (ReadInputs.f90:54,67,71,93,127,186-202,222,236-271), in its order ----
      IF( KIND_GRID == 2 ) THEN
        WRITE(6,'(/,A29,I5)') 'Initial Number of Functions: ', nfun
        WRITE(6,'(A29,I5)') 'Number of functions changed: ', sz%nfun
      END IF
      nfun = sz%nfun
      lmax = sz%lmax
      WRITE(6,'(A40,I5)') 'Number of B-spline Functions / l: nfun =', nfun
      WRITE(6,'(/,A38,I3)') 'Max. Angular Momenta Included: l_max =', lmax
      IF( KIND_POT == 1 ) THEN
!       Rogers screening parameters, echoed list-directed as the reference does (:95-128)
        nshell = (/ 2, 8, 8 /)
        rog = 0.D0
        rog(1,0:2) = (/ 0.8855D0, 0.2549D0, -0.0901D0 /)
        rog(2,0:2) = (/ 0.3386D0, 1.1323D0, -0.4904D0 /)
        rog(3,0:3) = (/ 0.1437D0, 0.9129D0, -0.6940D0, 0.2503D0 /)
        ntot_el = 0
        DO i = 1, 3
          ntot_el = ntot_el + nshell(i)
          xn = DBLE(Zatom - ntot_el)
          IF( xn == 0.D0 ) xn = 1.D0
          ssum = 0.D0
          DO l = 0, 3
            ssum = ssum + rog(i,l) / (xn**l)
          END DO
          WRITE(6,*) i, xn, (xn + 1.D0) * ssum
        END DO
      END IF
      WRITE(6,'(/,A17)') 'Field Parameters:'
      IF( A0 /= 0.D0 ) WRITE(6,'(A4,G12.5)') 'A0 =', A0
      IF( I0 /= 0.D0 ) WRITE(6,'(A4,G12.5)') 'I0 =', I0
      IF( w0 /= 0.D0 ) WRITE(6,'(A4,G12.5)') 'w0 =', w0
      IF( b0 /= 0.D0 ) WRITE(6,'(A4,G12.5)') 'b0 =', b0
      IF( Eph /= 0.D0 ) WRITE(6,'(A5,G12.5)') 'Eph =', Eph
      IF( A01 /= 0.D0 ) WRITE(6,'(A5,G12.5)') 'A01 =', A01
      IF( I01 /= 0.D0 ) WRITE(6,'(A5,G12.5)') 'I01 =', I01
      IF( moam /= 0 ) WRITE(6,'(A20,I3)') 'Topological Charge =', moam
      IF( afocus /= 0.D0 ) WRITE(6,'(A22,G12.5)') 'Focusing angle (Deg.):', afocus
      IF( ncyc /= 0 ) WRITE(6,'(A18,I3)') 'Num. Opt. Cycles =', ncyc
      IF( KIND_SCP == 1 ) WRITE(6,'(A25)') 'Lorenz Scalar-Potential Included'
      IF( KIND_TD /= 0 ) WRITE(6,'(A19,I2)') 'Runge--Kutta Order:', KIND_RK
      IF( t_delay /= 0.D0 ) WRITE(6,'(A11,G12.5,A3)') 'Time-Delay:', t_delay, ' fs'
      IF( A0x /= 0.D0 ) WRITE(6,'(A18)') 'Laser Pulse Pol. x'
      IF( A0y /= 0.D0 ) WRITE(6,'(A18)') 'Laser Pulse Pol. y'
      IF( A0z /= 0.D0 ) WRITE(6,'(A18)') 'Laser Pulse Pol. z'
      IF( nfib >= 0 ) THEN
        ifib = 0                                          ! Fibonacci number nfib (Modules.f90:947-973)
        IF( nfib == 1 .OR. nfib == 2 ) ifib = 1
        IF( nfib >= 3 ) THEN
          fa = 1; fb = 1
          DO i = 3, nfib
            fc = fa + fb; fa = fb; fb = fc
          END DO
          ifib = fb
        END IF
        WRITE(6,'(A41,I5)') 'Number of Pts for Fibonacci sampling Pts:', ifib
      END IF
      Epump = SQRT(I0 / I0_au)
      Eprobe = 0.D0
!     (KIND_PI >= 8 with the Coulomb potential: READ_INPUTS re-derives the pump / probe photon energies and cycle counts of the
!     field layer here and prints four 'Modified ...' lines, ReadInputs.f90:236-253.  The field layer is outside this
!     solver's scope (SURVEY 8b: its keys are accepted, nothing is computed from them), so those lines are not reproduced.)
      IF( KIND_PI >= 8 .AND. KIND_POT == 0 ) THEN
        CONTINUE
      ELSE IF( KIND_POT /= 0 ) THEN
        Eprobe = SQRT(I01 / I0_au)
        kph = Eph2 / c_au
        WRITE(6,'(A31,I5)') 'Modified Num. Opt. Cycles Pump:', ncyc
        WRITE(6,'(A28,G14.7)') 'Modified Photon Energy Pump:', Eph
        WRITE(6,'(A32,I5)') 'Modified Num. Opt. Cycles Probe:', ncyc2
        WRITE(6,'(A29,G14.7)') 'Modified Photon Energy Probe:', Eph2
        WRITE(6,'(A28,G14.7)') 'Modified Photon Wave Number:', kph
      END IF
      WRITE(6,'(A7,G14.7)') 'Epump =', Epump
      WRITE(6,'(A8,G14.7)') 'Eprobe =', Eprobe
!     ---- GRID (grid.f90:25,33,46-47,65-66) ----
      IF( KIND_GRID == 0 ) THEN
        WRITE(6,'(A23)') 'Linear Knotpts Sequence'
      ELSE IF( KIND_GRID == 1 ) THEN
        WRITE(6,'(A29)') 'Exponential Knotpts Sequence'
      ELSE
        WRITE(6,'(/,A35)') 'Exponential-Linear Knotpts Sequence'
        WRITE(6,'(A30,G12.5)') 'Limit of Exp. Sequence: rmax =', rmax
      END IF
      WRITE(6,'(/,A22,I6)') 'Number of Knot Points:' , sz%nkp
      WRITE(6,'(A27,2I3)') 'Multiplicity of END points:', sz%nbc1, sz%nbc2
!     ---- SEL_LM (grid.f90:113-236): the (l, m) of the final states, by KIND_PI ----
      WRITE(6,'(/,A22)') 'Selecting Final States'
      WRITE(6,'(/,A22)') 'Selected final states:'
      WRITE(6,'(T3,A1,T7,A2,T12,A2)') 'i', 'lf', 'mf'
      WRITE(6,'(T2,A12)') '------------'
      nsel = 0
      IF( KIND_PI == 0 ) THEN
        WRITE(6,'(T1,I3,T6,I3,T11,I3)') 1, l_ini, m_ini
      ELSE IF( KIND_PI <= 2 ) THEN                      ! dipolar: l0 - 1 (if it exists and can carry m0) and l0 + 1
        DO il = l_ini - 1, l_ini + 1, 2
          IF( il >= 0 .AND. il >= m_ini ) THEN
            nsel = nsel + 1
            WRITE(6,'(T1,I3,T6,I3,T11,I3)') nsel, il, m_ini
          END IF
        END DO
      ELSE IF( KIND_PI == 5 .OR. KIND_PI == 6 .OR. ((KIND_PI == 8 .OR. KIND_PI == 9) .AND. KIND_NLM == 0) ) THEN
        DO il = 0, lmax                                   ! every l that can carry m0
          IF( il >= ABS(m_ini) ) THEN
            nsel = nsel + 1
            WRITE(6,'(T1,I3,T6,I3,T11,I3)') nsel, il, m_ini
          END IF
        END DO
      ELSE IF( (KIND_PI == 8 .OR. KIND_PI == 9) .AND. KIND_NLM == 1 ) THEN
        DO il = 0, lmax                                   ! unpolarised initial state: |m| <= min(l, l0)
          la = MIN(il, l_ini)
          DO im = -la, la
            nsel = nsel + 1
            WRITE(6,'(T1,I3,T6,I3,T11,I3)') nsel, il, im
          END DO
        END DO
      ELSE IF( .NOT. (KIND_PI == 8 .OR. KIND_PI == 9) ) THEN
        DO il = 0, lmax                                   ! all (l, m)
          DO im = -il, il
            nsel = nsel + 1
            WRITE(6,'(T1,I3,T6,I3,T11,I3)') nsel, il, im
          END DO
        END DO
      END IF
      WRITE(6,'(/,A34)') 'Calculating S, V, U and T Matrices'

      ALLOCATE( En(nfun*(lmax+1)), info(lmax+1), ci(nfun) )
      have_wf = .FALSE.
      IF( world == 1 ) THEN
        nbase = lmax + 1; nrem = 0; l0s = 0; nls = lmax + 1; owner = 0
        rc = bspatom_solve(prob, 0_C_INT, INT(lmax+1,C_INT), En, info)
      ELSE
!       this rank's block of channels, the static partition of bspatom_amd/parallel.py::channel_range
        nbase = (lmax+1) / world; nrem = MOD(lmax+1, world)
        nls = nbase; IF( rank < nrem ) nls = nbase + 1
        l0s = rank*nbase + MIN(rank, nrem)
        owner = 0
        DO rr = 0, world-1
          r0s = rr*nbase + MIN(rr, nrem); nlr = nbase; IF( rr < nrem ) nlr = nbase + 1
          IF( l_ini >= r0s .AND. l_ini < r0s + nlr ) owner = rr
        END DO
        info = 0; rc = 0
        IF( nls > 0 ) rc = bspatom_solve(prob, INT(l0s,C_INT), INT(nls,C_INT), En(l0s*nfun+1:), info(l0s+1:))
      END IF
      IF( rc /= 0 ) THEN
        IF( rc == -3 ) WRITE(6,*) 'FATAL ERROR - BSPLVB'
        WRITE(0,*) 'bsp_atom_host: bspatom_solve failed, code ', rc
        STOP 1
      END IF
      IF( world > 1 .OR. use_rccl ) THEN
        npts = 10000
        IF( rank == owner ) THEN                          ! the consumed eigenvector and its WRITE_WF table, by the rank that solved l_ini
          ALLOCATE( r(0:npts), u(0:npts) )
          rc = bspatom_eigvec(prob, INT(l_ini,C_INT), INT(n0_ini,C_INT), ci)
          IF( rc == 0 ) rc = bspatom_write_wf(prob, ci, INT(npts,C_INT), r, u)
          IF( rc /= 0 ) THEN
            IF( rc == -3 ) WRITE(6,*) 'FATAL ERROR - BSPLVB'
            WRITE(0,*) 'bsp_atom_host: eigenvector / WRITE_WF failed, code ', rc
            STOP 1
          END IF
          have_wf = .TRUE.
        END IF
      END IF
      IF( use_rccl ) THEN
!       one record per rank: l0, nl, wf flag | info(nmax) | En(nfun, nmax) | r(0:npts), u(0:npts); all-gathered, rank 0 unpacks
        nmax = nbase; IF( nrem > 0 ) nmax = nbase + 1
        IF( world == 1 ) THEN
          nmax = lmax + 1; l0s = 0; nls = lmax + 1
        END IF
        reclen = 3 + nmax + INT(nmax,C_LONG)*nfun + 2*(npts+1)
        ALLOCATE( sbuf(reclen), rbuf(reclen*world) )
        sbuf = 0.D0
        sbuf(1) = l0s; sbuf(2) = nls; IF( have_wf ) sbuf(3) = 1.D0
        IF( nls > 0 ) THEN
          sbuf(4:3+nls) = info(l0s+1:l0s+nls)
          sbuf(4+nmax:3+nmax+nls*nfun) = En(l0s*nfun+1:(l0s+nls)*nfun)
        END IF
        o = 3 + nmax + nmax*nfun
        IF( have_wf ) THEN
          sbuf(o+1:o+npts+1) = r(0:npts); sbuf(o+npts+2:o+2*npts+2) = u(0:npts)
        END IF
        rc = bspatom_comm_allgather(comm, sbuf, rbuf, reclen)
        IF( rc /= 0 ) THEN
          WRITE(0,*) 'bsp_atom_host: the all-gather of the spectra failed, code ', rc
          STOP 1
        END IF
        IF( rank > 0 ) THEN
          CALL bspatom_comm_destroy(comm)
          CALL bspatom_problem_destroy(prob)
          STOP
        END IF
        DO rr = 1, world-1
          o = rr*INT(reclen)
          r0s = NINT(rbuf(o+1)); nlr = NINT(rbuf(o+2))
          IF( r0s /= rr*nbase + MIN(rr, nrem) .OR. nlr /= nbase + MERGE(1, 0, rr < nrem) ) THEN
            WRITE(0,*) 'bsp_atom_host: rank ', rr, ' sent channels ', r0s, nlr, ' (not its block of this run)'
            STOP 3
          END IF
          IF( nlr > 0 ) THEN
            info(r0s+1:r0s+nlr) = NINT(rbuf(o+4:o+3+nlr))
            En(r0s*nfun+1:(r0s+nlr)*nfun) = rbuf(o+4+nmax:o+3+nmax+nlr*nfun)
          END IF
          IF( rbuf(o+3) == 1.D0 ) THEN
            ALLOCATE( r(0:npts), u(0:npts) )
            r(0:npts) = rbuf(o+3+nmax+nmax*nfun+1:o+3+nmax+nmax*nfun+npts+1)
            u(0:npts) = rbuf(o+3+nmax+nmax*nfun+npts+2:o+3+nmax+nmax*nfun+2*npts+2)
            have_wf = .TRUE.
          END IF
        END DO
        WRITE(0,'(A,I0,A,I0,A)') 'bsp_atom_host: spectra of ', world, ' rank(s) gathered by RCCL all-gather (',                 &
     &                           bspatom_comm_collectives(comm), ' collective)'
        CALL bspatom_comm_destroy(comm)
        DEALLOCATE( sbuf, rbuf )
      ELSE IF( world > 1 ) THEN
        IF( rank > 0 ) THEN
!         written under another name and renamed: rank 0 never sees half a file
          WRITE(fname,'(A,A,A,A,I0)') TRIM(xdir), '/spec.', TRIM(token), '.', rank
          OPEN( UNIT=77, FILE=TRIM(fname)//'.tmp', ACCESS='STREAM', FORM='UNFORMATTED', ACTION='WRITE', STATUS='REPLACE' )
          WRITE(77) XMAGIC, INT(nfun,C_INT32_T), INT(lmax+1,C_INT32_T), INT(l0s,C_INT32_T), INT(nls,C_INT32_T)
          IF( nls > 0 ) WRITE(77) info(l0s+1:l0s+nls), En(l0s*nfun+1:(l0s+nls)*nfun)
          IF( have_wf ) THEN
            WRITE(77) 1_C_INT32_T, r, u
          ELSE
            WRITE(77) 0_C_INT32_T
          END IF
          CLOSE(77)
          rc = c_rename(TRIM(fname)//'.tmp'//C_NULL_CHAR, TRIM(fname)//C_NULL_CHAR)
          IF( rc /= 0 ) THEN
            WRITE(0,*) 'bsp_atom_host: cannot rename ', TRIM(fname)
            STOP 3
          END IF
          CALL bspatom_problem_destroy(prob)
          STOP
        END IF
        ALLOCATE( infoall(1) )
        DO rr = 1, world-1                                ! rank 0 collects
          WRITE(fname,'(A,A,A,A,I0)') TRIM(xdir), '/spec.', TRIM(token), '.', rr
          tries = 0
          DO
            INQUIRE( FILE=TRIM(fname), EXIST=ex )
            IF( ex ) EXIT
            tries = tries + 1
            IF( tries > 36000 ) THEN                      ! half an hour
              WRITE(0,*) 'bsp_atom_host: no spectra from rank ', rr, ' in ', TRIM(xdir)
              STOP 3
            END IF
            rc = c_usleep(50000_C_INT)
          END DO
          OPEN( UNIT=77, FILE=TRIM(fname), ACCESS='STREAM', FORM='UNFORMATTED', ACTION='READ', STATUS='OLD' )
          READ(77) hdr
          r0s = hdr(4); nlr = hdr(5)
!         the file names this run (token) and must describe this problem and exactly rank rr's block of it
          IF( hdr(1) /= XMAGIC .OR. hdr(2) /= nfun .OR. hdr(3) /= lmax+1 .OR. r0s /= rr*nbase + MIN(rr, nrem) .OR.                  &
     &        nlr /= nbase + MERGE(1, 0, rr < nrem) ) THEN
            WRITE(0,*) 'bsp_atom_host: ', TRIM(fname), ' does not belong to this run (header ', hdr, ')'
            STOP 3
          END IF
          IF( nlr > 0 ) READ(77) info(r0s+1:r0s+nlr), En(r0s*nfun+1:(r0s+nlr)*nfun)
          READ(77) infoall(1)
          IF( infoall(1) == 1 ) THEN
            ALLOCATE( r(0:npts), u(0:npts) )
            READ(77) r, u
            have_wf = .TRUE.
          END IF
          CLOSE(77, STATUS='DELETE')
        END DO
        DEALLOCATE( infoall )
      END IF
      WRITE(6,'(A19,/)') 'Matrices Calculated'

      n0_fin = -1; n1_fin = -1; nlim = 0; nbds = 0; ntemp = 0; ntemp0 = 0
      IF( KIND_PI >= 3 ) ALLOCATE( n01(0:lmax,3) )
      OPEN( UNIT=75, FILE='Enl.dat', ACTION='WRITE' )
      WRITE(75,*) nfun
      DO l = 0, lmax
        IF( info(l+1) /= 0 ) THEN
          WRITE(6,*) 'ERROR DIAGONALIZING THE MATRIX!', info(l+1)
          WRITE(6,100) 'l = ', l
          STOP
        END IF
        WRITE(6,100) 'l0 = ', l
        WRITE(6,110) 'HC = ESC eigenvalue solved'
        WRITE(6,120) 'n', 'Eigenvalues'
        WRITE(6,120) '-', '-----------'
        DO i = 1, nfun
          IF( i <= 20 ) WRITE(6,200) i+l, En(l*nfun+i)
          WRITE(75,200) i, En(l*nfun+i)
        END DO
        IF( l == l_ini ) THEN
          WRITE(6,'(/,A29,/)') 'Writing down Initial State WF'
          npts = 10000
          rc = 0
          IF( .NOT. have_wf ) THEN
            rc = bspatom_eigvec(prob, INT(l,C_INT), INT(n0_ini,C_INT), ci)
            ALLOCATE( r(0:npts), u(0:npts) )
            IF( rc == 0 ) rc = bspatom_write_wf(prob, ci, INT(npts,C_INT), r, u)
          END IF
          IF( rc == -3 ) THEN
            WRITE(6,*) 'FATAL ERROR - BSPLVB'
            STOP
          ELSE IF( rc /= 0 ) THEN
            WRITE(6,*) 'bsp_atom_host: eigenvector / WRITE_WF failed, code ', rc
            STOP 1
          END IF
          OPEN( UNIT=30, FILE='wf_n0.dat', ACTION='WRITE' )
          DO i = 0, npts
            WRITE(30,'(2G20.10)') r(i), u(i)
          END DO
          CLOSE(30)
          DEALLOCATE( r, u )
        END IF
        IF( KIND_PI >= 3 ) THEN
!         state limits of this channel (matrices.f90:290-328); n0_fin, n1_fin, ntemp, Emax_fin deliberately
!         persist from channel to channel, as they do in the reference
          IF( Emax_fin == -1.0D0 ) THEN
            Emax_fin = En(l*nfun+nfun)
            Elim = Emax_fin
          ELSE IF( KIND_PI >= 8 ) THEN
            Elim = Emax_fin
          ELSE
            Elim = Emax_fin + 0.25D0
          END IF
          nbold = 0
          DO i = 1, nfun
            Ei = En(l*nfun+i)
            IF( Ei < 0.D0 ) THEN
              n0_fin = i
              nbold = nbold + 1
            END IF
            IF( Ei <= Emax_fin ) n1_fin = i
            IF( Ei <= Elim ) ntemp = i
            IF( Ei > Emax_fin .AND. Ei > Elim ) EXIT
          END DO
          nbds = MAX(nbds,nbold)
          nE0 = n0_fin + 1
          n0_fin = nE0
          n1_fin = n1_fin + 1
          IF( KIND_PI >= 5 ) n0_fin = 1
          nlim = MAX(nlim,ntemp)
          n01(l,1) = n0_fin; n01(l,2) = n1_fin; n01(l,3) = nE0 - 1
          WRITE(6,'(/,A23,I3)') 'NUMBER OF BOUND STATES:', nbold
          WRITE(6,'(A14,I3,A6,2I5)') 'LIMITS FOR l =', l, ' STATE:', n0_fin+l, n1_fin+l
          ntemp = MIN(MAX(n1_fin+40,nlim),nfun)
          IF( l == 0 ) ntemp0 = ntemp
        END IF
      END DO
      CLOSE(75)

      IF( KIND_PI >= 3 ) THEN
!       Eigenvec_All.dat (matrices.f90:355-378): the first n1_max eigenvectors of every channel
        n1_max = MIN(MAX(MAXVAL(n01(:,2))+20,nlim),nfun)
        WRITE(6,'(A8,I5)') 'n1_max =', n1_max
        IF( n1_max > ntemp0 ) THEN
          WRITE(6,*) 'bsp_atom_host: n1_max exceeds the vectors the reference keeps (ctemp)'
          STOP 1
        END IF
        ALLOCATE( Zl(nfun,n1_max) )
        OPEN( UNIT=80, FILE='Eigenvec_All.dat', ACTION='WRITE' )
        WRITE(80,*) nfun, n1_max, lmax
        DO l = 0, lmax
          rc = bspatom_eigvecs(prob, INT(l,C_INT), 1_C_INT, INT(n1_max,C_INT), Zl)
          IF( rc /= 0 ) THEN
            WRITE(6,*) 'bsp_atom_host: bspatom_eigvecs failed, code ', rc
            STOP 1
          END IF
          WRITE(80,*) l
          DO ni = 1, n1_max
            WRITE(80,300) ni, (Zl(i,ni), i=1,nfun)
          END DO
        END DO
        CLOSE(80)
      END IF
      IF( KIND_PI == 1 .OR. KIND_PI == 2 ) THEN
!       final-state window at l = l_fin (matrices.f90:272-283)
        l0 = l_ini; m0 = m_ini
        IF( Emax_fin == -1.0D0 ) Emax_fin = En(lf*nfun+nfun)
        n0_fin = -1; n1_fin = -1
        DO i = 1, nfun
          IF( En(lf*nfun+i) < 0.D0 ) n0_fin = i
          IF( En(lf*nfun+i) <= Emax_fin ) n1_fin = i
        END DO
        n0_fin = MIN(n0_fin+1, nfun-1)
        WRITE(6,'(/,A26,I2,A3,X,2I4)') 'LIMITS FOR FINAL STATE (l=', lf, ') :', n0_fin, n1_fin
        IF( n1_fin + 1 > nfun .OR. n1_fin < n0_fin ) THEN
          WRITE(6,*) 'bsp_atom_host: the density-of-states factor needs E_fin(n1_fin+1); lower Emax_fin'
          STOP 2
        END IF
!       TRANS_AMP (PhotoIon.f90:36-107): T_fi(n) = An c0 <c_fin(n)| c1 rij1 + c2 rij2 |c_ini>
        WRITE(6,'(/,A33)') 'Calculating Transition Amplitudes'
        WRITE(6,'(A14,3I3)') 'Initial State:', n0_ini+l0, l0, m0
        T3ja = W3J(lf,1,l0,-mf,mph,m0)
        IF( KIND_PI == 1 ) THEN
          T3jb = W3J(lf,1,l0,0,0,0)
          c1 = (-1.D0)**(lf+l0+mf) * SQRT( DBLE((2*lf+1)*(2*l0+1)) ) * T3ja * T3jb
          c0 = 1.0D0
          avec = (/ c1, 0.D0, 0.D0 /)                  ! A = c1 * int B r B
        ELSE
          c0 = SQRT(DBLE(l0 + 1)) * T3ja
          c1 = DBLE(l0 + 1); c2 = -1.D0                ! lf = l0 + 1
          avec = (/ 0.D0, c1, c2 /)                    ! A = c1 * int B B / r + c2 * int B B'
        END IF
        ALLOCATE( dip(n1_fin-n0_fin+1), T_fi(n0_fin:n1_fin) )
        rc = bspatom_dipole_elements(prob, INT(l0,C_INT), INT(n0_ini,C_INT), INT(lf,C_INT), INT(n0_fin,C_INT),            &
     &                               INT(n1_fin-n0_fin+1,C_INT), avec, dip)
        IF( rc /= 0 ) THEN
          WRITE(6,*) 'bsp_atom_host: bspatom_dipole_elements failed, code ', rc
          STOP 1
        END IF
        DO ni = n0_fin, n1_fin
          An = SQRT( 2.D0 / (En(lf*nfun+ni+1) - En(lf*nfun+ni-1)) )
          T_fi(ni) = An * c0 * dip(ni-n0_fin+1)
        END DO
!       CROSS_SECTIONS (PhotoIon.f90:274-468), plane-wave branches
        WRITE(6,'(/,A26)') 'Calculating Cross Sections'
        WRITE(6,*) 'E0=', En(l0*nfun+n0_ini)
        M_au = (a_au**2) * 1.0D18
        cc0 = 4.D0 * (PI**2) / (c_au)
        cc1 = 1.D0 / DBLE(2*l0 + 1)
        IF( KIND_PI == 1 ) THEN
          OPEN( UNIT=30, FILE='CSs/CrossSection_Len.dat', ACTION='WRITE' )
        ELSE
          OPEN( UNIT=30, FILE='CSs/CrossSection_Vel.dat', ACTION='WRITE' )
        END IF
        DO nf = n0_fin, n1_fin
          d1 = En(lf*nfun+nf) - En(l0*nfun+n0_ini)
          IF( KIND_PI == 2 ) d1 = 1.D0 / d1
          Ef = En(lf*nfun+nf)
          d2 = T_fi(nf)**2
          csl = M_au * cc0 * cc1 * d1 * d2
          WRITE(30,400) Ef, csl
        END DO
        CLOSE(30)
      END IF
      CALL bspatom_problem_destroy(prob)
      IF( KIND_PI <= 2 ) WRITE(6,'(/,A17)') 'Program Finished!'

100   FORMAT(/,T2,A5,I2)
110   FORMAT(T2,A26,/)
120   FORMAT(T5,A1,T9,A11)
200   FORMAT(T2,I4,T8,G22.15)
300   FORMAT(I5,5000G20.10)
400   FORMAT(2G20.10E3)
      END PROGRAM BSP_ATOM_MI355X

!     Wigner 3j symbol for integer arguments: Racah's sum with log-factorials, the smallest exponent taken out of the
!     alternating sum (the scheme of the reference's THREE_J, Funs_WignerSymbols.for:1-62, so the two agree to rounding)
      FUNCTION W3J(j1, j2, j3, m1, m2, m3) RESULT(w)
      IMPLICIT NONE
      INTEGER, PARAMETER :: DP = KIND(1.0D0)
      INTEGER, INTENT(IN) :: j1, j2, j3, m1, m2, m3
      REAL(DP) :: w, lf(0:200), delta, g, acc, ex
      INTEGER :: i, z, zmin, zmax
      w = 0.D0
      IF( m1 + m2 + m3 /= 0 ) RETURN
      IF( j1 + j2 + j3 + 1 > 200 ) RETURN
      lf(0) = 0.D0
      DO i = 1, j1 + j2 + j3 + 1
        lf(i) = lf(i-1) + LOG(DBLE(i))
      END DO
      zmin = MAX(0, j2 - j3 - m1, j1 + m2 - j3)
      zmax = MIN(j1 + j2 - j3, j1 - m1, j2 + m2)
      IF( zmax < zmin ) RETURN
      delta = 0.5D0 * ( lf(j1+j2-j3) + lf(j3+j1-j2) + lf(j3+j2-j1) - lf(j1+j2+j3+1)                                      &
     &                + lf(j1+m1) + lf(j1-m1) + lf(j2+m2) + lf(j2-m2) + lf(j3+m3) + lf(j3-m3) )
      g = 250.D0
      DO z = zmin, zmax
        ex = lf(z) + lf(j1+j2-j3-z) + lf(j1-m1-z) + lf(j2+m2-z) + lf(j3-j2+m1+z) + lf(j3-j1-m2+z)
        g = MIN(g, ex)
      END DO
      acc = 0.D0
      DO z = zmin, zmax
        ex = lf(z) + lf(j1+j2-j3-z) + lf(j1-m1-z) + lf(j2+m2-z) + lf(j3-j2+m1+z) + lf(j3-j1-m2+z)
        acc = acc + (-1.D0)**z * EXP(-(ex - g))
      END DO
      w = (-1.D0)**(j1-j2-m3) * EXP(delta - g) * acc
      END FUNCTION W3J

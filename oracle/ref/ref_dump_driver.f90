!  ref_dump_driver.f90 -- TEST INFRASTRUCTURE (oracle/_ref build only).
!  Drives the *reference's own* routines in the order of PROGRAM BSP_ATOM_PI for
!  KIND_PI = 0 (Bsp_Atom.f90:45-75: READ_INPUTS, GRID, SEL_LM, MATRIX_SVT, SOLVE_SYSTEM)
!  and dumps the module state they leave behind as one unformatted stream file
!  `ref_dump.bin`, from which tests/golden/make_golden.py extracts fixtures.
!  Nothing here computes anything; it only reads the reference's module variables.
      PROGRAM REF_DUMP
      USE MOD_TYPES
      USE MOD_GRID
      USE MOD_BSPLINES
      USE MOD_MATRICES
      USE MOD_PHOTOION
      IMPLICIT NONE
      INTEGER :: c0, c1, c2, crate, envlen
      CHARACTER(LEN=8) :: envbuf
      CALL READ_INPUTS
      CALL GRID
      CALL SEL_LM
      CALL SYSTEM_CLOCK(c0, crate)
      CALL MATRIX_SVT
      CALL SYSTEM_CLOCK(c1)
      OPEN(UNIT=91, FILE='ref_dump.bin', ACCESS='STREAM', FORM='UNFORMATTED', ACTION='WRITE')
      WRITE(91) nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax
      WRITE(91) rt(1:nkp)
      WRITE(91) xg(1:ka), wg(1:ka)
      WRITE(91) Aind(1:nfun,1:2)
      WRITE(91) Sij, Tij, Vij, Uij
      CLOSE(91)
!     KIND_PI = 1, 2: the dipole matrices MATRIX_SVT leaves in rij (matrices.f90:159-163), for SURVEY 8(f).2
      IF( ALLOCATED(rij) .AND. KIND_PI >= 1 .AND. KIND_PI <= 2 ) THEN
        OPEN(UNIT=92, FILE='ref_rij.bin', ACCESS='STREAM', FORM='UNFORMATTED', ACTION='WRITE')
        WRITE(92) nfun, KIND_PI
        WRITE(92) rij
        CLOSE(92)
      END IF
      CALL SOLVE_SYSTEM
      CALL SYSTEM_CLOCK(c2)
!     KIND_PI = 1, 2: PROGRAM BSP_ATOM_PI continues into TRANS_AMP (Bsp_Atom.f90:77-80).  The unmodified routine
!     dies in a diagnostic WRITE of an unallocated array (see build_ref.sh); with that line shortened it runs, and
!     the amplitudes T_fi(n0_fin:n1_fin, l_fin) it leaves in the module are dumped for SURVEY 8(f).2.
!     CROSS_SECTIONS (Bsp_Atom.f90:88) as written reads two module variables SOLVE_SYSTEM sets for KIND_PI >= 3 only:
!     Enl(n0,l0) (PhotoIon.f90:302, printed as E0 and used for KIND_PI >= 5 only) and the loop bound n1_max (:385, the
!     records are written for nf <= n1_fin, :408).  This driver gives both the values the routine evidently means --
!     Enl(:,l_ini) = E_ini, Enl(:,l_fin) = E_fin, n1_max = n1_fin -- and then calls the reference's OWN routine, so
!     that CSs/CrossSection_Len.dat / _Vel.dat (SURVEY 8(f).2) are pinned by the reference's arithmetic and FORMAT.
      IF( KIND_PI == 1 .OR. KIND_PI == 2 ) THEN
        CALL TRANS_AMP
        OPEN(UNIT=94, FILE='ref_tfi.bin', ACCESS='STREAM', FORM='UNFORMATTED', ACTION='WRITE')
        WRITE(94) nfun, KIND_PI, n0_ini, lmf(0,1), lmf(0,2), lmf(nlm,1), lmf(nlm,2), mph, n0_fin, n1_fin
        WRITE(94) Emax_fin
        WRITE(94) E_ini(1:nfun), E_fin(1:nfun)
        WRITE(94) ci_ini(1:nfun)
        WRITE(94) ci_fin(1:nfun,n0_fin:n1_fin)
        WRITE(94) T_fi(n0_fin:n1_fin,lmf(nlm,1))
        CLOSE(94)
        IF( .NOT. ALLOCATED(Enl) ) ALLOCATE( Enl(nfun,0:lmax) )
        Enl = 0.D0
        Enl(:,l_ini) = E_ini(:)
        Enl(:,lmf(nlm,1)) = E_fin(:)
        n1_max = n1_fin
        CALL EXECUTE_COMMAND_LINE('mkdir -p CSs')
        CALL CROSS_SECTIONS
      END IF
!     KIND_PI >= 3: the state-selection bookkeeping SOLVE_SYSTEM leaves in the module (matrices.f90:290-358):
!     n01(l,1:3), n1_max, Emax_fin as modified, the spectra Enl and the density-of-states factors rEki, for
!     SURVEY 8(f).1 (the eigenvectors themselves are read from the Eigenvec_All.dat the same run wrote)
      IF( KIND_PI >= 3 .AND. ALLOCATED(n01) ) THEN
        OPEN(UNIT=93, FILE='ref_pi3.bin', ACCESS='STREAM', FORM='UNFORMATTED', ACTION='WRITE')
        WRITE(93) nfun, lmax, n1_max, KIND_PI
        WRITE(93) n01(0:lmax,1:3)
        WRITE(93) Emax_fin
        WRITE(93) Enl(1:nfun,0:lmax)
        WRITE(93) rEki(1:nfun,0:lmax)
        CLOSE(93)
      END IF
!     KIND_PI >= 3, on request (REF_DUMP_TRANS_AMP_PI3 set): the reference's own TRANS_AMP as far as it runs in this container.
!     Its Gaussian / LG-beam branch needs MAKE_F_ANG (Ang_Ints.f90), which aborts with a heap corruption under flang -O2
!     (Bsp_Atom.f90:66 is not called here); without it ncomp = nket = 0 and TRANS_AMP writes the header of
!     CSs/MatElem_All.dat (PhotoIon.f90:255-256) and no record -- that header is what the hand-off fixture keeps of it.
      CALL GET_ENVIRONMENT_VARIABLE('REF_DUMP_TRANS_AMP_PI3', envbuf, envlen)
      IF( KIND_PI >= 3 .AND. envlen > 0 ) THEN
        CALL EXECUTE_COMMAND_LINE('mkdir -p CSs')
        CALL TRANS_AMP
      END IF
      WRITE(6,'(A,F12.4)') 'REF_TIME_MATRIX_SVT_S ', DBLE(c1-c0)/DBLE(crate)
      WRITE(6,'(A,F12.4)') 'REF_TIME_SOLVE_SYSTEM_S ', DBLE(c2-c1)/DBLE(crate)
      END PROGRAM REF_DUMP

!  ref_handoff_driver.f90 -- TEST INFRASTRUCTURE (SURVEY 8(f).3): the hand-off files of the structure run, consumed by the
!  REFERENCE'S OWN reader and produced by the reference's own WRITE forms.
!
!  mode 'R': calls the unmodified READ_COUP (ReadInputs.f90:277-369, the reader the TDSE tools use) in a directory that holds
!            Enl.dat and CSs/MatElem_All.dat -- written by bspatom_amd/host.py -- and dumps what it read (ref_handoff.bin):
!            nfun, n1_max, nbra, nket, nfields, Enl(nfun,0:lmax), n01(0:lmax,3), zHint_ij(nbra,nket,nfields).
!  mode 'W': writes MatElem_All.dat from numbers on stdin with the two statement forms TRANS_AMP uses (PhotoIon.f90:255-266:
!            list-directed header, FORMAT(2I8,X,20G20.10) records), i.e. the bytes this Fortran runtime produces for them.
!            (The reference itself cannot write a non-empty file in this container: for KIND_PI >= 3 its MAKE_F_ANG,
!            Ang_Ints.f90, aborts with a heap corruption under flang -O2, and without it ncomp = 0: DESIGN.md section 2.)
!  stdin: mode; then for 'R': lmax, KIND_PI, Emax_fin, n0_ini, l0, m0;  for 'W': n1_max, nbra, nket, ncomp, then the complex
!  values zT(ibra,jket,i) in the order of the records.
      PROGRAM REF_HANDOFF
      USE MOD_TYPES
      USE MOD_GRID, ONLY: lmax
      USE MOD_BSPLINES, ONLY: nfun
      USE MOD_PHOTOION
      IMPLICIT NONE
      CHARACTER(LEN=1) :: mode
      INTEGER :: l0, m0, ib, jk, i, nc
      COMPLEX(DPC), ALLOCATABLE :: zT(:,:,:)
      REAL(DP) :: re, im
      READ(5,*) mode
      IF( mode == 'R' ) THEN
        READ(5,*) lmax, KIND_PI, Emax_fin, n0_ini, l0, m0
        ALLOCATE( lmf(0:0,2) )
        lmf(0,1) = l0; lmf(0,2) = m0
        CALL READ_COUP
        OPEN(UNIT=95, FILE='ref_handoff.bin', ACCESS='STREAM', FORM='UNFORMATTED', ACTION='WRITE')
        WRITE(95) nfun, n1_max, nbra, nket, nfields, lmax
        WRITE(95) Enl(1:nfun,0:lmax)
        WRITE(95) n01(0:lmax,1:3)
        WRITE(95) zHint_ij(1:nbra,1:nket,1:nfields)
        CLOSE(95)
      ELSE
        READ(5,*) n1_max, nbra, nket, nc
        ALLOCATE( zT(nbra,nket,nc) )
        zT = 0.D0
        DO ib = 1, nbra
          DO jk = ib, nket
            DO i = 1, nc
              READ(5,*) re, im
              zT(ib,jk,i) = DCMPLX(re,im)
            END DO
          END DO
        END DO
        OPEN( UNIT=60, FILE='MatElem_All.dat', ACTION='WRITE' )
        WRITE(60,*) n1_max, nbra, nket
        DO ib = 1, nbra
          DO jk = ib, nket
            WRITE(60,500) ib, jk, (zT(ib,jk,i), i=1,nc)
          END DO
        END DO
        CLOSE(60)
      END IF
500   FORMAT(2I8,X,20G20.10)
      END PROGRAM REF_HANDOFF

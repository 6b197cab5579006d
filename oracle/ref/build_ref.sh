#!/usr/bin/env bash
# build_ref.sh -- compile the UNMODIFIED reference sources where they lie
# (/root/reference/src) into oracle/_ref/ (git-ignored; binaries only).  TEST INFRASTRUCTURE.
#
# Outputs:
#   oracle/_ref/Bsp_Atom_ref.x  the reference program itself (stdin namelist -> stdout,
#                               Enl.dat, wf_n0.dat), `INQUIRE(DIRECTORY=` (an ifort-only
#                               extension, Bsp_Atom.f90:59) spelled `INQUIRE(FILE=` in a
#                               temporary stream so that flang accepts it;
#   oracle/_ref/ref_dump.x      ref_dump_driver.f90 + the same reference objects (PhotoIon.f90 with one
#                               diagnostic WRITE shortened, see below); also writes ref_dump.bin
#                               (module state) for the golden fixtures.
#   oracle/_ref/ref_handoff.x   ref_handoff_driver.f90 + the reference objects: the reference's READ_COUP (ReadInputs.f90:277-369)
#                               reads Enl.dat / CSs/MatElem_All.dat the PRODUCT wrote; its WRITE forms for MatElem_All.dat.
#   oracle/_ref/Bsp_Atom_gpu.x  THE SAME reference objects as Bsp_Atom_ref.x, linked with -lbspatom_lapack in front of
#                               the CPU LAPACK: `dsygv_` (matrices.f90:248) resolves to the GPU library, every
#                               other BLAS symbol still to OpenBLAS -- the link-level drop-in of INTEGRATION.md 1,
#                               built when bspatom_amd/libbspatom_lapack.so exists.
# LAPACK/BLAS: scipy's bundled OpenBLAS (LAPACK 3.12.0) through lapack_forward.c, since the
# reference's `-mkl` is not in this image.  No reference source is copied into the repo:
# the two filtered translation units live in a mktemp dir that is removed on exit.
set -euo pipefail
REF=${BSP_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
[ -d "$REF/src" ] || { echo "build_ref: $REF/src not present (GPU box?) - skipping"; exit 0; }
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
[ -x "$FC" ] || { echo "build_ref: no flang at $FC - skipping"; exit 0; }
SCIPY_LIBS=$(python3 -c "import scipy,os;print(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)),'scipy.libs'))")
OPENBLAS=$(ls "$SCIPY_LIBS"/libscipy_openblas*.so | head -1)
mkdir -p "$OUT/obj"
TMP=$(mktemp -d /tmp/bspref.XXXXXX); trap 'rm -rf "$TMP"' EXIT
cd "$OUT/obj"
FFLAGS="-O2"
"$FC" $FFLAGS -c "$REF/src/Modules.f90" -o Modules.o
for f in ReadInputs matrices PhotoIon WriteWF grid CubicSpline bsplvb interv Ang_Ints Ang_Ints_Aux \
         TorusFuns TorusFunsInts Funs_AssLegendre Funs_AssLaguerre Funs_SphHarms Funs_Bessel; do
  "$FC" $FFLAGS -c "$REF/src/$f.f90" -o $f.o
done
"$FC" $FFLAGS -ffixed-form -c "$REF/src/Funs_WignerSymbols.for" -o Funs_WignerSymbols.o
# the program unit, one token made portable
sed "s/INQUIRE( DIRECTORY='CSs'/INQUIRE( FILE='CSs\/.'/" "$REF/src/Bsp_Atom.f90" > "$TMP/main_unit.f90"
"$FC" $FFLAGS -c "$TMP/main_unit.f90" -o Bsp_Atom.o
# WRITE_WF / END_PROG only (everything after the program unit), for the dump driver
sed -n '/^ *SUBROUTINE WRITE_WF/,$p' "$REF/src/Bsp_Atom.f90" > "$TMP/subs_unit.f90"
"$FC" $FFLAGS -c "$TMP/subs_unit.f90" -o Bsp_Atom_subs.o
# TRANS_AMP prints Enl(1,lf), Enl(n1_max,lf) (PhotoIon.f90:47), an array SOLVE_SYSTEM allocates for KIND_PI >= 3 only:
# with flang the KIND_PI = 1, 2 run dies in that WRITE.  For the dump driver only, the two items are dropped from the
# WRITE in a temporary stream (a diagnostic line; nothing computed changes), so that T_fi can be pinned
sed "s/'Energy limits Final State:', Enl(1,lf), Enl(n1_max,lf)/'Energy limits Final State:'/" "$REF/src/PhotoIon.f90" > "$TMP/photoion_unit.f90"
"$FC" $FFLAGS -c "$TMP/photoion_unit.f90" -o PhotoIon_dump.o
"$FC" $FFLAGS -c "$HERE/ref_dump_driver.f90" -o ref_dump_driver.o
gcc -O2 -c "$HERE/lapack_forward.c" -o lapack_forward.o
COMMON="Modules.o ReadInputs.o matrices.o PhotoIon.o WriteWF.o grid.o CubicSpline.o bsplvb.o interv.o \
 Ang_Ints.o Ang_Ints_Aux.o TorusFuns.o TorusFunsInts.o Funs_AssLegendre.o Funs_AssLaguerre.o \
 Funs_SphHarms.o Funs_Bessel.o Funs_WignerSymbols.o lapack_forward.o"
LINK="$OPENBLAS -Wl,-rpath,$SCIPY_LIBS -lm"
"$FC" -o "$OUT/Bsp_Atom_ref.x" Bsp_Atom.o $COMMON $LINK
"$FC" -o "$OUT/ref_dump.x" ref_dump_driver.o Bsp_Atom_subs.o ${COMMON/PhotoIon.o/PhotoIon_dump.o} $LINK
# hand-off formats: the reference's own reader READ_COUP on files the product wrote, see the driver's header
"$FC" $FFLAGS -c "$HERE/ref_handoff_driver.f90" -o ref_handoff_driver.o
"$FC" -o "$OUT/ref_handoff.x" ref_handoff_driver.o Bsp_Atom_subs.o $COMMON $LINK
# link-level drop-in: the reference's own objects against libbspatom_lapack.so (dsygv_ -> GPU).  The forwarder object
# is compiled WITHOUT its dsygv_ (-DNO_DSYGV) so that the only definition of that symbol is the GPU library's.
PROD=$(cd "$HERE/../../bspatom_amd" && pwd)
if [ -f "$PROD/libbspatom_lapack.so" ]; then
  gcc -O2 -DNO_DSYGV -c "$HERE/lapack_forward.c" -o lapack_forward_nodsygv.o
  "$FC" -o "$OUT/Bsp_Atom_gpu.x" Bsp_Atom.o ${COMMON/lapack_forward.o/lapack_forward_nodsygv.o} \
        -L"$PROD" -lbspatom_lapack -lbspatom -Wl,-rpath,"$PROD" $LINK
  echo "build_ref: built $OUT/Bsp_Atom_gpu.x (dsygv_ from $PROD/libbspatom_lapack.so)"
fi
rm -f *.mod
echo "build_ref: built $OUT/Bsp_Atom_ref.x and $OUT/ref_dump.x (LAPACK: $OPENBLAS)"

#!/usr/bin/env bash
# TEST INFRASTRUCTURE -- builds the *reference* Fortran into oracle/_ref/libspeedy_ref.so.
#
# The reference sources are compiled where they lie under /root/reference/speedy.f90 (read-only);
# objects and .mod files go to a scratch directory under /tmp, and the only output kept is
# oracle/_ref/libspeedy_ref.so (git-ignored; it travels to the GPU box like any built .so).
# No reference source is copied into this repository.
#
# Compiler: amdflang (LLVM flang, ROCm 7.2) -- gfortran is not installed in this image.
# One compatibility edit is required for flang: initialization.f90:30 declares
# `intent(out) :: control_params`; gfortran leaves the components set earlier by controlparams_init
# untouched, flang re-initialises the object.  The edit (intent(out) -> intent(inout)) is applied by `sed`
# on the way into the scratch directory; the file under /root/reference is never modified.
set -euo pipefail

REF=${REF_SRC:-/root/reference/speedy.f90}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
SCRATCH=${REF_SCRATCH:-/tmp/pyspeedy_ref_build}
FC=${FC:-/opt/rocm/bin/amdflang}
FFLAGS=${FFLAGS:--O2 -cpp -fPIC -w}

if [ ! -d "$REF" ]; then
    echo "build_ref.sh: $REF not present (GPU box?) -- keeping prebuilt $OUT/libspeedy_ref.so" >&2
    exit 0
fi

mkdir -p "$OUT" "$SCRATCH"
cd "$SCRATCH"

ORDER="types params error_codes physical_constants mod_radcon geometry legendre fftpack fourier spectral
matrix_inversion horizontal_diffusion implicit model_control model_state interpolation boundaries humidity
convection large_scale_condensation shortwave_radiation longwave_radiation land_model sea_model
surface_fluxes vertical_diffusion sppt coupler physics geopotential diagnostics prognostics tendencies
time_stepping forcing initialization speedy speedy_driver"

OBJS=""
for f in $ORDER; do
    src="$REF/$f.f90"
    if [ "$f" = "initialization" ]; then
        # flang compatibility edit, see header.  Compiled from stdin-equivalent scratch file.
        sed 's/type(ControlParams_t), intent(out) :: control_params/type(ControlParams_t), intent(inout) :: control_params/' \
            "$src" > "$SCRATCH/_initialization_flang.f90"
        src="$SCRATCH/_initialization_flang.f90"
    fi
    if [ ! -f "$f.o" ] || [ "$src" -nt "$f.o" ]; then
        $FC $FFLAGS -c "$src" -o "$f.o"
    fi
    OBJS="$OBJS $f.o"
done

$FC $FFLAGS -c "$HERE/ref_shim.f90" -o ref_shim.o
$FC -shared -o "$OUT/libspeedy_ref.so" $OBJS ref_shim.o
rm -f "$SCRATCH/_initialization_flang.f90"
echo "built $OUT/libspeedy_ref.so"

#!/bin/bash
# Builds libiamx.so (gfx950 only) next to the python package.  Cross-compiles without a GPU.
# IAMX_ABLATE=1 builds libiamx_ablate.so instead: the same library plus the iamxdbg_* timing
# variants used by tools/*_ablate.py (never loaded by the product; select it with IAMX_LIB).
# IAMX_REBUILD=1 ignores the object cache: every source is compiled (what __graft_entry__.build()
# asks for, so that a build check proves the SOURCES build, not that stale objects link).
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OUT="$HERE/../libiamx.so"
OBJDIR="$HERE/obj"
if [ -n "$IAMX_ABLATE" ]; then
    OUT="$HERE/../libiamx_ablate.so"
    OBJDIR="$HERE/obj_ablate"
    FLAGS="$FLAGS -DIAMX_ABLATE"
fi
SRCS="$HERE/common.hip $HERE/match_knn2.hip $HERE/match_knn2v2.hip $HERE/match_knn2sym.hip $HERE/match_post.hip $HERE/host_cleanup.hip $HERE/triangulate.hip $HERE/ba_kernels.hip $HERE/ba_linalg.hip $HERE/ba_schur.hip $HERE/trf_vec.hip $HERE/comm.hip $HERE/sift.hip $HERE/image_prep.hip $HERE/jpeg.hip $HERE/cache_codec.hip"
mkdir -p "$OBJDIR"
OBJS=""
for f in $SRCS; do
    o="$OBJDIR/$(basename ${f%.hip}).o"
    if [ -n "$IAMX_REBUILD" ] || [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/iamx_common.h" -nt "$o" ] || [ "$HERE/../../include/iamx.h" -nt "$o" ]; then
        EXTRA=""
        # the TRF helpers restate numpy expressions: separately rounded multiply and add
        [ "$(basename $f)" = "trf_vec.hip" ] && EXTRA="-ffp-contract=off"
        # the one-wave-per-SIMD sweep (form 2; ablation variants 600-602) needs its MFMA
        # accumulators in VGPRs (the allocator's default for > 256 registers is the AGPR half,
        # at a v_accvgpr_read per element the VALU touches)
        [ "$(basename $f)" = "match_knn2sym.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
        $HIPCC $FLAGS $EXTRA ${IAMX_EXTRA_FLAGS} -c "$f" -o "$o" &
    fi
    OBJS="$OBJS $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS -lz -lpthread
echo "built $OUT"

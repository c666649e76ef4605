#!/bin/bash
# Builds libiamx.so (gfx950 only) next to the python package.  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libiamx.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
SRCS="$HERE/common.hip $HERE/match_knn2.hip $HERE/match_knn2v2.hip $HERE/match_post.hip $HERE/host_cleanup.hip $HERE/triangulate.hip $HERE/ba_kernels.hip $HERE/ba_linalg.hip $HERE/sift.hip $HERE/image_prep.hip"
mkdir -p "$HERE/obj"
OBJS=""
for f in $SRCS; do
    o="$HERE/obj/$(basename ${f%.hip}).o"
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/iamx_common.h" -nt "$o" ] || [ "$HERE/../../include/iamx.h" -nt "$o" ]; then
        $HIPCC $FLAGS ${IAMX_EXTRA_FLAGS} -c "$f" -o "$o" &
    fi
    OBJS="$OBJS $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $OUT"

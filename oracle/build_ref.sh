#!/usr/bin/env bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
#
# Builds oracle/_ref/libmoped_ref.so (golden numerics: no -ffast-math) and
# oracle/_ref/libmoped_ref_fast.so (the reference's own optimisation flags,
# moped2/libmoped/Makefile:44-47 minus -march=native so the binary also runs on
# the GPU box's host CPU) from the reference sources WHERE THEY LIE:
#   /root/reference/moped2/libmoped/include/moped.hpp         (header, -I)
#   /root/reference/moped2/libmoped/libs/libs.tgz             (vendored ANN 1.1.1,
#        levmar 2.4, libsiftfast 1.1 -- unpacked into a temp dir outside the repo,
#        compiled with gcc/g++ directly, temp dir removed afterwards)
# plus our own wrapper oracle/ref_harness.cpp.  The reference's build system is
# not run, nothing from the reference is copied into the repo, and the only
# outputs are the two .so files under oracle/_ref/ (git-ignored).
#
# No-op (exit 0) when /root/reference is absent (GPU box): the prebuilt files
# travel with the snapshot.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${MOPED_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
LIBMOPED="$REF/moped2/libmoped"
if [ ! -f "$LIBMOPED/libs/libs.tgz" ]; then
  echo "build_ref: $LIBMOPED not present; keeping prebuilt oracle/_ref as is"
  exit 0
fi
mkdir -p "$OUT"
if [ "$OUT/libmoped_ref.so" -nt "$HERE/ref_harness.cpp" ] && \
   [ "$OUT/libmoped_ref_fast.so" -nt "$HERE/ref_harness.cpp" ] && \
   [ "$OUT/libmoped_ref.so" -nt "$HERE/build_ref.sh" ]; then
  echo "build_ref: up to date"
  exit 0
fi
TMP="$(mktemp -d /tmp/moped_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
tar xzf "$LIBMOPED/libs/libs.tgz" -C "$TMP" levmar-2.4 ann_1.1.1 libsiftfast-1.1-src

build_variant() {  # $1 = output name, $2.. = optimisation flags
  local name="$1"; shift
  local obj="$TMP/obj_$name"; mkdir -p "$obj"
  local f
  # levmar: LAPACK is #undef'd in lm.h:28-29 -> built-in LU (Axb_core.c)
  for f in lm Axb misc; do
    gcc "$@" -w -fPIC -I"$TMP/levmar-2.4" -c "$TMP/levmar-2.4/$f.c" -o "$obj/lev_$f.o"
  done
  for f in "$TMP"/ann_1.1.1/*.cpp "$TMP"/ann_1.1.1/ANN/ANN.cpp; do
    g++ "$@" -w -fPIC -I"$TMP/ann_1.1.1" -I"$TMP/ann_1.1.1/ANN" -c "$f" -o "$obj/ann_$(basename "$f" .cpp).o"
  done
  g++ "$@" -w -fPIC -fopenmp -msse2 -I"$TMP/libsiftfast-1.1-src" \
      -c "$TMP/libsiftfast-1.1-src/libsiftfast.cpp" -o "$obj/siftfast.o"
  g++ "$@" -w -fPIC -std=gnu++98 -fno-delete-null-pointer-checks -fopenmp \
      -I"$LIBMOPED/include" -I"$TMP/ann_1.1.1/ANN" -I"$TMP/levmar-2.4" -I"$TMP/libsiftfast-1.1-src" \
      -c "$HERE/ref_harness.cpp" -o "$obj/harness.o"
  g++ -shared -fopenmp -o "$OUT/$name" "$obj"/*.o -lm
}

build_variant libmoped_ref.so -O2
build_variant libmoped_ref_fast.so -O3 -mfpmath=sse -funroll-loops -ffast-math -fno-trapping-math
echo "build_ref: wrote $OUT/libmoped_ref.so $OUT/libmoped_ref_fast.so"

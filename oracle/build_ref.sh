#!/usr/bin/env bash
# build_ref.sh -- compile the REAL reference (Piezoid/Blight) into oracle/_ref/ as two C-ABI shared libraries.
#
# TEST INFRASTRUCTURE.  The reference's sources are compiled where they lie under $REF (default /root/reference);
# nothing of them is copied into this repository.  The only intermediate is a sed-patched copy of kmer.h holding the
# three one-token fixes of SURVEY.md App. B-1 (and the reference's own kmer_t typedef switch for the 128-bit build);
# it lives in a mktemp directory that is removed before this script returns.
#
#   P1  kmer.h:796, kmer.h:802   canonize(mmer, k) -> canonize(mmer, m)          (shipped query path crashes)
#   P2  kmer.h:708               _mini.fill(it) -> _mini.fill(dna_ascii_iter<const char*>(it))
#   T128 kmer.h:18               kmer_t = uint64_t -> __uint128_t                  (k > 32)
#
# The reference's own build system (makefile) is not run; flags follow makefile:9,18 except -march (portable v3 so the
# binaries also run on the GPU box's host) and -flto (single TU).
set -euo pipefail
REF="${REF:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF/blight.cpp" ]; then
	echo "build_ref.sh: $REF not present -- keeping any prebuilt oracle/_ref" >&2
	exit 0
fi
mkdir -p "$OUT"
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/p64" "$T/p128"
sed -e '796s/canonize(mmer, k)/canonize(mmer, m)/' -e '802s/canonize(mmer, k)/canonize(mmer, m)/' \
	-e '708s/_mini.fill(it);/_mini.fill(dna_ascii_iter<const char*>(it));/' "$REF/kmer.h" >"$T/p64/kmer.h"
sed -e '18s/= uint64_t;/= __uint128_t;/' "$T/p64/kmer.h" >"$T/p128/kmer.h"
# the patches must have changed exactly lines 708, 796, 802 (+18)
n64=$(diff "$REF/kmer.h" "$T/p64/kmer.h" | grep -c '^<' || true)
n128=$(diff "$REF/kmer.h" "$T/p128/kmer.h" | grep -c '^<' || true)
if [ "$n64" != "3" ] || [ "$n128" != "4" ]; then
	echo "build_ref.sh: reference kmer.h does not match the expected revision (patched $n64/$n128 lines)" >&2
	exit 1
fi
CXXFLAGS="-DNDEBUG -O3 -march=x86-64-v3 -fopenmp -fPIC -shared -w"
g++ -std=c++11 $CXXFLAGS -DREF_PATCHED_KMER_H="\"$T/p64/kmer.h\"" -I"$REF" "$HERE/ref_harness.cpp" -lz -o "$OUT/libblight_ref64.so"
g++ -std=gnu++11 $CXXFLAGS -DREF_T128 -DREF_PATCHED_KMER_H="\"$T/p128/kmer.h\"" -I"$REF" "$HERE/ref_harness.cpp" -lz -o "$OUT/libblight_ref128.so"
g++ -std=c++11 -O2 -fPIC -shared -w -I"$REF" "$HERE/ref_murmur.cpp" "$REF/MurmurHash3.cpp" -o "$OUT/libmurmur3_ref.so"
echo "built $OUT/libblight_ref64.so $OUT/libblight_ref128.so $OUT/libmurmur3_ref.so"

#!/bin/bash
# Build the REAL reference (shubhamchandak94/HARC) from its own sources where they lie under
# /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU box as prebuilt binaries).
# Nothing is copied: reorder.cpp / encoder.cpp `#include "config.h"`, which the reference's bash
# driver generates per run (harc:52-63); we generate the same macros into a per-config include dir
# and pass it with -I.  BBHash (src/BooPHF.h) is picked up next to the sources.
#
#   oracle/build_ref.sh            -> tools (preprocess, decoder, pack_order, unpack_order, merge_N, generators)
#   oracle/build_ref.sh L T        -> reorder_L<L>_t<T>.out, encoder_L<L>_t<T>.out
#   oracle/build_ref.sh quality L     -> reorder_quality_L<L>.out (-q without -p, harc:121-124)
#   oracle/build_ref.sh preserve L E  -> decoder_preserve_L<L>_e<E>.out (the -p decoder, harc:173-178; num_thr 1)
set -e
REF=${HARC_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ ! -d "$REF/src" ]; then echo "reference sources not present at $REF (ok on the GPU box)"; exit 0; fi
CXXFLAGS="-w -march=x86-64-v2 -O3 -fopenmp -std=c++11"   # harc:65 uses -march=native; v2 so the binary runs on the GPU box's host too
if [ $# -eq 0 ]; then
  for p in preprocess decoder pack_order unpack_order; do
    [ -x "$OUT/$p.out" ] || g++ "$REF/src/$p.cpp" $CXXFLAGS -o "$OUT/$p.out"
  done
  [ -x "$OUT/merge_N.out" ] || g++ "$REF/src/merge_N.cpp" -w -O3 -std=c++11 -o "$OUT/merge_N.out"
  [ -x "$OUT/gen_fastq_noRC" ] || g++ -w -std=c++11 -O3 -o "$OUT/gen_fastq_noRC" "$REF/util/gen_fastq_noRC/gen_fastq_noRC.cpp"
  [ -x "$OUT/gen_fastq" ] || g++ -w -std=c++11 -O3 -o "$OUT/gen_fastq" "$REF/util/gen_fastq/gen_fastq.cpp"
  exit 0
fi
if [ "$1" = quality ]; then
  L=$2
  CFG="$OUT/cfgq_L${L}"
  mkdir -p "$CFG"
  echo "#define readlen $L" > "$CFG/config.h"
  [ -x "$OUT/reorder_quality_L${L}.out" ] || g++ "$REF/src/reorder_quality.cpp" -I"$CFG" -w -march=x86-64-v2 -O3 -std=c++11 -o "$OUT/reorder_quality_L${L}.out"   # harc:123
  exit 0
fi
if [ "$1" = preserve ]; then
  L=$2; E=$3
  CFG="$OUT/cfgp_L${L}_e${E}"
  mkdir -p "$CFG"
  { echo "#define readlen $L"; echo "#define MAX_BIN_SIZE 7"; echo "#define num_thr 1"; echo "#define num_thr_e $E"; } > "$CFG/config.h"   # harc:174-177
  [ -x "$OUT/decoder_preserve_L${L}_e${E}.out" ] || g++ "$REF/src/decoder_preserve.cpp" -I"$CFG" $CXXFLAGS -o "$OUT/decoder_preserve_L${L}_e${E}.out"
  exit 0
fi
L=$1; T=$2
CFG="$OUT/cfg_L${L}_t${T}"
mkdir -p "$CFG"
{ echo "#define maxmatch $((L/2))"; echo "#define thresh 4"; echo "#define thresh_s 24"; echo "#define numdict 2"
  echo "#define maxsearch 1000"
  echo "#define dict1_start $(( L > 100 ? L/2-32 : L/2-L*32/100 ))"; echo "#define dict1_end $((L/2-1))"
  echo "#define dict2_start $((L/2))"; echo "#define dict2_end $(( L > 100 ? L/2-1+32 : L/2-1+L*32/100 ))"
  echo "#define readlen $L"; echo "#define num_thr $T"; } > "$CFG/config.h"          # harc:52-63
[ -x "$OUT/reorder_L${L}_t${T}.out" ] || g++ "$REF/src/reorder.cpp" -I"$CFG" $CXXFLAGS -lpthread -o "$OUT/reorder_L${L}_t${T}.out"
[ -x "$OUT/encoder_L${L}_t${T}.out" ] || g++ "$REF/src/encoder.cpp" -I"$CFG" $CXXFLAGS -lpthread -o "$OUT/encoder_L${L}_t${T}.out"

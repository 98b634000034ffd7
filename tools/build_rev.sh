#!/bin/bash
# Build libmggan_hip.so of another revision (same C ABI) for same-box A/B runs through MGGAN_HIP_LIB:
#   bash tools/build_rev.sh HEAD scratch/lib_head.so
REV=${1:-HEAD}; OUT=$(realpath -m ${2:-scratch/lib_rev.so}); T=$(mktemp -d)
mkdir -p $T/mg-gan_amd/csrc $T/mg-gan_amd/mggan/hip $T/include $(dirname $OUT)
git archive $REV mg-gan_amd/csrc include | tar -x -C $T
make -s -C $T/mg-gan_amd/csrc -j8 OUT=$OUT >/dev/null 2>$T/err || { tail -5 $T/err; exit 1; }
rm -rf $T; ls -la $OUT

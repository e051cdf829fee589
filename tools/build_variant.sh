#!/bin/bash
# builds a variant of liblsf_hip.so for A/B measurements: tools/build_variant.sh NAME [git-rev | -] [extra hipcc flags...]
#   git-rev: take csrc/ and include/ from that revision instead of the working tree ("-" = working tree)
# output: levelsetfusion-python_amd/lib/variants/NAME.so   (use with LSF_HIP_LIBRARY=...)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REV=${2:--}; shift; shift || true
OUT=$R/levelsetfusion-python_amd/lib/variants
mkdir -p $OUT
W=$(mktemp -d)
if [ "$REV" = "-" ]; then
  mkdir -p $W/levelsetfusion-python_amd $W/include
  cp -r $R/levelsetfusion-python_amd/csrc $W/levelsetfusion-python_amd/
  cp $R/include/*.h $W/include/
else
  (cd $R && git archive $REV levelsetfusion-python_amd/csrc include) | tar -x -C $W
fi
# lsf_build_id() of a variant: the hash of ITS sources and extra flags (bench.py reports the committed PMC traffic only for
# the build it was measured on; a variant must not inherit the shipped library's id)
ID=$( (cat $(ls $W/levelsetfusion-python_amd/csrc/* $W/include/*.h | sort); echo "$@") | sha256sum | cut -c1-16)
# lsf_abi_hash() of a variant: the hash of ITS header (the binding refuses a variant whose structs / prototypes differ)
ABI=$(python3 -c "import importlib.util as u,sys; s=u.spec_from_file_location('b','$R/levelsetfusion-python_amd/_build.py'); m=u.module_from_spec(s); s.loader.exec_module(m); print(m.abi_hash('$W/include/lsf_hip.h'))")
OBJS=""; PIDS=""
for f in $W/levelsetfusion-python_amd/csrc/*.hip; do
  o=$W/$(basename $f .hip).o
  hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function "-DLSF_BUILD_ID=\"v$ID\"" "-DLSF_ABI_HASH=\"$ABI\"" "$@" -c $f -o $o &
  PIDS="$PIDS $!"
  OBJS="$OBJS $o"
done
for p in $PIDS; do wait $p || { echo "compile failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OBJS -ldl
rm -rf $W
echo $OUT/$NAME.so

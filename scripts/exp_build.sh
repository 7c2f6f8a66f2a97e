#!/bin/bash
# Builds a variant of libmdisp.so for a kernel experiment: scripts/exp_build.sh NAME "-DFLAG ..." file.hip [file.hip ...]
# recompiles the named sources with the extra flags and links them with the objects of the regular build into
# cameracalibration_amd/lib/libmdisp_NAME.so (use with MDISP_LIB_PATH).
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/cameracalibration_amd/lib
mkdir -p $lib/obj_$name
objs=""
for o in $lib/obj/*.o; do
  b=$(basename $o .o)
  skip=0
  for f in "$@"; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $flags \
     -c $root/cameracalibration_amd/csrc/$f -o $lib/obj_$name/${f%.hip}.o &
done
wait
for f in "$@"; do objs="$objs $lib/obj_$name/${f%.hip}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/libmdisp_$name.so $objs
echo $lib/libmdisp_$name.so

#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# Build a variant of the library from one re-compiled source: bash tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# -> meld_amd/libmeld_hip_<name>.so (other objects are taken from meld_amd/build/ as they are)
set -e
name=$1; src=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/variant_${name}_$(basename $src .hip).o
file_flags=$(python3 -c "import sys; sys.path.insert(0, '$here'); from meld_amd import build as b; print(' '.join(b.FILE_FLAGS.get('$src', [])))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $file_flags "$@" -c $here/meld_amd/csrc/$src -o $obj
objs=""
for o in $here/meld_amd/build/*.o; do
  if [ "$(basename $o)" == "$(basename $src .hip).o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $here/meld_amd/libmeld_hip_$name.so
echo built meld_amd/libmeld_hip_$name.so

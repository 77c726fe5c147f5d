#!/bin/bash
# Build a variant of libvitmi_hip.so for same-box A/B timing: recompiles ONE source with extra flags and links it with
# the up-to-date objects of the regular build.  usage: tools/build_alt.sh <name> <source.hip> [hipcc flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python visiontransformer_amd/build.py > /dev/null
mkdir -p tools/altlib /tmp/altobj
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function "$@" -c visiontransformer_amd/csrc/$src -o /tmp/altobj/${base}_$name.o
objs=""
for o in visiontransformer_amd/csrc/build/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs /tmp/altobj/${base}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/altlib/libvitmi_hip_$name.so $objs
echo built tools/altlib/libvitmi_hip_$name.so

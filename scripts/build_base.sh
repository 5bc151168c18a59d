#!/bin/bash
# Build the library of another commit (default HEAD) as cat-generator_amd/lib/libcatgan_hip_base.so for same-box A/B runs:
#   CATGAN_LIB=cat-generator_amd/lib/libcatgan_hip_base.so python scripts/kbench.py ...
set -e
REV=${1:-HEAD}
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$ROOTD" archive "$REV" cat-generator_amd/csrc include | tar -x -C "$T"
OBJS=""
for f in "$T"/cat-generator_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -c "$f" -o "$f.o" &
  OBJS="$OBJS $f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$ROOTD/cat-generator_amd/lib/libcatgan_hip_base.so" -ldl
rm -rf "$T"
echo built "$ROOTD/cat-generator_amd/lib/libcatgan_hip_base.so" from $REV

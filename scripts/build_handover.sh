#!/bin/bash
# libraries with the cross-workgroup hand-over variants of the column reductions (common.h CG_COL_HANDOVER) for a same-box A/B
set -e
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOTD/cat-generator_amd/lib
python -c "import importlib; importlib.import_module('cat-generator_amd.build').build()" >/dev/null
for v in ${VARIANTS:-0 3 4}; do
  ( for f in ops fused; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DCG_COL_HANDOVER=$v -c "$ROOTD/cat-generator_amd/csrc/$f.hip" -o "$L/obj/${f}_ho$v.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$L/obj/gemm.o" "$L/obj/winograd.o" "$L/obj/ops_ho$v.o" "$L/obj/fused_ho$v.o" "$L/obj/comm.o" "$L/obj/locnet.o" "$L/obj/net.o" -o "$L/libcatgan_hip_ho$v.so" -ldl && echo built ho$v ) &
done
wait

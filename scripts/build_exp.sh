#!/bin/bash
# Ablation libraries for timing experiments: gemm.hip compiled with -DCG_EXP=<bits> (see gemm.hip; `trace` = -DCG_TRACE, the per-workgroup timestamps), linked with the regular
# objects into cat-generator_amd/lib/libcatgan_hip_exp<bits>.so.  Use: CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_exp1.so python scripts/kbench.py ...
set -e
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOTD/cat-generator_amd/lib
python -c "import importlib; importlib.import_module('cat-generator_amd.build').build()" >/dev/null
for e in "$@"; do
  D="-DCG_EXP=$e"; [ "$e" = trace ] && D="-DCG_TRACE"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include $D -c "$ROOTD/cat-generator_amd/csrc/gemm.hip" -o "$L/obj/gemm_exp$e.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$L/obj/gemm_exp$e.o" "$L/obj/winograd.o" "$L/obj/ops.o" "$L/obj/fused.o" "$L/obj/comm.o" "$L/obj/locnet.o" "$L/obj/net.o" -o "$L/libcatgan_hip_exp$e.so" -ldl && echo built exp$e ) &
done
wait

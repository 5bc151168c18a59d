#!/bin/bash
# round 5, call b: loop-structure lab for the TN (weight-gradient) GEMM
mkdir -p gpurun_out/r05b
for s in 8 4 2; do ./tools/gemm_lab $s > gpurun_out/r05b/lab2_split$s.txt 2>&1; cat gpurun_out/r05b/lab2_split$s.txt; done

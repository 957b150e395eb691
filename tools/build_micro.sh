#!/bin/bash
# Prebuild the GEMM micro-benchmark variants (tools/bin/ is git-ignored but travels to the GPU box with gpurun).
cd "$(dirname "$0")" && mkdir -p bin
for v in BASE NOGLOAD NOMMA NOEPI; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQAGNN_ABLATE_$v -o bin/gemm_ablate_$v gemm_ablate.hip; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/gather_micro gather_micro.hip

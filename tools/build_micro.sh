#!/bin/bash
# Prebuild the GEMM micro-benchmark variants (tools/bin/ is git-ignored but travels to the GPU box with gpurun).
cd "$(dirname "$0")" && mkdir -p bin
for v in BASE NOGLOAD NOMMA NOEPI; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQAGNN_ABLATE_$v -o bin/gemm_ablate_$v gemm_ablate.hip; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/gather_micro gather_micro.hip
# whole-library variants for kernel A/B runs (QAGNN_LIB=tools/bin/libqagnn_hip_u6.so python bench.py ...)
for u in 6 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEDGE_UNROLL=$u -o bin/libqagnn_hip_u$u.so ../qagnn_amd/csrc/graph_prep.hip ../qagnn_amd/csrc/gemm.hip ../qagnn_amd/csrc/elementwise.hip ../qagnn_amd/csrc/edge_attn.hip ../qagnn_amd/csrc/pool.hip ../qagnn_amd/csrc/hop.hip; done
# timing ablations of the NN split GEMM (operand split arithmetic removed for B / for A and B): upper bounds of what pre-split operands buy
SRC="../qagnn_amd/csrc/graph_prep.hip ../qagnn_amd/csrc/gemm.hip ../qagnn_amd/csrc/elementwise.hip ../qagnn_amd/csrc/edge_attn.hip ../qagnn_amd/csrc/pool.hip ../qagnn_amd/csrc/hop.hip ../qagnn_amd/csrc/optim.hip ../qagnn_amd/csrc/gemm_split.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQAGNN_ABL_NOBSPLIT -o bin/libqagnn_hip_nobsplit.so $SRC
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQAGNN_ABL_NOBSPLIT -DQAGNN_ABL_NOASPLIT -o bin/libqagnn_hip_nosplit.so $SRC

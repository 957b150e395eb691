#!/bin/bash
# Prebuild the GEMM micro-benchmark variants (tools/bin/ is git-ignored but travels to the GPU box with gpurun).
cd "$(dirname "$0")" && mkdir -p bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/gather_micro gather_micro.hip
# whole-library variants for kernel A/B runs (QAGNN_LIB=tools/bin/libqagnn_hip_u6.so python bench.py ...)
for u in 6 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEDGE_UNROLL=$u -o bin/libqagnn_hip_u$u.so ../qagnn_amd/csrc/graph_prep.hip ../qagnn_amd/csrc/gemm.hip ../qagnn_amd/csrc/elementwise.hip ../qagnn_amd/csrc/edge_attn.hip ../qagnn_amd/csrc/pool.hip ../qagnn_amd/csrc/hop.hip; done
# timing ablations of the NN split GEMM (operand split arithmetic removed for B / for A and B): upper bounds of what pre-split operands buy
SRC="../qagnn_amd/csrc/graph_prep.hip ../qagnn_amd/csrc/gemm.hip ../qagnn_amd/csrc/elementwise.hip ../qagnn_amd/csrc/edge_attn.hip ../qagnn_amd/csrc/pool.hip ../qagnn_amd/csrc/hop.hip ../qagnn_amd/csrc/optim.hip ../qagnn_amd/csrc/gemm_split.hip ../qagnn_amd/csrc/gemm_nn2.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQAGNN_ABL_NOBSPLIT -o bin/libqagnn_hip_nobsplit.so $SRC
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQAGNN_ABL_NOBSPLIT -DQAGNN_ABL_NOASPLIT -o bin/libqagnn_hip_nosplit.so $SRC
# round 4: timing ablations of the second-generation NN kernel and of the weight-gradient kernels (bits: see QAGNN_NN2_ABL / QAGNN_TNW_ABL in the
# kernel sources), the placement probes behind DESIGN 6f (which blocks share a CU, which waves share a SIMD)
INC="-I../include -I../qagnn_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 $INC -o bin/nn2_ablate_0 nn2_ablate.hip
for v in 1 2 4 7 32 64 71; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $INC -DQAGNN_NN2_ABL=$v -o bin/nn2_x_abl$v nn2_ablate.hip; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 $INC -o bin/tn_abl_0 tn_ablate.hip
for v in 1 2 3 4 12 15 16 28; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $INC -DQAGNN_TNW_ABL=$v -o bin/tn_abl_$v tn_ablate.hip; done
hipcc --offload-arch=gfx950 -O3 -o bin/cu_census cu_census.hip
hipcc --offload-arch=gfx950 -O3 -o bin/simd_probe simd_probe.hip
# round 5: the library with the graph preparation's zero fill as a hipMemsetAsync node (the faulty form: scripts/r5_memset_node_fault.sh)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQAGNN_PREP_MEMSET_NODE -I../include -o bin/libqagnn_hip_memset_node.so $SRC
# round 6: phase ablation of the three-MFMA / one-MFMA forms of k_gemm_nn2 (profiles/r6_run17_nn2_ablation.txt): bits as above
for v in 0 1 2 4 16 32 64 33 35 39 96 103; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $INC -DQAGNN_NN2_ABL=$v -o bin/nn2_r6_abl$v nn2_ablate.hip; done

#!/bin/bash
# builds the diagnostic harnesses (binaries are git-ignored; they travel to the GPU box with gpurun)
set -e
cd "$(dirname "$0")"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../vq-vae_amd/csrc -I ../../include"
EXTRA="../../vq-vae_amd/csrc/pack_cache.hip ../../vq-vae_amd/csrc/defer.hip ../../vq-vae_amd/csrc/tcn_hot_bwd3.hip ../../vq-vae_amd/csrc/tcn_hot_bwd4.hip"
$CC -o tcn_bwd_v2.bin tcn_bwd_stamps.hip $EXTRA
$CC -DTH_STAMPS -o tcn_bwd_v2_stamps.bin tcn_bwd_stamps.hip $EXTRA
$CC -o vq_stamps.bin vq_stamps.hip ../../vq-vae_amd/csrc/defer.hip
$CC -o c3_stamps.bin c3_stamps.hip ../../vq-vae_amd/csrc/conv3x3_wgrad.hip ../../vq-vae_amd/csrc/pack_cache.hip ../../vq-vae_amd/csrc/defer.hip
$CC -o tcn_bwd3.bin tcn_bwd3_stamps.hip ../../vq-vae_amd/csrc/pack_cache.hip ../../vq-vae_amd/csrc/defer.hip
$CC -DB3_STAMPS -o tcn_bwd3_stamps.bin tcn_bwd3_stamps.hip ../../vq-vae_amd/csrc/pack_cache.hip ../../vq-vae_amd/csrc/defer.hip
$CC -DB4_STAMPS -fno-slp-vectorize -o tcn_bwd4_stamps.bin tcn_bwd3_stamps.hip ../../vq-vae_amd/csrc/pack_cache.hip ../../vq-vae_amd/csrc/defer.hip

// md2_oracle.h — CPU restatement of the Monodepth2.jl self-supervised training step.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load libmd2oracle.so, and
// there only as the checker / reported CPU baseline.  The product path is the HIP library
// in monodepth2.jl_amd/csrc (libmd2hip.so), which never links or calls this code.
//
// PARITY PINNING STATUS:
//   pinned by the reference's own tests (test/runtests.jl):  so3_exp_map / hat / composeT,
//     SSIM (3 properties), smooth_loss (KAT 0.2542), disparity_to_depth range,
//     Backproject∘Project∘grid_sample(zeros) identity warp.
//   PARITY UNPINNED (reference is Julia + un-vendored, unpinned Flux/NNlib/Zygote/ResNet.jl;
//     no Julia toolchain in the build container, so the reference cannot be executed):
//     grid_sample(:border) on non-identity grids, upsample_bilinear, Conv/BatchNorm/MaxPool,
//     the ResNet-18 encoder, DepthDecoder, PoseDecoder, Model, photometric/prediction/
//     automasking/train_loss, every gradient except hat's, ADAM.  Those follow the documented
//     Flux/NNlib semantics and are cross-checked against an independent
//     torch-CPU implementation (tests/golden, "torch-derived, not Flux-derived").
//
// Layout: everything here is in the REFERENCE layout.  A Julia column-major array of size
// (W,H,C,N) is byte-identical to a row-major C array [N][C][H][W]; (W,H,C,L,N) == [N][L][C][H][W];
// grid (2,W,H,N) == [N][H][W][2]; rvec (3,N) == [N][3]; R (3,3,N) == [N][col][row] (column-major 3x3
// per batch item, i.e. R[n][j][i] = R_ij).
#pragma once
#include <cstdint>
#include <cstddef>

#define O_API extern "C" __attribute__((visibility("default")))

// bf16 NT GEMM for the fc layers at full batch: 256 x 256 output tile, 8 wavefronts (2 sample
// halves x 4 feature quarters, each wave 128 samples x 64 features = 4 x 2 MFMA 32x32 tiles,
// 128 accumulator registers), K-step 64, operands staged HBM -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), two LDS buffers.
//
// LDS image = the 128-byte-row tile of common.cuh (lds_tile_off): a global_load_lds wave
// instruction writes 1 KiB = 8 rows lane-linearly, so the XOR swizzle is applied to the per-lane
// SOURCE chunk (lane l of row group g fills physical chunk l&7 of row 8g + (l>>3) with logical chunk
// (l&7) ^ ((row>>1)&7)) and again on the ds_read side (guide rule 21: both sides or neither).
//
// Block -> tile map: the F/256 feature tiles of one 256-row sample tile run back to back on ONE
// XCD (blocks b and b+8 share an XCD under round-robin dispatch), so the second read of the A tile
// hits that XCD's L2 instead of HBM.  A speed choice only: correctness does not depend on placement.
//
// Epilogues as in gemm_nt.cuh (EPI_FWD: bias + ReLU + BN sums; EPI_DGRAD: dropout + BN-backward sums).
#pragma once
#include "common.cuh"
#include "gemm_nt.cuh"

typedef __attribute__((address_space(3))) void lds_void_t;

// One LDS-DMA wave instruction (1 KiB): LDS[lds_dst + 16*lane] = *(16 bytes at gsrc), issued from
// inline asm so that hipcc does NOT track it: with the builtin the compiler waits vmcnt(0) before
// the next ds_read of the same __shared__ array, which serialises the next tile's loads with the
// current tile's MFMAs.  M0 (the LDS base) is compiler-reserved, so it is saved, set and restored
// inside the one statement (guide 5.7).  Completion is ordered by the explicit vmcnt waits below.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}

#ifdef CP_VARIANTS   // the one-tile-per-block 256x256 NT kernel (superseded by gemm_nt256p.cuh / gemm_ws.cuh)
#include "../../tools/variants/gemm_nt256_onetile.cuh"
#endif

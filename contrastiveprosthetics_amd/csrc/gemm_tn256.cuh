// bf16 weight-gradient GEMM at full batch:  C[p][q] = sum_m X[m][p] * Y[m][q]   (X = g_y, Y = layer input)
// 256 x 256 output tile per block, 8 wavefronts (2 x 4, each 128(p) x 64(q) = 4 x 2 MFMA tiles, 128
// accumulator registers), the window axis m split over blocks (one f32 slab per split, summed by
// reduce_slabs_kernel).  Each step consumes 32 rows of both operands; rows are staged HBM -> LDS with
// global_load_lds_dwordx4 into a ring of 4 stages (3 steps of loads in flight, counted vmcnt) and the
// MFMA fragments (8 consecutive m for one column) are read with ds_read_b64_tr_b16.
//
// LDS image of one operand stage: [32 rows][256 columns] bf16 = 512-byte rows.  A transposed read
// touches 4 consecutive rows x 64 bytes per 32 lanes; with 512-byte rows those four segments would
// share their banks, so the 64-byte block index of a row is XORed with (row & 3).  The LDS-DMA writes
// lane-linearly, so the permutation is applied to the per-lane SOURCE column (and to the read address).
//
// Block -> (split, tile) map: all tiles of one split run on one XCD (blocks b and b+8 share an XCD
// under round-robin dispatch) so the operand rows are fetched from HBM once and re-read from that
// XCD's L2.  A speed choice only.
#pragma once
#include "common.cuh"
#include "gemm_tn.cuh"
#include "gemm_nt256.cuh"      // glds16
#include "gemm_ws.cuh"         // f32x4_t

#define TN256_STAGES 4

struct GemmTN256Args {
    const bf16_t* X;   // [M][ldx]
    const bf16_t* Y;   // [M][ldy]
    float* slabs;      // [splits][P][Q]
    int64_t M;
    int64_t rows_per_split;   // multiple of 32
    int ldx, ldy, P, Q, splits;
    // optional second problem of the same shape in the same launch (X2 != nullptr): two layers' weight gradients share
    // the GPU, each with half the splits -- half the slab bytes written here and re-read by reduce_slabs_kernel
    const bf16_t* X2;
    const bf16_t* Y2;
    float* slabs2;
};

__device__ __forceinline__ uint4 tn256_frag(const unsigned char* tile, int m_off, int col0, int lane) {
    // rows m_off + 8*(g>>1) + q (+4), columns col0 + 16*(g&1) + 4*pp .. +3   (see tn_frag_bf16)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int row = m_off + 8 * (g >> 1) + q;
    const int cb = (col0 + 16 * (g & 1) + 4 * pp) * 2;                  // logical byte offset in the row
    const unsigned char* p = tile + row * 512 + ((((cb >> 6) ^ (row & 3)) << 6) | (cb & 63));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 512));   // (row+4)&3 == row&3
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// the same for v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, kg = lane >> 4) holds rows m_off + 8*kg .. +7 of column col0 + i
__device__ __forceinline__ uint4 tn256_frag16(const unsigned char* tile, int m_off, int col0, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int row = m_off + 8 * g + q;
    const int cb = (col0 + 4 * pp) * 2;
    const unsigned char* p = tile + row * 512 + ((((cb >> 6) ^ (row & 3)) << 6) | (cb & 63));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 512));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// M16: v_mfma_f32_16x16x32_bf16 (one 32-row step = one k block) instead of 32x32x16 -- measured, not faster (see the launcher)
template <bool M16>
__global__ __launch_bounds__(512) void gemm_tn256_kernel(GemmTN256Args a) {
    constexpr int OP_BYTES = 32 * 512;                 // one operand stage: 32 rows x 256 cols bf16
    constexpr int STAGE = 2 * OP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TN256_STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wq = wave & 3;
    const int tiles_q = a.Q / 256, ntiles = (a.P / 256) * tiles_q;
    // bid = xcd + 8 * (tile + ntiles * (split / 8)),  split = 8 * (..) + xcd
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int per_problem = ntiles * ((a.splits + 7) / 8);
    const bool second = j >= per_problem;              // block-uniform
    if (second) j -= per_problem;
    const bf16_t* __restrict__ Xg = second ? a.X2 : a.X;
    const bf16_t* __restrict__ Yg = second ? a.Y2 : a.Y;
    const int tile = j % ntiles;
    const int split = (j / ntiles) * 8 + xcd;
    if (split >= a.splits) return;
    const int p0 = (tile / tiles_q) * 256, q0 = (tile % tiles_q) * 256;
    const int64_t mb = (int64_t)split * a.rows_per_split;
    int64_t me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int nsteps = (int)((me - mb + 31) / 32);

    // staging: a stage holds 32 rows x 512 B per operand = 16 LDS-DMA instructions per operand; each of
    // the 8 waves issues 2 per operand.  Instruction i covers tile rows 2i, 2i+1.
    const int srow_l = lane >> 5;                      // row within the instruction's pair
    const int pb = (lane & 31) >> 2, sub = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage = [&](int slot, int step) {
        const int64_t ms = mb + (int64_t)step * 32;
        const uint32_t Xs = lds0 + slot * STAGE, Ys = Xs + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int inst = wave * 2 + i;             // 0..15
            const int row = inst * 2 + srow_l;
            int64_t m = ms + row;
            if (m >= me) m = me - 1;                   // rows past the end are zeroed by the caller's contract below
            const int colb = (((pb ^ (row & 3)) << 6) | (sub << 4)) >> 1;   // logical column (elements)
            glds16(Xg + m * a.ldx + p0 + colb, Xs + inst * 1024);
            glds16(Yg + m * a.ldy + q0 + colb, Ys + inst * 1024);
        }
    };

    float* slab = (second ? a.slabs2 : a.slabs) + (int64_t)split * a.P * a.Q;
    auto wait_and_stage = [&](int step) {
        // each wave has issued 4 LDS-DMA instructions per step; steps step+1, step+2 may stay in flight
        const int ahead = (nsteps - 1 - step) < (TN256_STAGES - 2) ? (nsteps - 1 - step) : (TN256_STAGES - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                // stage `step` landed for every wave; slot (step-1)%S is free
        if (step + TN256_STAGES - 1 < nsteps) stage((step + TN256_STAGES - 1) % TN256_STAGES, step + TN256_STAGES - 1);
    };
    // prologue: up to STAGES-1 steps in flight
#pragma unroll
    for (int s = 0; s < TN256_STAGES - 1; ++s)
        if (s < nsteps) stage(s, s);

    if constexpr (M16) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int step = 0; step < nsteps; ++step) {
            wait_and_stage(step);
            const unsigned char* Xs = smem + (step % TN256_STAGES) * STAGE;
            const unsigned char* Ys = Xs + OP_BYTES;
            const int valid = (int)(me - (mb + (int64_t)step * 32));      // >= 32 except possibly in the last step
            uint4 fx[8], fy[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) fx[i] = tn256_frag16(Xs, 0, wp * 128 + i * 16, lane);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fy[jj] = tn256_frag16(Ys, 0, wq * 64 + jj * 16, lane);
            if (valid < 32) {
                // fragment element e of lane group kg is row 8*kg + e: zero the X side of dead rows
                const int base = 8 * (lane >> 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint32_t* w = (uint32_t*)&fx[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t lo = (base + 2 * e) < valid ? 0xFFFFu : 0u;
                        const uint32_t hi = (base + 2 * e + 1) < valid ? 0xFFFF0000u : 0u;
                        w[e] &= (lo | hi);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, fx[i]), __builtin_bit_cast(s16x8, fy[jj]), acc[i][jj], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        const int cj = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int q = q0 + wq * 64 + jj * 16 + cj;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int p = p0 + wp * 128 + i * 16 + 4 * rg + e;
                    slab[(int64_t)p * a.Q + q] = acc[i][jj][e];
                }
            }
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][jj][g] = 0.f;
        for (int step = 0; step < nsteps; ++step) {
            wait_and_stage(step);
            const unsigned char* Xs = smem + (step % TN256_STAGES) * STAGE;
            const unsigned char* Ys = Xs + OP_BYTES;
            // rows >= me of the last step were loaded from a clamped (valid) row: mask them out of the sum
            const int valid = (int)(me - (mb + (int64_t)step * 32));      // >= 32 except possibly in the last step
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 fx[4], fy[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) fx[i] = tn256_frag(Xs, ks * 16, wp * 128 + i * 32, lane);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) fy[jj] = tn256_frag(Ys, ks * 16, wq * 64 + jj * 32, lane);
                if (valid < 32) {
                    // fragment element e of lane half h is row ks*16 + 8h + e: zero the X side of dead rows
                    const int base = ks * 16 + 8 * (lane >> 5);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint32_t* w = (uint32_t*)&fx[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t lo = (base + 2 * e) < valid ? 0xFFFFu : 0u;
                            const uint32_t hi = (base + 2 * e + 1) < valid ? 0xFFFF0000u : 0u;
                            w[e] &= (lo | hi);
                        }
                    }
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) mma_chunk<bf16_t>(fx[i], fy[jj], acc[i][jj]);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int q = q0 + wq * 64 + jj * 32 + r;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int p = p0 + wp * 128 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                    slab[(int64_t)p * a.Q + q] = acc[i][jj][g];
                }
            }
    }
}

#ifdef CP_VARIANTS   // weight-gradient variants measured and not faster (16x16x32 form, one wave per SIMD)
#include "../../tools/variants/gemm_tn256_variants.cuh"
#endif

static inline hipError_t launch_gemm_tn256(const GemmTN256Args& a, hipStream_t st) {
    const int ntiles = (a.P / 256) * (a.Q / 256);
    const int groups = (a.splits + 7) / 8;
    const int problems = a.X2 ? 2 : 1;
    // ($CPNATIVE_TN16: the 16x16x32 form.  Measured in the step, alternating runs on one box: 137.0 / 150.6 us per launch against
    //  134.5 / 147.2 for the 32x32x16 form -- the shape that gained 18 % in the weight-stationary forward gains nothing here, where
    //  both operands come through ds_read_b64_tr_b16 at 0.75 fragment reads per MFMA-equivalent either way.)
#ifdef CP_VARIANTS
    if (g_var.tn_w4) { hipLaunchKernelGGL(gemm_tn256w4_kernel, dim3((unsigned)(problems * groups * 8 * ntiles)), dim3(256), 0, st, a); return hipGetLastError(); }
    if (g_var.tn16) { hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3((unsigned)(problems * groups * 8 * ntiles)), dim3(512), 0, st, a); return hipGetLastError(); }
#endif
    hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3((unsigned)(problems * groups * 8 * ntiles)), dim3(512), 0, st, a);
    return hipGetLastError();
}

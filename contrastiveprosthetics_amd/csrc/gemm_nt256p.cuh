// Persistent form of the bf16 256 x 256 NT GEMM (gemm_nt256.cuh) for the fc forward pass and for
// data gradients that carry no statistics: ONE 8-wave block per CU for the whole launch, each block
// walking its own list of output tiles.
//
// Why: with 64 KiB LDS stages only one block fits a CU, and a wave cannot retire before its stores
// are acknowledged, so in the one-tile-per-block kernel every tile pays, back to back and with
// nothing else on the CU, (a) the fill of its first stage, (b) its K loop, (c) the drain of its
// 128 KiB of C stores.  Here the K loop is flattened across the block's tiles: the first stage of
// tile n+1 is requested during the last K step of tile n, and tile n's stores -- written straight
// from the accumulator registers, no LDS staging, no barrier (see the register-direct epilogue in
// gemm_nt256.cuh) -- drain underneath tile n+1's first K step.
//
// Tile schedule, XCD-aware, two forms behind one template flag (cp_set_tile_schedule picks per process).
// Dynamic: block b runs on XCD b & 7 and draws that XCD's tiles -- sample tile
// (item / tiles_f) * 8 + xcd, column tile item % tiles_f -- from a per-XCD counter, one tile ahead of the one it
// is computing.  Neighbouring items share an A tile and are drawn by different blocks at about the same time, so
// its second read is an L2 hit.  The bias / BatchNorm coefficients of every column tile sit in LDS; the
// BatchNorm column sums leave as ONE partial row per SAMPLE TILE (not per block), so the sums downstream add
// up in the same order whichever block ran which tile: results do not depend on the schedule.
// Static: block j of an XCD has a fixed column tile and every (J/tiles_f)-th sample tile, carries the sums in
// registers and writes one partial row per block.  Alone on the GPU that is 2-3 % faster (147 / 182 / 174 us for the
// three fc launches against 150 / 187 / 178, tools/ab_sched.sh), but with 8-32 CUs held by another stream's kernel
// -- RCCL at N > 1, a neighbour process in a packed hyper-parameter sweep -- the blocks that start late hold the
// whole launch back: 201-210 us against 150, where the dynamic form stays at 148-149 (tools/contention_bench.py).
// Measured on the dynamic form and dropped: flushing a tile's sums from the first K step of the next tile instead
// of behind a barrier of their own (+5 us forward: the K loop does not tolerate extra code), drawing two tiles ahead
// with an untracked asm atomic plus per-half partial rows written straight from registers (+4 us forward).
// Also measured on top of the dynamic schedule: handing out the leftover round (164 tiles on an XCD's 32 CUs = 5
// rounds + 4 tiles) in quarter tiles of 64 rows.  A quarter's K step has 8 MFMAs per wave to cover the same
// stage latency, so it took well over half a tile's time: forward 154 us instead of 148.
//
// Measured and dropped on top of this kernel: applying the layer below's BatchNorm + ReLU backward in the
// data-gradient epilogue (coefficients are known beforehand thanks to bn_bwd_sums_from_wgrad_kernel; the
// saved activation was fetched in the store layout and un-swapped with the same permlane32 swap).  With
// 128 accumulator registers live the fetched tile and the coefficients did not fit: 53-63 spilled VGPRs,
// +160..220 us per launch against the 94 us of the separate pass it replaced (which already streams at
// 5.5 TB/s).
//
// Also measured (round 2): contiguous row ranges per block pair instead of whole tiles dealt round-robin -- every CU then
// gets 1,312 rows = 5 tiles + a 32-row ragged tile whose dead waves skip their MFMAs, instead of 5.125 rounds "paid as 6".
// A/B in alternating bench runs on one box: forward 154.5 vs 148.0 us, BN-mode data gradient 178.5 vs 181.3 us, step 4.50
// vs 4.46 ms (median).  The sixth round was never a full round: its 32 tiles run on an otherwise idle chip (no contention for
// L2, HBM and the power budget) in ~0.6 of a tile time, about what 256 ragged tiles with their full W fills cost.
//
// Also measured: F = 768 (fc1's data gradient: 3 column tiles, so the static form runs 30 blocks per XCD and 9 rounds)
// through the per-tile code path with round-robin items on all 32 blocks (8 rounds): that launch 19 us faster (-8 %),
// the step unchanged -- its 656 x 12 partial rows of 64 channels cost the reduction kernels what the launch gained.
//
// Also measured: starting the XCDs 1.6 / 3.2 us apart, or the blocks of an XCD 6.4 us x (j & 3) apart, so that the
// CUs' 128 KiB store bursts do not reach HBM together: no consistent change -- in the same process the same launch
// moved between 133 and 155 us from one batch of 30 launches to the next (clock state), more than any of these.
//
// Also measured: staging through registers (global_load_dwordx4 at the start of a K step, ds_write_b128 at its end)
// to avoid the LDS-DMA issue cost (8 issues per wave per K step, 100-185 cycles each beside ds_reads and MFMAs).  With
// 128 accumulator + 48 fragment registers live there is no room for the 32 staging registers: hipcc parks them in
// scratch right after the loads (a vmcnt wait + scratch_store per load), 338 us against 131 us.  Once the column sums
// stopped being carried through the K loop there was room for HALF a stage (the A rows, 16 registers as four scalar
// uint4 -- an array of them still went to scratch): 4 LDS-DMA requests per wave and K step instead of 8, no scratch,
// same speed (151.8 vs 149.2 us forward, 130.1 vs 128.1 us data gradient).  So the DMA issue cost is not what holds the
// K loop back.  What the numbers say instead: a K step moves 64 KiB into LDS and 196 KiB out of it (24 fragment reads x
// 8 waves); at the guide's LDS rates that is ~1,800 of the 2,048 cycles the step's MFMAs take on each SIMD pair -- the
// loop is co-limited by LDS bandwidth and the matrix pipe, which leaves nothing to hide latencies behind.  Fewer LDS
// bytes per MFMA needs 128 x 128 wave tiles (256 accumulator registers, one wave per SIMD).  That kernel was written
// for the plain data gradient (4 waves per block, accumulators in AGPRs, 141 VGPRs, no spills, bit-identical output) and
// measured 145 us against 127-132 us here, with the stage requests either spread behind the MFMAs or issued up front:
// with a lone wave per SIMD nothing covers its LDS and barrier latencies, and hipcc's schedule does not either.
#pragma once
#include <mutex>
#include "gemm_nt256.cuh"

#define NT256P_MAX_TILES_F 3      // F <= 768: the encoder's widest data gradient
#define NT256P_SCHED_SLOTS 64     // streams per process that may run this kernel

// quad-permuted copy of w (DPP, no LDS crossbar)
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float w) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w), CTRL, 0xF, 0xF, true));
}

// one 16-byte chunk of C.  (Non-temporal stores were measured: 203 vs 143 us forward, 195 vs 131 us data gradient --
// the four chunks of a 128-byte line no longer merge in L2.)
__device__ __forceinline__ void store_c16(bf16_t* p, const uint4& c) { *(uint4*)p = c; }
// (cvt_pk_bf16<RELU>: common.cuh)

// Register-direct epilogue of one tile (see gemm_nt256.cuh for the lane algebra): converts the accumulators
// into 16-byte chunks (features i*32 + 8*(2kk + h) .. +7 of row mw0 + jj*32 + r; base = &C[mw0 + r][fw0 + 8h]),
// stores them and folds the BatchNorm sums.  FULL = every row of the tile exists (only the launch's very
// last sample tile can be ragged).  EPI_FWD: the accumulators already hold the bias (they were initialised
// with it); ReLU is applied to the packed bf16 pairs; the sums are taken over the values as stored and
// folded, per tile, from 32 to 8 values per statistic with two DPP butterfly steps inside each quad of lanes.
template <int EPI, int MT, bool FULL>
__device__ __forceinline__ void nt256p_convert(const GemmNTArgs& a, f32x16 (&acc)[2][MT], int64_t mw0,
                                               bf16_t* base, int r, int lane, float (&qs1)[8], float (&qs2)[8]) {
    float ps1[2][16], ps2[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) ps1[i][v] = ps2[i][v] = 0.f;
    // sample tile outermost: the four 16-byte stores that make up one 128-byte line of a row (i = 0,1 x
    // kk = 0,1) leave back to back, so L2 merges them into one full-line write (with the feature half
    // outermost the PMC pass showed 210 MB written per launch for 172 MB of output)
#pragma unroll
    for (int jj = 0; jj < MT; ++jj) {
        const bool live = FULL || (mw0 + jj * 32 + r) < a.M;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint2 pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pk[q].x = cvt_pk_bf16<EPI == EPI_FWD>(acc[i][jj][4 * q], acc[i][jj][4 * q + 1]);
                pk[q].y = cvt_pk_bf16<EPI == EPI_FWD>(acc[i][jj][4 * q + 2], acc[i][jj][4 * q + 3]);
                if constexpr (EPI == EPI_FWD) {
                    float g0 = __uint_as_float(pk[q].x << 16), g1 = __uint_as_float(pk[q].x & 0xffff0000u);
                    float g2 = __uint_as_float(pk[q].y << 16), g3 = __uint_as_float(pk[q].y & 0xffff0000u);
                    if (!live) g0 = g1 = g2 = g3 = 0.f;
                    const int o = 4 * q;
                    ps1[i][o] += g0; ps1[i][o + 1] += g1; ps1[i][o + 2] += g2; ps1[i][o + 3] += g3;
                    ps2[i][o] = fmaf(g0, g0, ps2[i][o]); ps2[i][o + 1] = fmaf(g1, g1, ps2[i][o + 1]);
                    ps2[i][o + 2] = fmaf(g2, g2, ps2[i][o + 2]); ps2[i][o + 3] = fmaf(g3, g3, ps2[i][o + 3]);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * kk].x, pk[2 * kk + 1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * kk].y, pk[2 * kk + 1].y, false, false);
                // lanes 0..31: features 16kk..16kk+7 of row r; lanes 32..63: features 16kk+8..16kk+15
                const uint4 c = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                if (live) store_c16(base + (int64_t)(jj * 32) * a.ldc + i * 32 + 16 * kk, c);
            }
        }
    }
    if constexpr (EPI == EPI_FWD) {
        // values v = i*16 + 4q + e: two butterfly steps inside each quad of lanes, 16 -> 4 per statistic and half
        const bool o0 = lane & 1, o1 = (lane >> 1) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float k1 = o0 ? ps1[i][2 * p + 1] : ps1[i][2 * p], g1 = o0 ? ps1[i][2 * p] : ps1[i][2 * p + 1];
                const float k2 = o0 ? ps2[i][2 * p + 1] : ps2[i][2 * p], g2 = o0 ? ps2[i][2 * p] : ps2[i][2 * p + 1];
                ps1[i][p] = k1 + dpp_quad<0xB1>(g1);        // quad_perm [1,0,3,2]
                ps2[i][p] = k2 + dpp_quad<0xB1>(g2);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float k1 = o1 ? ps1[i][2 * p + 1] : ps1[i][2 * p], g1 = o1 ? ps1[i][2 * p] : ps1[i][2 * p + 1];
                const float k2 = o1 ? ps2[i][2 * p + 1] : ps2[i][2 * p], g2 = o1 ? ps2[i][2 * p] : ps2[i][2 * p + 1];
                qs1[4 * i + p] += k1 + dpp_quad<0x4E>(g1);       // quad_perm [2,3,0,1]
                qs2[4 * i + p] += k2 + dpp_quad<0x4E>(g2);
            }
        }
    }
}

// sum over the 4 lanes of a quad of four per-lane values v0..v3 (two DPP butterfly steps): lane (b1 b0) of the quad ends
// with the quad's total of v[2*b1 + b0] -- the first two steps of the 32-lane reduction of the BatchNorm column sums
__device__ __forceinline__ float quad_fold(float v0, float v1, float v2, float v3, bool o0, bool o1) {
    const float a = (o0 ? v1 : v0) + dpp_quad<0xB1>(o0 ? v0 : v1);       // quad_perm [1,0,3,2]
    const float b = (o0 ? v3 : v2) + dpp_quad<0xB1>(o0 ? v2 : v3);
    return (o1 ? b : a) + dpp_quad<0x4E>(o1 ? a : b);                    // quad_perm [2,3,0,1]
}

// ---- epilogues that work against the saved activation R (EPI_DGRAD_BN, EPI_DGRAD_ST) -------------------------------
// The R tile (256 x 256 bf16 = 128 KiB) comes through the ring buffer that the finished K loop has just released
// (64 KiB; the other one already holds the next tile's first stage), a QUARTER at a time: sample tile jj of both sample
// halves x all 256 features = 64 rows x 512 bytes = 32 KiB, two quarters in flight (LDS-DMA, no registers), so a quarter's
// fetch hides behind the previous quarter's arithmetic and stores.  A quarter holds whole output rows, so a wave still
// writes the four 16-byte chunks of a row's 128-byte line back to back (L2 merges them only then: splitting a line
// over two quarters showed 8-25 % more bytes written in the PMC pass).
// Quarter image: local row lr = ws*32 + r, 512 bytes per row = 32 granules of 8 features, granule g = wf*8 + i*4 + q
// stored at physical granule g ^ (lr & 31) (the swizzle is applied to the DMA's per-lane SOURCE; a ds_read_b64 of a
// wave then meets each bank pair twice instead of 32 times).
template <int MT>
__device__ __forceinline__ void nt256p_issue_quarter(const GemmNTArgs& a, int64_t m0, int f0, int jj, uint32_t q_lds,
                                                     int wave_u, int lane) {
    const bf16_t* Rg = (const bf16_t*)a.R;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = wave_u * 4 + t;                                 // DMA instruction 0..31 = local rows 2k, 2k+1
        const int lr = 2 * k + (lane >> 5);
        const int g = (lane & 31) ^ (lr & 31);
        int64_t m = m0 + (lr >> 5) * (32 * MT) + jj * 32 + (lr & 31);
        if (m >= a.M) m = a.M - 1;
        glds16(Rg + m * a.ldr + f0 + g * 8, q_lds + k * 1024);
    }
}

// arithmetic + stores of one quarter (sample tile JJ of this wave, both feature halves); JJ compile-time (it indexes the
// accumulator registers).  The statistics of each feature half are folded into qs right away (one 16-value scratch set).
template <int EPI, int MT, int JJ>
__device__ __forceinline__ void nt256p_quarter(const GemmNTArgs& a, f32x16 (&acc)[2][MT], const unsigned char* Rq, const float* coef_s,
                                               int64_t mw0, bf16_t* base, int fw0, int ws, int wf, int r, int h, int lane, uint32_t key,
                                               float (&qs1)[8], float (&qs2)[8]) {
    constexpr int jj = JJ;
    const int64_t m = mw0 + jj * 32 + r;
    const bool live = m < a.M;
    const int lr = ws * 32 + r;
    const bool o0 = lane & 1, o1 = (lane >> 1) & 1;
    uint4 out[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint2 rr = *(const uint2*)(Rq + lr * 512 + (((wf * 8 + i * 4 + q) ^ (lr & 31)) << 4) + 8 * h);
            const float r0 = __uint_as_float(rr.x << 16), r1 = __uint_as_float(rr.x & 0xffff0000u);
            const float r2 = __uint_as_float(rr.y << 16), r3 = __uint_as_float(rr.y & 0xffff0000u);
            float y0 = acc[i][jj][4 * q], y1 = acc[i][jj][4 * q + 1], y2 = acc[i][jj][4 * q + 2], y3 = acc[i][jj][4 * q + 3];
            if constexpr (EPI == EPI_DGRAD_BN) {
                const int fl = wf * 64 + i * 32 + 8 * q + 4 * h;
                const float4 ca = *(const float4*)(coef_s + fl), cb = *(const float4*)(coef_s + 256 + fl), cz = *(const float4*)(coef_s + 512 + fl);
                y0 = r0 > 0.f ? fmaf(ca.x, y0, fmaf(cb.x, r0, cz.x)) : 0.f;
                y1 = r1 > 0.f ? fmaf(ca.y, y1, fmaf(cb.y, r1, cz.y)) : 0.f;
                y2 = r2 > 0.f ? fmaf(ca.z, y2, fmaf(cb.z, r2, cz.z)) : 0.f;
                y3 = r3 > 0.f ? fmaf(ca.w, y3, fmaf(cb.w, r3, cz.w)) : 0.f;
            } else if (a.dp_thresh != 0) {
                const uint32_t col = (uint32_t)(fw0 + i * 32 + 8 * q + 4 * h);      // column of y0 in the output row (even)
                const uint32_t p0 = dropout_pair(key, (uint32_t)m, (uint32_t)a.ldc, col);
                const uint32_t p1 = dropout_pair(key, (uint32_t)m, (uint32_t)a.ldc, col + 2);
                y0 *= dropout_scale(p0, 0, a.dp_thresh, a.dp_inv_keep);
                y1 *= dropout_scale(p0, 1, a.dp_thresh, a.dp_inv_keep);
                y2 *= dropout_scale(p1, 0, a.dp_thresh, a.dp_inv_keep);
                y3 *= dropout_scale(p1, 1, a.dp_thresh, a.dp_inv_keep);
            }
            pk[q].x = cvt_pk_bf16<false>(y0, y1);
            pk[q].y = cvt_pk_bf16<false>(y2, y3);
            // sums of the values as stored; rows past the end of a ragged tile do not count
            float g0 = __uint_as_float(pk[q].x << 16), g1 = __uint_as_float(pk[q].x & 0xffff0000u);
            float g2 = __uint_as_float(pk[q].y << 16), g3 = __uint_as_float(pk[q].y & 0xffff0000u);
            if (!live) g0 = g1 = g2 = g3 = 0.f;
            qs1[4 * i + q] += quad_fold(g0, g1, g2, g3, o0, o1);
            if constexpr (EPI == EPI_DGRAD_ST) qs2[4 * i + q] += quad_fold(g0 * r0, g1 * r1, g2 * r2, g3 * r3, o0, o1);
            // keep the next quad's LDS reads (saved activation, coefficients) from being hoisted up here: with 128
            // accumulator registers live, eight quads' worth of operands in flight is what made hipcc spill
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * kk].x, pk[2 * kk + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * kk].y, pk[2 * kk + 1].y, false, false);
            out[i][kk] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
        if constexpr (EPI == EPI_DGRAD_ST) {
            // (this mode also carries the second statistic and the dropout hashes: holding both halves' chunks spills 15
            //  registers and measured slower; its two half-lines leave a few hundred cycles apart, from the same wave:
            //  180 MB written per 172 MB of output)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (live) store_c16(base + (int64_t)(jj * 32) * a.ldc + i * 32 + 16 * kk, out[i][kk]);
        }
    }
    if constexpr (EPI != EPI_DGRAD_ST) {
        // the row's four chunks leave together
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (live) store_c16(base + (int64_t)(jj * 32) * a.ldc + i * 32 + 16 * kk, out[i][kk]);
    }
}

// MT = 32-row sample tiles per wave: the block's tile is (64 * MT) x 256.  Measured at 167,936 x 512 x 512:
// MT = 3 (7 rounds of 192 rows instead of 6 of 256: 12.5 % fewer padded rows) ran no faster, its 30 % more
// weight-tile refills cost what the rounding saved; holding a converted 192-row tile in 48 registers to
// release its stores two per K step of the next tile made hipcc spill (118-280 VGPRs) and ran 1.4-1.8x
// slower.  The launcher therefore uses MT = 4.
// DYN: tiles drawn from the per-XCD counters (above); !DYN: the static assignment, block j of an XCD takes items
// j, j + J, j + 2J, ... (J = blocks per XCD, a multiple of tiles_f, so its column tile is fixed), carries the column
// sums in registers across its tiles and writes one partial row.
template <int EPI, int MT, bool DYN>
__global__ __launch_bounds__(512) void gemm_nt256p_kernel(GemmNTArgs a) {
    using T = bf16_t;
    constexpr int BM = 64 * MT, BN = 256, BK = 64, EPC = 8;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + W_BYTES;
    constexpr int MAX_TF = NT256P_MAX_TILES_F;
    constexpr bool RMODE = (EPI == EPI_DGRAD_BN || EPI == EPI_DGRAD_ST);
    constexpr bool STATS = (EPI == EPI_FWD || RMODE);
    // ring + [which][ws][BN] sums of one tile + per-column-tile bias / [3][BN] coefficients
    constexpr int LDS_BYTES = 2 * STAGE + 4 * BN * 4 + MAX_TF * 3 * BN * 4;
    static_assert(!RMODE || MT == 4, "the R epilogues are written for 256-row tiles");
    static_assert(LDS_BYTES <= 160 * 1024 - 64, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    __shared__ int s_item[2];
    float* red = (float*)(smem + 2 * STAGE);             // [which][ws][BN]
    float* bias_all = red + 4 * BN;                      // [tf][BN] or [tf][3][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_f = a.F / BN;
    const int64_t tiles_m = (a.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7;
    // work items of this XCD, handed out by its counter: item -> sample tile (item / tiles_f) * 8 + xcd, column tile
    // item % tiles_f, so the column tiles of one sample tile run at about the same time behind the same L2
    const int items = xcd < tiles_m ? (int)((tiles_m - xcd + 7) / 8) * tiles_f : 0;
    int* ctr = a.sched + xcd * 32;                       // a 128-byte line per counter
    const int J = gridDim.x >> 3;
    const int ws = wave >> 2, wf = wave & 3;

    const T* __restrict__ Ag = (const T*)a.A;
    const T* __restrict__ Wg = (const T*)a.W;
    const int lrow = lane >> 3, pch = lane & 7;
    const T* asrc[MT];
    const T* wsrc[4];
    const int tf_static = (int)(blockIdx.x >> 3) % tiles_f;     // !DYN: the block's column tile
    auto set_wsrc = [&](int tf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave + 8 * i) * 8 + lrow;
            const int lch = pch ^ ((row >> 1) & 7);
            wsrc[i] = Wg + (int64_t)(tf * BN + row) * a.K + lch * EPC;
        }
    };
    auto set_src = [&](int item) {
        const int64_t tm = (int64_t)(item / tiles_f) * 8 + xcd;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = (wave + 8 * i) * 8 + lrow;
            const int lch = pch ^ ((row >> 1) & 7);
            int64_t m = tm * BM + row;
            if (m >= a.M) m = a.M - 1;                               // clamp: such rows are never stored
            asrc[i] = Ag + m * a.lda + lch * EPC;
        }
        if constexpr (DYN) set_wsrc(item % tiles_f);
    };
    if constexpr (!DYN) set_wsrc(tf_static);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage = [&](int buf, int kt) {
        const uint32_t As = lds0 + buf * STAGE;
        const uint32_t Ws = As + A_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = wave_u + 8 * i;
            if (i < MT) glds16(asrc[i] + k0, As + rg * 1024);
            glds16(wsrc[i] + k0, Ws + rg * 1024);
        }
    };

    const int nk = a.K / BK;
    if constexpr (DYN)
        if (tid == 0) s_item[0] = atomicAdd(ctr, 1);
    if constexpr (EPI == EPI_FWD)
        for (int q = tid; q < a.F; q += 512) bias_all[q] = a.bias[q];
    if constexpr (EPI == EPI_DGRAD_BN)
        for (int q = tid; q < tiles_f * 3 * BN; q += 512) {
            const int tf = q / (3 * BN), c = (q / BN) % 3, f = tf * BN + q % BN;
            bias_all[q] = a.coef[c * a.coef_mod + f % a.coef_mod];
        }
    const uint32_t dkey = (EPI == EPI_DGRAD_ST && a.dp_thresh != 0) ? (a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key) : 0u;
    __syncthreads();
    // the block's second item is drawn only now, behind every other block's first: neighbouring items -- the column
    // tiles of one sample tile -- go to different CUs at the same time and share the tile's rows in L2
    int cur = DYN ? __builtin_amdgcn_readfirstlane(s_item[0]) : (int)(blockIdx.x >> 3);   // block-uniform: SGPRs
    int slot = 0;                                        // s_item[slot] takes the pull issued at the start of the current tile
    T* Cg = (T*)a.C;
    int buf = 0;
    if (cur < items) {
        int pulled = 0;
        if constexpr (DYN)
            if (tid == 0) pulled = atomicAdd(ctr, 1);
        set_src(cur);
        stage(0, 0);
        if constexpr (DYN)
            if (tid == 0) s_item[1] = pulled;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int nxt = !DYN ? cur + J : cur < items ? __builtin_amdgcn_readfirstlane(s_item[1]) : 0;
    float tot1 = 0.f, tot2 = 0.f;                        // !DYN: column sums over the block's tiles (see the reduction below)

    while (cur < items) {
        const int64_t m0 = ((int64_t)(cur / tiles_f) * 8 + xcd) * BM;
        const int tf = DYN ? cur % tiles_f : tf_static;
        const int f0 = tf * BN;
        const bool has_next = nxt < items;
        // the item after the next one: requested a whole tile ahead, parked in LDS behind the first K step (whose closing
        // wait and barrier it shares: waiting for the counter's round trip here cost 10 us per launch), read at the end
        // of this tile; the slots alternate so the next tile's pull cannot overtake that read
        int pulled = 0;
        if constexpr (DYN)
            if (tid == 0 && has_next) pulled = atomicAdd(ctr, 1);
        const float* bias_s = bias_all + tf * (EPI == EPI_DGRAD_BN ? 3 * BN : BN);
        f32x16 acc[2][MT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 b0;
            if constexpr (EPI == EPI_FWD) {
                // accumulators start at the bias: register g of a 32x32 tile is feature 8*(g>>2) + 4h + (g&3)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *(const float4*)(bias_s + wf * 64 + i * 32 + 8 * q + 4 * h);
                    b0[4 * q] = b4.x; b0[4 * q + 1] = b4.y; b0[4 * q + 2] = b4.z; b0[4 * q + 3] = b4.w;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) b0[g] = 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < MT; ++jj) acc[i][jj] = b0;
        }

        for (int kt = 0; kt < nk; ++kt) {
            const unsigned char* As = smem + buf * STAGE;
            const unsigned char* Ws = As + A_BYTES;
            uint4 fw[2][2], fs[2][MT];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[0][i] = *(const uint4*)(Ws + lds_tile_off(wf * 64 + i * 32 + r, h));
#pragma unroll
            for (int jj = 0; jj < MT; ++jj) fs[0][jj] = *(const uint4*)(As + lds_tile_off(ws * (BM / 2) + jj * 32 + r, h));
            // the next stage's DMA requests go out while the first fragments are on their way from LDS (issued before those
            // reads: +1 % step time; behind the first MFMA group: +3 %)
            asm volatile("" ::: "memory");
            if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
            else if (has_next) { set_src(nxt); stage(buf ^ 1, 0); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur_f = ks & 1, nxt_f = cur_f ^ 1;
                if (ks + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        fw[nxt_f][i] = *(const uint4*)(Ws + lds_tile_off(wf * 64 + i * 32 + r, 2 * (ks + 1) + h));
#pragma unroll
                    for (int jj = 0; jj < MT; ++jj)
                        fs[nxt_f][jj] = *(const uint4*)(As + lds_tile_off(ws * (BM / 2) + jj * 32 + r, 2 * (ks + 1) + h));
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < MT; ++jj) mma_chunk<T>(fw[cur_f][i], fs[cur_f][jj], acc[i][jj]);
                __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (DYN)
                if (kt == 0 && tid == 0 && has_next) s_item[slot] = pulled;
            // the stage requested above has had this step's MFMAs to land; every wave is done reading buf
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }

        float qs1[8], qs2[8];                                  // this tile's sums after the two quad steps: value 4p + (lane & 3)
#pragma unroll
        for (int p = 0; p < 8; ++p) qs1[p] = qs2[p] = 0.f;
        if constexpr (RMODE) {
            // four quarters (sample tiles jj = 0..3); two 32 KiB halves of the released ring buffer take them in
            // turn.  Waits are counted: vmcnt retires in issue order, so "all but the N youngest" leaves a quarter's 4 stores
            // and the NEXT quarter's 4 DMAs in flight while guaranteeing the quarter about to be read has landed (a ragged
            // tile, whose row-masked stores may be skipped, waits for everything instead).
            // One instantiation with row masks serves both: a second, mask-free copy of this code cost registers.
            // The epilogue's per-lane index arithmetic (16 DMA sources, LDS offsets, store addresses) is tile-invariant;
            // hoisted out of the tile loop it would sit in ~60 registers through the K loop and spill.  An opaque zero,
            // re-read every tile, keeps it inside the epilogue (a few dozen integer instructions per tile).
            int zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
            const int lane = (tid & 63) + zero, r = lane & 31, h = lane >> 5;
            const int64_t mw0 = m0 + ws * (BM / 2);
            const int fw0 = f0 + wf * 64;
            T* base = Cg + (mw0 + r) * a.ldc + fw0 + 8 * h;
            const unsigned char* Rlo = smem + (buf ^ 1) * STAGE;
            const unsigned char* Rhi = Rlo + 32768;
            const uint32_t lo = lds0 + (buf ^ 1) * STAGE, hi = lo + 32768;
            const bool full = m0 + BM <= a.M;
            nt256p_issue_quarter<MT>(a, m0, f0, 0, lo, wave_u, lane);
            nt256p_issue_quarter<MT>(a, m0, f0, 1, hi, wave_u, lane);
            if (full) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nt256p_quarter<EPI, MT, 0>(a, acc, Rlo, bias_s, mw0, base, fw0, ws, wf, r, h, lane, dkey, qs1, qs2);
            __syncthreads();                                                     // lo has been read by every wave
            nt256p_issue_quarter<MT>(a, m0, f0, 2, lo, wave_u, lane);
            if (full) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nt256p_quarter<EPI, MT, 1>(a, acc, Rhi, bias_s, mw0, base, fw0, ws, wf, r, h, lane, dkey, qs1, qs2);
            __syncthreads();                                                     // hi has been read by every wave
            nt256p_issue_quarter<MT>(a, m0, f0, 3, hi, wave_u, lane);
            if (full) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nt256p_quarter<EPI, MT, 2>(a, acc, Rlo, bias_s, mw0, base, fw0, ws, wf, r, h, lane, dkey, qs1, qs2);
            if (full) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nt256p_quarter<EPI, MT, 3>(a, acc, Rhi, bias_s, mw0, base, fw0, ws, wf, r, h, lane, dkey, qs1, qs2);
            if (has_next) __syncthreads();                                       // the next tile's second stage goes into this buffer
        } else {
            const int64_t mw0 = m0 + ws * (BM / 2);
            T* base = Cg + (mw0 + r) * a.ldc + f0 + wf * 64 + 8 * h;
            if (m0 + BM > a.M) nt256p_convert<EPI, MT, false>(a, acc, mw0, base, r, lane, qs1, qs2);
            else nt256p_convert<EPI, MT, true>(a, acc, mw0, base, r, lane, qs1, qs2);
        }
        if constexpr (STATS) {
            // remaining butterfly steps (lane bits 2..4)
#pragma unroll
            for (int s = 2, n = 8; s < 5; ++s, n >>= 1) {
                const bool odd = (lane >> s) & 1;
#pragma unroll
                for (int p = 0; p < n / 2; ++p) {
                    const float k1 = odd ? qs1[2 * p + 1] : qs1[2 * p], g1 = odd ? qs1[2 * p] : qs1[2 * p + 1];
                    qs1[p] = k1 + __shfl_xor(g1, 1 << s, 64);
                    if constexpr (EPI != EPI_DGRAD_BN) {
                        const float k2 = odd ? qs2[2 * p + 1] : qs2[2 * p], g2 = odd ? qs2[2 * p] : qs2[2 * p + 1];
                        qs2[p] = k2 + __shfl_xor(g2, 1 << s, 64);
                    }
                }
            }
            // lane r of half h holds value r = i*16 + 4q + e  ->  feature wf*64 + i*32 + 8q + 4h + e.  One partial row
            // per sample tile, whichever block ran it: the sums downstream do not depend on the schedule.
            if constexpr (DYN) {
                const int fl = wf * 64 + (r >> 4) * 32 + ((r >> 2) & 3) * 8 + 4 * h + (r & 3);
                red[ws * BN + fl] = qs1[0];
                if constexpr (EPI != EPI_DGRAD_BN) red[(2 + ws) * BN + fl] = qs2[0];
                __syncthreads();                               // the next write of red is a K loop of barriers away
                const int which = tid / BN, col = tid % BN;
                const int64_t prow = (int64_t)(cur / tiles_f) * 8 + xcd;
                const float v = red[(which * 2) * BN + col] + red[(which * 2 + 1) * BN + col];
                if constexpr (EPI == EPI_DGRAD_BN) {
                    if (which == 0) a.partials[prow * a.F + f0 + col] = v;        // bias gradient of the layer below: rows of F
                } else {
                    a.partials[(prow * 2 + which) * a.F + f0 + col] = v;
                }
            } else {
                tot1 += qs1[0];
                tot2 += qs2[0];
            }
        }
        if (!has_next) break;
        cur = nxt;
        if constexpr (DYN) {
            nxt = __builtin_amdgcn_readfirstlane(s_item[slot]);
            slot ^= 1;
        } else {
            nxt += J;
        }
    }

    if constexpr (STATS && !DYN) {
        // one partial row per block and column tile: row (j / tiles_f) * 8 + xcd, the block's first sample tile
        const int j = blockIdx.x >> 3;
        if (j < items) {
            const int fl = wf * 64 + (r >> 4) * 32 + ((r >> 2) & 3) * 8 + 4 * h + (r & 3);
            red[ws * BN + fl] = tot1;
            red[(2 + ws) * BN + fl] = tot2;
            __syncthreads();
            const int which = tid / BN, col = tid % BN;
            const int64_t prow = (int64_t)(j / tiles_f) * 8 + xcd;
            const int f0 = (j % tiles_f) * BN;
            const float v = red[(which * 2) * BN + col] + red[(which * 2 + 1) * BN + col];
            if constexpr (EPI == EPI_DGRAD_BN) {
                if (which == 0) a.partials[prow * a.F + f0 + col] = v;
            } else {
                a.partials[(prow * 2 + which) * a.F + f0 + col] = v;
            }
        }
    }
    // the last block out leaves the counters at zero for the next launch on this stream
    if constexpr (DYN)
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(a.sched + 8 * 32, 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x < 9; ++x) a.sched[x * 32] = 0;
            __threadfence();
        }
    }
}

// One block per CU; the blocks of an XCD draw its tiles from a counter, so a CU that starts late or is shared with
// another stream's kernel (RCCL's, at N > 1) takes fewer tiles instead of holding the launch back: beside 8-32 held
// CUs the statically assigned version of this kernel ran 202-208 us instead of 150 (tools/contention_bench.py).
// The counters live in a per-stream slot of a module-global table, zero at load and re-zeroed by the last block of
// every launch.  *stat_rows = partial rows written.  EPI_FWD always applies ReLU (every fc layer of the reference
// has one; the projection runs the 128-tile kernel).
__device__ int g_nt256p_sched[NT256P_SCHED_SLOTS][9 * 32];      // [8 XCD counters + blocks finished] x 128 bytes

static inline int* nt256p_sched_slot(hipStream_t st) {
    static std::mutex mu;
    static hipStream_t seen[NT256P_SCHED_SLOTS];
    static int n = 0;
    static int* table = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!table && hipGetSymbolAddress((void**)&table, HIP_SYMBOL(g_nt256p_sched)) != hipSuccess) return nullptr;
    int s = 0;
    while (s < n && seen[s] != st) ++s;
    if (s == n) {
        if (n == NT256P_SCHED_SLOTS) return nullptr;
        seen[n++] = st;
    }
    return table + s * 9 * 32;
}

// dynamic: tiles drawn from the per-XCD counters (a GPU shared with other streams or processes); otherwise the static
// assignment (2-4 % faster alone on the GPU)
template <int EPI>
static inline hipError_t launch_gemm_nt256p(GemmNTArgs a, hipStream_t st, int* stat_rows, bool dynamic) {
    const int tiles_f = a.F / 256;
    if (tiles_f > NT256P_MAX_TILES_F) return hipErrorInvalidValue;
    const int64_t tiles_m = (a.M + 255) / 256;
    if (dynamic) {
        a.sched = nt256p_sched_slot(st);
        if (!a.sched) return hipErrorOutOfMemory;
        if (stat_rows) *stat_rows = (int)tiles_m;                // one partial row per sample tile
        hipLaunchKernelGGL((gemm_nt256p_kernel<EPI, 4, true>), dim3(256), dim3(512), 0, st, a);
    } else {
        const int J = 32 - (32 % tiles_f);                       // blocks per XCD, a multiple of tiles_f
        const int64_t slots = (int64_t)(J / tiles_f) * 8;        // sample tiles per round = partial rows
        a.sched = nullptr;
        if (stat_rows) *stat_rows = (int)(tiles_m < slots ? tiles_m : slots);
        hipLaunchKernelGGL((gemm_nt256p_kernel<EPI, 4, false>), dim3(8 * J), dim3(512), 0, st, a);
    }
    return hipGetLastError();
}

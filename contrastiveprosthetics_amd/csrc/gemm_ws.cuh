// Weight-stationary persistent NT GEMM for the fc layers with K = 512 (bf16):  C[m][f] = sum_k A[m][k] * W[f][k].
//
// The fc layers multiply 167,936 sample rows by the SAME 512 x 512 weight matrix.  gemm_nt256p.cuh re-stages a 256 x 64
// weight panel through LDS for every K step of every 256 x 256 tile: half of its 0.69 GB of L2 -> LDS fills and a third
// of its LDS fragment reads per launch are weights, and those two streams are what its K loop waits for.  MI355X has
// 512 KiB of vector registers per CU -- exactly one 512 x 512 bf16 matrix.  Here a workgroup is 4 wavefronts, one per
// SIMD with the whole 512-register file of its SIMD; wave w of a workgroup keeps the MFMA "A" fragments of 64 features x all
// 512 k (2 x 32 fragments of 4 registers = 256 registers) for the whole launch, loaded once.  Only sample rows move:
//
//   * a tile is 64 rows x 256 features (4 waves x 64 features); its A operand is 64 rows x 1 KiB = 64 KiB, i.e. WHOLE
//     rows: one LDS-DMA wave instruction (1 KiB) fetches one row, perfectly coalesced; two such buffers alternate;
//   * one barrier per TILE (128 MFMAs per wave), not per K step: the next tile's 64 row fetches are requested when a tile
//     starts and have the whole tile to land;
//   * per 16-deep k block a wave issues 2 fragment reads and 4 MFMAs (the old kernel: 3 reads per 4 MFMAs, plus the DMA
//     writes of the weight panel); L2 -> LDS traffic per launch falls from 0.69 GB to 0.35 GB;
//   * two accumulator sets alternate between tiles, and the finished tile's epilogue (bias is the initial accumulator; ReLU,
//     BatchNorm sums, bf16 conversion, 16-byte stores) is written into the NEXT tile's k loop, so that it issues in the
//     shadow of that tile's MFMAs -- with one wave per SIMD nothing else would overlap it;
//   * the BatchNorm column sums of a wave's 64 features stay in 64 registers for the whole launch (a wave always owns the
//     same features): no per-tile cross-lane reduction, one 32-lane butterfly at the end, one partial row per workgroup.
//
// LDS image of a tile: row r at r * 1024; its 64 16-byte chunks are XOR-swizzled with (r & 15) so that the 16 rows a
// ds_read_b128 lane group touches at one k position fall on 16 different bank groups; the DMA writes lane-linearly, so
// the permutation is applied to the per-lane SOURCE chunk (guide rule 21).
//
// Workgroup -> tile map: block b runs on XCD b & 7 (round-robin dispatch: speed only); its 32 blocks form 32 / nfb row
// workers of nfb = F / 256 blocks (one per 256-feature block) that walk the same row tiles at the same time, so a tile's
// rows come from HBM once per XCD.  Row tile t of 64 rows goes to XCD t & 7, worker (t >> 3) % workers.
#pragma once
#include "gemm_nt256p.cuh"     // glds16, quad_fold

#define WS_K 512
#define WS_RT 64
#define WS_TILE_BYTES (WS_RT * WS_K * 2)

// A 16-byte buffer store of freshly computed data.  hipcc (ROCm 7.2) may place a VALU instruction that overwrites the store's data
// registers directly behind the store -- for MUBUF stores with the offset in an SGPR LLVM's hazard recogniser assumes that is safe.  On
// gfx950 it is not: the store then sent the NEW register contents for lanes 12..15 and 44..47 (round 2: every "wrong values in lanes
// 12-15" fault of these kernels -- with spills, with coefficients kept in registers, with the split-k kernel -- came and went with the
// register allocation; two independent failing builds became bit-exact with this statement and nothing else).  Two wait states with the
// data registers still live.
template <typename V4, typename RSRC>
__device__ __forceinline__ void store_b128_settled(const V4& c, RSRC rsrc, uint32_t voff, uint32_t soff, int /*aux: always 0*/) {
    __builtin_amdgcn_raw_buffer_store_b128(c, rsrc, voff, soff, 0);
    asm volatile("s_nop 1" :: "v"(c));
}

// One LDS-DMA row fetch through a buffer descriptor: LDS[lds_dst + 16*lane] = 16 bytes at base + soff + voff, ZERO where
// that lies past the end of the buffer (the rows behind the last one of a ragged tile need no clamping).  Issued from
// inline asm so that hipcc does not track it (gemm_nt256.cuh, glds16); M0 saved, set and restored in the statement.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void bufl16_lds(const u32x4_t& rsrc, uint32_t voff, uint32_t soff_uniform, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(soff_uniform), "s"(lds_dst_uniform)
                 : "memory");
}

#ifdef CP_VARIANTS   // the 32x32x16 form of the weight-stationary forward kernel (superseded by gemm_ws16_kernel)
#include "../../tools/variants/gemm_ws32_fwd.cuh"
#endif

// ---------------------------------------------------------------------------------------------------------------------
// The forward kernel on v_mfma_f32_16x16x32_bf16.  Under the chip's power cap the 16x16x32 shape delivers 1.12-1.15x the
// FLOP/s of 32x32x16 at equal cycles per FLOP (guide, "DVFS give-back" item 7), and this kernel's floor IS that cap.
// Same tiles, same LDS image, same fragment-read count (per 32 k: 4 reads of 16 samples x 32 k feed 16 MFMAs); what changes
// is the accumulator layout -- register e of tile (ft, st) = feature ft*16 + 4*(lane>>4) + e of sample st*16 + (lane&15) --
// and with it the epilogue: v_permlane16_swap pairs the feature tiles ft, ft+1 so that a lane owns 8 consecutive features
// (one store instruction = 16 rows x 64 contiguous bytes instead of 32 x 32), and the BatchNorm sums are summed over the 4
// sample tiles in the lane first and folded over the 16 sample lanes once per tile (four DPP steps; see row16_fold8 on why not ds_swizzle:
// 16 values -> 1), carried in ONE register per statistic.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// 48-row tiles (3 sample tiles of 16): with 64 rows the two accumulator sets (128 registers) + 256 weight registers + fragments
// left hipcc 24 registers short (6 weight fragments spilled and reloaded every tile -- and spills are fatal here, see above)
// (64-row tiles fit without spills once the epilogue sits behind the k loop -- 214 registers -- and measured 97-98 us against 96)
#define WS16_RT 48
#define WS16_ST (WS16_RT / 16)
#define WS16_TILE_BYTES (WS16_RT * WS_K * 2)
struct Ws16Acc {
    f32x4_t t[4][WS16_ST];   // [16-feature tile ft][16-sample tile st]
};

// halving butterfly over a DPP row of 16 lanes: in = 8 per-lane values, out = the row's total of value (lane & 7) (lanes s and
// s ^ 8 end with the same number)
__device__ __forceinline__ float row16_fold8(float (&v)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = (b0 ? v[2 * p + 1] : v[2 * p]) + dpp_quad<0xB1>(b0 ? v[2 * p] : v[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 2; ++p) v[p] = (b1 ? v[2 * p + 1] : v[2 * p]) + dpp_quad<0x4E>(b1 ? v[2 * p] : v[2 * p + 1]);
    const float give = b2 ? v[0] : v[1];
    // lane ^ 4 and lane ^ 8 inside a row of 16 by DPP alone (no LDS crossbar): quad reverse (^3) then row_half_mirror (^7) = ^4;
    // row_ror:8 = ^8.  (-DFOLD_SWIZZLE: the ds_swizzle form, same lanes.)
#ifdef FOLD_SWIZZLE
    const float w = (b2 ? v[1] : v[0]) + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(give), 0x101F));            // lane ^ 4
    return w + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(w), 0x201F));                                          // lane ^ 8
#else
    const float w = (b2 ? v[1] : v[0]) + dpp_quad<0x141>(dpp_quad<0x1B>(give));
    return w + dpp_quad<0x128>(w);
#endif
}

// STATS = false (round 4): evaluation with the running statistics -- nobody reads the column sums, so the epilogue is max / convert / store
// (1.5 instead of 3.5 vector instructions per output beside the matrix pipe).
template <bool STATS = true>
__global__ __launch_bounds__(256, 1) void gemm_ws16_kernel(GemmNTArgs a) {
    constexpr int K = WS_K, KB = K / 32, RPW = WS16_RT / 4, ST = WS16_ST;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WS16_TILE_BYTES + 768 * 4];
    float* bias_s = (float*)(smem + 2 * WS16_TILE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + WS16_RT - 1) / WS16_RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    for (int q = tid; q < a.F; q += 256) bias_s[q] = a.bias[q];
    if (ntile == 0) return;
    const int f0 = fb * 256 + wave * 64;

    // weights: fragment (ft, kb) = rows f0 + ft*16 + (lane & 15), k = kb*32 + 8*(lane >> 4) .. +7; pinned in the accumulator file
    s16x8 wreg[4][KB];
    {
        const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)(f0 + s16) * K + 8 * q4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wreg[ft][kb]) : "v"(Wg + (int64_t)ft * 16 * K + kb * 32) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; kb += 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(wreg[ft][kb]), "+a"(wreg[ft][kb + 1]), "+a"(wreg[ft][kb + 2]), "+a"(wreg[ft][kb + 3]),
                             "+a"(wreg[ft][kb + 4]), "+a"(wreg[ft][kb + 5]), "+a"(wreg[ft][kb + 6]), "+a"(wreg[ft][kb + 7]));
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * (K * 2)), 0x00020000u};
    auto fetch_row = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, (uint32_t)(((lane ^ ((wave * RPW + q) & 15)) << 4) + q * 1024), tile_soff, lds0 + buf * WS16_TILE_BYTES + (wave * RPW + q) * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * WS16_RT; };

    // BatchNorm sums over all tiles: [fp] = the 16-lane row's total of feature f0 + fp*32 + ((lane >> 2) & 1)*16 + 4*q4 + (lane & 3)
    float qs1[2] = {0.f, 0.f}, qs2[2] = {0.f, 0.f};
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    // after the permlane16 swap a lane owns features foff .. foff + 7 of the 32-feature pair: foff = {0, 16, 8, 24}[q4]
    const int foff = ((q4 & 1) << 4) | ((q4 >> 1) << 3);
    const uint32_t c_lane = (uint32_t)(s16 * a.ldc + f0 + foff) * 2;
    const int d16 = (q4 ^ s16) << 4;

    float t1[8], t2[8];               // sums over the 4 sample tiles of the current feature-tile pair: value o*4 + e
    // epilogue slot u = 0 .. 2*ST-1 of a finished tile: feature-tile pair fp = u / ST, sample tile st = u % ST
    auto epi_slot = [&](Ws16Acc& old, int u, uint32_t s_old, const bool (&live)[ST]) {
        const int fp = u / ST, st = u % ST;
        uint2 pk[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int ft = 2 * fp + o;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(old.t[ft][st][e]));
                if constexpr (STATS) {
                    const float w = live[st] ? v[e] : 0.f;
                    if (st == 0) { t1[o * 4 + e] = w; t2[o * 4 + e] = w * w; }
                    else { t1[o * 4 + e] += w; t2[o * 4 + e] = fmaf(w, w, t2[o * 4 + e]); }
                }
            }
            pk[o].x = cvt_pk_bf16<false>(v[0], v[1]);
            pk[o].y = cvt_pk_bf16<false>(v[2], v[3]);
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
        const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
        store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.ldc + fp * 32) * 2, 0);
        if constexpr (STATS) {
            if (st == ST - 1) {
                qs1[fp] += row16_fold8(t1, lane);
                qs2[fp] += row16_fold8(t2, lane);
            }
        }
    };

    auto step = [&](Ws16Acc& acc, Ws16Acc& old, int ti, int buf, bool has_next, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const uint32_t next_soff = has_next ? (uint32_t)((row0(ti + 1) + wave * RPW) * (K * 2)) : 0xFFF00000u;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const float4 b4 = *(const float4*)(bias_s + f0 + ft * 16 + 4 * q4);
            const f32x4_t b0 = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int st = 0; st < ST; ++st) acc.t[ft][st] = b0;
        }
        const unsigned char* At = smem + buf * WS16_TILE_BYTES + s16 * 1024;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        bool all_live[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) all_live[st] = true;
        // one fragment per sample tile, re-read for the next k block right behind the 4 MFMAs that consumed it: the read has
        // the other sample tiles' 8 MFMAs (128 cycles) to return
        uint4 fa[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[st] = *(const uint4*)(At + st * 16384 + (0 ^ d16));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb < RPW) fetch_row(next_soff, buf ^ 1, kb);
#pragma unroll
            for (int st = 0; st < ST; ++st) {
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    acc.t[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ft][kb], __builtin_bit_cast(s16x8, fa[st]), acc.t[ft][st], 0, 0, 0);
                if (kb + 1 < KB) fa[st] = *(const uint4*)(At + st * 16384 + ((((kb + 1) * 4) << 4) ^ d16));
            }
#ifdef WS16_WOVEN_EPI
            if constexpr (WITH_EPI)
                if (kb >= 6 && kb < 6 + 2 * ST) epi_slot(old, kb - 6, s_old, all_live);
#endif
        }
#ifndef WS16_WOVEN_EPI
        // The previous tile's epilogue runs BEHIND the k loop, not inside it: with the matrix pipe of every CU busy, vector
        // instructions issued beside the MFMAs cost more than their own time (tools/coissue_probe.hip: 257 us of MFMAs + 141 us of
        // FMAs take 472 us interleaved); measured in the step 99-101 us per launch against 106-107 woven (-DWS16_WOVEN_EPI).
        // It stays the PREVIOUS tile's epilogue (two accumulator sets): with one set and the tile's own epilogue behind its k loop
        // the epilogue waits for the last MFMAs and the next tile's first MFMAs for the epilogue -- 114-119 us.
        if constexpr (WITH_EPI) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2 * ST; ++u) epi_slot(old, u, s_old, all_live);
            __builtin_amdgcn_sched_barrier(0);
            // the row fetches are older than the 2*ST stores just issued: wait for them only.  (vmcnt retires an LDS-DMA load
            // before every younger store: tools/vmcnt_order_probe.hip, 5e8 lane-reads behind vmcnt(1/4/6) without a stale one,
            // 99.8 % stale with the count one too high.  Safe alternatives measured slower: wait + barrier in front of the
            // epilogue 100-102 us, the outputs held in registers until the barrier is passed 100-102 us, this 94-97 us.)
            static_assert(2 * ST == 6, "the wait below counts the epilogue's stores");
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();
    };
    auto drain = [&](Ws16Acc& old, int64_t m_old) {
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        bool live[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) live[st] = m_old + st * 16 + s16 < a.M;
#pragma unroll
        for (int u = 0; u < 2 * ST; ++u) epi_slot(old, u, s_old, live);
    };

    Ws16Acc accA, accB;
    {
        const uint32_t soff0 = (uint32_t)((row0(0) + wave * RPW) * (K * 2));
#pragma unroll
        for (int q = 0; q < RPW; ++q) fetch_row(soff0, 0, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step(accA, accB, 0, 0, ntile > 1, std::false_type{}, 0);
    int ti = 1;
    while (ti + 1 < ntile) {
        step(accB, accA, ti, ti & 1, true, std::true_type{}, row0(ti - 1));
        step(accA, accB, ti + 1, (ti + 1) & 1, ti + 2 < ntile, std::true_type{}, row0(ti));
        ti += 2;
    }
    if (ti < ntile) {
        step(accB, accA, ti, ti & 1, false, std::true_type{}, row0(ti - 1));
        drain(accB, row0(ti));
    } else {
        drain(accA, row0(ntile - 1));
    }
    // the four 16-lane rows (q4) of the wave hold DIFFERENT features: no further reduction.  Lane (q4, s16), pair fp: value
    // s16 & 7 = o*4 + e  ->  feature f0 + fp*32 + o*16 + 4*q4 + e; lanes s16 < 8 write
    if (STATS && s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
            const int f = f0 + fp * 32 + (s16 >> 2) * 16 + 4 * q4 + (s16 & 3);
            a.partials[(prow * 2 + 0) * a.F + f] = qs1[fp];
            a.partials[(prow * 2 + 1) * a.F + f] = qs2[fp];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fc1 forward (K = 768): rows of 1536 bytes, LDS-DMA units of 1 KiB that run across row boundaries, the 16-byte chunks XOR-swizzled with
// the row inside 256-byte groups (applied to the DMA's per-lane source), so the 16 rows of a fragment read fall on 16 different chunk positions.
#define WSK_K 768
#define WSK_RT 32
#define WSK_ROWB (WSK_K * 2)
#define WSK_TILE_BYTES (WSK_RT * WSK_ROWB)
#ifdef CP_VARIANTS   // rounds 2-3's form: the k range split over wave pairs, partial sums exchanged through LDS (superseded by gemm_ws16n_kernel)
#include "../../tools/variants/gemm_ws16k_splitk.cuh"
#endif

// ---------------------------------------------------------------------------------------------------------------------
// fc1 forward (K = 768), round 4 ("the 8-bit kernel got rid of the wave-pair exchange by making one wave own a whole
// output slice; find the bf16 equivalent"): a wave keeps 32 features x ALL 768 k = 2 x 24 fragments = 192 registers, so a workgroup is
// 4 waves x 32 = 128 features as before, but no k split: no partial-sum exchange through LDS, no second barrier role, the tile loop
// is gemm_ws16_kernel's -- 48-row tiles (3 sample tiles; 2 x 72 KiB of LDS), the finished tile's epilogue behind the next tile's k loop,
// one barrier per tile.  Per tile and wave: 144 MFMAs on 72 fragment reads (the split-k form: 96 on 24 + the exchange); every wave reads
// the whole tile from LDS, which is what the exchange bought off -- measured, not assumed: see DESIGN.md section 9.
// LDS image and fetch units as gemm_ws16k_kernel (1 KiB units across 1536-byte rows, chunks XOR-swizzled with the row in 256-byte groups).
// ---------------------------------------------------------------------------------------------------------------------
#define WSN_RT 48
#define WSN_TILE_BYTES (WSN_RT * WSK_ROWB)
template <bool STATS = true>          // (false: evaluation with the running statistics, as gemm_ws16_kernel)
__global__ __launch_bounds__(256, 1) void gemm_ws16n_kernel(GemmNTArgs a) {
    constexpr int K = WSK_K, KB = K / 32, RT = WSN_RT, ST = RT / 16, UPW = WSN_TILE_BYTES / 1024 / 4;
    static_assert(UPW <= KB, "one fetch unit per k block and wave");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WSN_TILE_BYTES + 512 * 4];
    float* bias_s = (float*)(smem + 2 * WSN_TILE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 7;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + RT - 1) / RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    for (int q = tid; q < a.F; q += 256) bias_s[q] = a.bias[q];
    if (ntile == 0) return;
    const int f0 = fb * 128 + wave * 32;

    s16x8 wreg[2][KB];
    {
        const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)(f0 + s16) * K + 8 * q4;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wreg[ft][kb]) : "v"(Wg + (int64_t)ft * 16 * K + kb * 32) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; kb += 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(wreg[ft][kb]), "+a"(wreg[ft][kb + 1]), "+a"(wreg[ft][kb + 2]), "+a"(wreg[ft][kb + 3]),
                             "+a"(wreg[ft][kb + 4]), "+a"(wreg[ft][kb + 5]), "+a"(wreg[ft][kb + 6]), "+a"(wreg[ft][kb + 7]));
    }

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * WSK_ROWB), 0x00020000u};
    uint32_t fsrc[UPW];
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        const int g = (wave * UPW + q) * 64 + lane, row = g / 96, pc = g % 96;
        fsrc[q] = (uint32_t)(row * WSK_ROWB + (((pc & ~15) | ((pc ^ row) & 15)) << 4));
    }
    auto fetch_unit = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, fsrc[q], tile_soff, lds0 + buf * WSN_TILE_BYTES + (wave * UPW + q) * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };

    float qs1 = 0.f, qs2 = 0.f;
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    const int foff = ((q4 & 1) << 4) | ((q4 >> 1) << 3);
    const uint32_t c_lane = (uint32_t)(s16 * a.ldc + f0 + foff) * 2;
    const int d16 = (q4 ^ s16) << 4;

    float t1[8], t2[8];
    auto epi_slot = [&](f32x4_t (&old)[2][ST], int st, uint32_t s_old, const bool (&live)[ST]) {
        uint2 pk[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(old[o][st][e]));
                if constexpr (STATS) {
                    const float w = live[st] ? v[e] : 0.f;
                    if (st == 0) { t1[o * 4 + e] = w; t2[o * 4 + e] = w * w; }
                    else { t1[o * 4 + e] += w; t2[o * 4 + e] = fmaf(w, w, t2[o * 4 + e]); }
                }
            }
            pk[o].x = cvt_pk_bf16<false>(v[0], v[1]);
            pk[o].y = cvt_pk_bf16<false>(v[2], v[3]);
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
        const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
        store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.ldc) * 2, 0);
        if constexpr (STATS) {
            if (st == ST - 1) {
                qs1 += row16_fold8(t1, lane);
                qs2 += row16_fold8(t2, lane);
            }
        }
    };

    auto step = [&](f32x4_t (&acc)[2][ST], f32x4_t (&old)[2][ST], int ti, int buf, bool has_next, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const uint32_t next_soff = has_next ? (uint32_t)(row0(ti + 1) * WSK_ROWB) : 0xFFF00000u;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const float4 b4 = *(const float4*)(bias_s + f0 + ft * 16 + 4 * q4);
            const f32x4_t b0 = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int st = 0; st < ST; ++st) acc[ft][st] = b0;
        }
        const unsigned char* At = smem + buf * WSN_TILE_BYTES + s16 * WSK_ROWB;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        bool all_live[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) all_live[st] = true;
        // fragments of k block kb + 1 are requested in front of the 6 MFMAs of block kb (two register sets)
        uint4 fa[2][ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[0][st] = *(const uint4*)(At + st * 16 * WSK_ROWB + (0 ^ d16));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb < UPW) fetch_unit(next_soff, buf ^ 1, kb);
            if (kb + 1 < KB) {
#pragma unroll
                for (int st = 0; st < ST; ++st) fa[(kb + 1) & 1][st] = *(const uint4*)(At + st * 16 * WSK_ROWB + ((((kb + 1) * 4) << 4) ^ d16));
            }
#pragma unroll
            for (int st = 0; st < ST; ++st)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
                    acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ft][kb], __builtin_bit_cast(s16x8, fa[kb & 1][st]), acc[ft][st], 0, 0, 0);
        }
        if constexpr (WITH_EPI) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < ST; ++u) epi_slot(old, u, s_old, all_live);
            __builtin_amdgcn_sched_barrier(0);
            static_assert(ST == 3, "the wait below counts the epilogue's stores");
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                 // the fetches are older than the 3 stores (gemm_ws16_kernel)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    auto drain = [&](f32x4_t (&old)[2][ST], int64_t m_old) {
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        bool live[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) live[st] = m_old + st * 16 + s16 < a.M;
#pragma unroll
        for (int u = 0; u < ST; ++u) epi_slot(old, u, s_old, live);
    };

    f32x4_t accA[2][ST], accB[2][ST];
    {
        const uint32_t soff0 = (uint32_t)(row0(0) * WSK_ROWB);
#pragma unroll
        for (int q = 0; q < UPW; ++q) fetch_unit(soff0, 0, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step(accA, accB, 0, 0, ntile > 1, std::false_type{}, 0);
    int ti = 1;
    while (ti + 1 < ntile) {
        step(accB, accA, ti, ti & 1, true, std::true_type{}, row0(ti - 1));
        step(accA, accB, ti + 1, (ti + 1) & 1, ti + 2 < ntile, std::true_type{}, row0(ti));
        ti += 2;
    }
    if (ti < ntile) {
        step(accB, accA, ti, ti & 1, false, std::true_type{}, row0(ti - 1));
        drain(accB, row0(ti));
    } else {
        drain(accA, row0(ntile - 1));
    }
    // lane (q4, s16 < 8): value s16 = o*4 + e  ->  feature f0 + o*16 + 4*q4 + e
    if (STATS && s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
        const int f = f0 + (s16 >> 2) * 16 + 4 * q4 + (s16 & 3);
        a.partials[(prow * 2 + 0) * a.F + f] = qs1;
        a.partials[(prow * 2 + 1) * a.F + f] = qs2;
    }
}

static inline hipError_t launch_gemm_ws16n(const GemmNTArgs& a, hipStream_t st, int* stat_rows) {
    if (a.K != WSK_K || (a.F & 127) || a.F > 512 || a.lda != WSK_K || !a.bias) return hipErrorInvalidValue;
    const int nwk = 32 / (a.F >> 7);
    const int64_t tiles = (a.M + WSN_RT - 1) / WSN_RT, workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
    if (a.partials != nullptr) hipLaunchKernelGGL(gemm_ws16n_kernel<true>, dim3(256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm_ws16n_kernel<false>, dim3(256), dim3(256), 0, st, a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// The same structure for the fc DATA gradients whose epilogue applies BatchNorm + ReLU backward of the layer below
// (EPI_DGRAD_BN; gemm_nt256p.cuh for what that means):  C[m][f] = [r > 0] (ca[f] * (A W^T)[m][f] + cb[f] * r[m][f] + cz[f]),
// r = the saved activation of the layer below, column sums of C = that layer's bias gradient.  K = 512, F = 512 or 768.
// Differences from the forward kernel: a tile is 32 rows (A: 32 KiB per buffer), which leaves LDS room for the saved
// activation -- each wave fetches the 32 x 64 sub-tile it will need (4 KiB, four LDS-DMA instructions of 8 rows x 128 bytes,
// two buffers) while the tile's k loop runs and reads it back in the accumulator layout (ds_read_b64, 16-byte chunks
// XOR-swizzled with (row >> 1) & 7) during the NEXT tile's k loop, where the epilogue is woven in; the wave consumes only
// what it fetched itself, so its own vmcnt wait at the end of the tile is all the ordering that is needed.  Two accumulator
// sets of 32 registers.  The three coefficient rows of the block's 256 features sit in LDS.
// ---------------------------------------------------------------------------------------------------------------------
#define WSD_RT 32
#define WSD_TILE_BYTES (WSD_RT * WS_K * 2)
#define WSD16_RT 32          // 48-row tiles (as in the forward) spill here: the epilogue holds R and the coefficients as well

#ifdef CP_VARIANTS   // the 32x32x16 form of the BN-fused data gradient (superseded by gemm_wsd16_kernel<0>)
#include "../../tools/variants/gemm_wsd32_bn.cuh"
#endif

// The data-gradient kernel on v_mfma_f32_16x16x32_bf16 (see gemm_ws16_kernel): 32-row tiles = 2 sample tiles of 16, the saved
// activation read in the 16x16 accumulator layout (lane (q4, s): features ft*16 + 4*q4 .. +3 of row st*16 + s).
// MODE 0 (EPI_DGRAD_BN): BatchNorm + ReLU backward of the layer below in the epilogue (coef), column sums of the result = its bias
// gradient.  MODE 1 (EPI_DGRAD_ST, behind a dropout): the dropout mask of the layer's input (the forward pass's hash) and the two
// BatchNorm-backward sums of the masked gradient against the saved activation, partial rows [2][F].
template <int MODE>
__global__ __launch_bounds__(256, 1) void gemm_wsd16_kernel(GemmNTArgs a) {
    constexpr bool STATS = MODE == 1;
    constexpr int RT = WSD16_RT, TILE_BYTES = RT * WS_K * 2, R_BYTES = RT * 128;          // R: per wave and buffer, rows of 64 features
    constexpr int K = WS_K, KB = K / 32, RPW = RT / 4, ST = RT / 16;
    constexpr int R_OFF = 2 * TILE_BYTES, COEF_OFF = R_OFF + 4 * 2 * R_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[COEF_OFF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + RT - 1) / RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    if (ntile == 0) return;
    const uint32_t dkey = (STATS && a.dp_thresh != 0) ? (a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key) : 0u;
    const int f0 = fb * 256 + wave * 64;
    // MODE 0: the three BatchNorm-backward coefficients of this lane's 16 features stay in registers (48 of them; read from an LDS
    // table per epilogue slot they were 6 of its 8 ds_read_b128)
    float4 cfa[STATS ? 1 : 4], cfb[STATS ? 1 : 4], cfz[STATS ? 1 : 4];
    if constexpr (!STATS) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float v[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] = a.coef[c * a.coef_mod + (f0 + ft * 16 + 4 * (lane >> 4) + e) % a.coef_mod];
            cfa[ft] = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
            cfb[ft] = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
            cfz[ft] = make_float4(v[2][0], v[2][1], v[2][2], v[2][3]);
        }
    }

    s16x8 wreg[4][KB];
    {
        const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)(f0 + s16) * K + 8 * q4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wreg[ft][kb]) : "v"(Wg + (int64_t)ft * 16 * K + kb * 32) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; kb += 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(wreg[ft][kb]), "+a"(wreg[ft][kb + 1]), "+a"(wreg[ft][kb + 2]), "+a"(wreg[ft][kb + 3]),
                             "+a"(wreg[ft][kb + 4]), "+a"(wreg[ft][kb + 5]), "+a"(wreg[ft][kb + 6]), "+a"(wreg[ft][kb + 7]));
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A, r_base = (uint64_t)(uintptr_t)a.R;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * (K * 2)), 0x00020000u};
    const u32x4_t r_rsrc = {(uint32_t)r_base, (uint32_t)(r_base >> 32) & 0xFFFFu, (uint32_t)(a.M * a.ldr * 2), 0x00020000u};
    auto fetch_a = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, (uint32_t)(((lane ^ ((wave * RPW + q) & 15)) << 4) + q * 1024), tile_soff, lds0 + buf * TILE_BYTES + (wave * RPW + q) * 1024);
    };
    const uint32_t r_lane = (uint32_t)((lane >> 3) * a.ldr * 2 + f0 * 2);
    auto fetch_r = [&](uint32_t tile_soff, int buf, int k) {
        const int row = 8 * k + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        bufl16_lds(r_rsrc, r_lane + (uint32_t)(8 * k * a.ldr * 2 + lc * 16), tile_soff, lds0 + R_OFF + (wave * 2 + buf) * R_BYTES + k * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };

    float qs1[2] = {0.f, 0.f}, qs2[2] = {0.f, 0.f};
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    const int foff = ((q4 & 1) << 4) | ((q4 >> 1) << 3);
    const uint32_t c_lane = (uint32_t)(s16 * a.ldc + f0 + foff) * 2;
    const int d16 = (q4 ^ s16) << 4;

    float t1[8], t2[8];
    // epilogue slot u = 0 .. 2*ST-1 of a finished tile: feature-tile pair fp = u / ST, sample tile st = u % ST
    auto epi_slot = [&](f32x4_t (&old)[4][ST], const unsigned char* Rw, int u, uint32_t s_old, const bool (&live)[ST], int64_t m_old) {
        const int fp = u / ST, st = u % ST;
        const int row = st * 16 + s16;
        const int rsw = (row >> 1) & 7;
        uint2 pk[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int ft = 2 * fp + o;
            const uint2 rr = *(const uint2*)(Rw + row * 128 + (((ft * 2 + (q4 >> 1)) ^ rsw) << 4) + 8 * (q4 & 1));
            const float r0 = __uint_as_float(rr.x << 16), r1 = __uint_as_float(rr.x & 0xffff0000u);
            const float r2 = __uint_as_float(rr.y << 16), r3 = __uint_as_float(rr.y & 0xffff0000u);
            float y0 = old[ft][st][0], y1 = old[ft][st][1], y2 = old[ft][st][2], y3 = old[ft][st][3];
            if constexpr (!STATS) {
                const float4 ca = cfa[ft], cb = cfb[ft], cz = cfz[ft];
                y0 = r0 > 0.f ? fmaf(ca.x, y0, fmaf(cb.x, r0, cz.x)) : 0.f;
                y1 = r1 > 0.f ? fmaf(ca.y, y1, fmaf(cb.y, r1, cz.y)) : 0.f;
                y2 = r2 > 0.f ? fmaf(ca.z, y2, fmaf(cb.z, r2, cz.z)) : 0.f;
                y3 = r3 > 0.f ? fmaf(ca.w, y3, fmaf(cb.w, r3, cz.w)) : 0.f;
            } else if (a.dp_thresh != 0) {
                const uint32_t col = (uint32_t)(f0 + ft * 16 + 4 * q4);            // column of y0 in the output row (even)
                const uint32_t m = (uint32_t)(m_old + row);
                const uint32_t p0 = dropout_pair(dkey, m, (uint32_t)a.ldc, col);
                const uint32_t p1 = dropout_pair(dkey, m, (uint32_t)a.ldc, col + 2);
                y0 *= dropout_scale(p0, 0, a.dp_thresh, a.dp_inv_keep);
                y1 *= dropout_scale(p0, 1, a.dp_thresh, a.dp_inv_keep);
                y2 *= dropout_scale(p1, 0, a.dp_thresh, a.dp_inv_keep);
                y3 *= dropout_scale(p1, 1, a.dp_thresh, a.dp_inv_keep);
            }
            pk[o].x = cvt_pk_bf16<false>(y0, y1);
            pk[o].y = cvt_pk_bf16<false>(y2, y3);
            // column sums of the values as stored (MODE 0: the layer's bias gradient); rows past the end do not count
            float g[4] = {__uint_as_float(pk[o].x << 16), __uint_as_float(pk[o].x & 0xffff0000u),
                          __uint_as_float(pk[o].y << 16), __uint_as_float(pk[o].y & 0xffff0000u)};
            const float rv[4] = {r0, r1, r2, r3};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w = live[st] ? g[e] : 0.f;
                if (st == 0) t1[o * 4 + e] = w; else t1[o * 4 + e] += w;
                if constexpr (STATS) {
                    if (st == 0) t2[o * 4 + e] = w * rv[e]; else t2[o * 4 + e] = fmaf(w, rv[e], t2[o * 4 + e]);
                }
            }
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
        const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
        store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.ldc + fp * 32) * 2, 0);
        if (st == ST - 1) {
            qs1[fp] += row16_fold8(t1, lane);
            if constexpr (STATS) qs2[fp] += row16_fold8(t2, lane);
        }
    };

    auto step = [&](f32x4_t (&acc)[4][ST], f32x4_t (&old)[4][ST], int ti, int buf, bool has_next, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const uint32_t next_soff = has_next ? (uint32_t)((row0(ti + 1) + wave * RPW) * (K * 2)) : 0xFFF00000u;
        const uint32_t r_soff = (uint32_t)(row0(ti) * a.ldr * 2);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int st = 0; st < ST; ++st) acc[ft][st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const unsigned char* At = smem + buf * TILE_BYTES + s16 * 1024;
        const unsigned char* Rw = smem + R_OFF + (wave * 2 + (buf ^ 1)) * R_BYTES;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        bool all_live[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) all_live[st] = true;
        uint4 fa[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[st] = *(const uint4*)(At + st * 16384 + (0 ^ d16));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb < RPW) fetch_a(next_soff, buf ^ 1, kb);
            if ((kb & 1) == 0 && (kb >> 1) < RT / 8) fetch_r(r_soff, buf, kb >> 1);
#pragma unroll
            for (int st = 0; st < ST; ++st) {
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ft][kb], __builtin_bit_cast(s16x8, fa[st]), acc[ft][st], 0, 0, 0);
                if (kb + 1 < KB) fa[st] = *(const uint4*)(At + st * 16384 + ((((kb + 1) * 4) << 4) ^ d16));
            }
            // MODE 1 (dropout): woven into the k loop; MODE 0 (BN): behind it, as in gemm_ws16_kernel.  Measured in the step,
            // alternating runs: BN mode 131 us behind the loop against 136 woven, dropout mode 135 against 126.
            if constexpr (WITH_EPI && STATS)
                if (kb >= 4 && kb < 4 + 2 * 2 * ST && (kb & 1) == 0) epi_slot(old, Rw, (kb - 4) >> 1, s_old, all_live, m_old);
        }
        if constexpr (WITH_EPI && !STATS) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2 * ST; ++u) epi_slot(old, Rw, u, s_old, all_live, m_old);
            __builtin_amdgcn_sched_barrier(0);
            static_assert(2 * ST == 4, "the wait below counts the epilogue's stores");
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // the fetches are older than the 4 stores
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    auto drain = [&](f32x4_t (&old)[4][ST], int buf, int64_t m_old) {
        const unsigned char* Rw = smem + R_OFF + (wave * 2 + buf) * R_BYTES;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        bool live[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) live[st] = m_old + st * 16 + s16 < a.M;
#pragma unroll
        for (int u = 0; u < 2 * ST; ++u) epi_slot(old, Rw, u, s_old, live, m_old);
    };

    f32x4_t accA[4][ST], accB[4][ST];
    {
        const uint32_t soff0 = (uint32_t)((row0(0) + wave * RPW) * (K * 2));
#pragma unroll
        for (int q = 0; q < RPW; ++q) fetch_a(soff0, 0, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step(accA, accB, 0, 0, ntile > 1, std::false_type{}, 0);
    int ti = 1;
    while (ti + 1 < ntile) {
        step(accB, accA, ti, ti & 1, true, std::true_type{}, row0(ti - 1));
        step(accA, accB, ti + 1, (ti + 1) & 1, ti + 2 < ntile, std::true_type{}, row0(ti));
        ti += 2;
    }
    if (ti < ntile) {
        step(accB, accA, ti, ti & 1, false, std::true_type{}, row0(ti - 1));
        drain(accB, ti & 1, row0(ti));
    } else {
        drain(accA, (ntile - 1) & 1, row0(ntile - 1));
    }
    if (s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
            const int f = f0 + fp * 32 + (s16 >> 2) * 16 + 4 * q4 + (s16 & 3);
            if constexpr (STATS) {
                a.partials[(prow * 2 + 0) * a.F + f] = qs1[fp];                  // rows of [2][F]
                a.partials[(prow * 2 + 1) * a.F + f] = qs2[fp];
            } else {
                a.partials[prow * a.F + f] = qs1[fp];                            // bias gradient of the layer below: rows of F
            }
        }
    }
}

// ---- the projection's data gradient (K = 16 live columns of dz, F = 512) behind fc7's dropout ---------------------------------
// g = mask * (dz W) is a rank-16 product: cheap enough to compute TWICE instead of writing it out and reading it back for the
// BatchNorm + ReLU backward pass (172 MB each way + the pass's own launch).  PASS 0: the two BatchNorm-backward sums of g against the
// saved activation R, nothing stored (reads R once).  PASS 1, after the coefficients are final: recompute g, apply
// r > 0 ? ca*g + cb*r + cz : 0, store, column sums of the result (= fc7's bias gradient).  One v_mfma_f32_16x16x16_bf16 per 16 x 16
// outputs, its operands straight from global memory (8 bytes per lane); a wave owns 64 features of a 32-row tile, fetches its R sub-tile
// by LDS-DMA one tile ahead and never meets the other waves (no barrier in the loop).  Partial rows as gemm_wsd16_kernel writes them.
#define PROJ_RT 32
template <int PASS>
__global__ __launch_bounds__(256, 2) void proj_dgrad_kernel(GemmNTArgs a) {
    constexpr int RT = PROJ_RT, ST = RT / 16, R_BYTES = RT * 128;
    constexpr int COEF_OFF = 4 * 2 * R_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[COEF_OFF + 3 * 256 * 4];
    float* coef_s = (float*)(smem + COEF_OFF);                               // [3][256]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = (gridDim.x >> 3) / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + RT - 1) / RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    if (ntile == 0) return;
    if constexpr (PASS == 1) {
        for (int q = tid; q < 3 * 256; q += 256) {
            const int c = q >> 8, f = fb * 256 + (q & 255);
            coef_s[q] = a.coef[c * a.coef_mod + f % a.coef_mod];
        }
        __syncthreads();
    }
    const int f0 = fb * 256 + wave * 64, fl0 = wave * 64;
    const uint32_t dkey = a.dp_thresh != 0 ? (a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key) : 0u;

    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    s16x4_t wfrag[4];                                                        // W[f0 + ft*16 + s16][4*q4 .. +3]
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) wfrag[ft] = *(const s16x4_t*)((const bf16_t*)a.W + (int64_t)(f0 + ft * 16 + s16) * a.K + 4 * q4);

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint64_t r_base = (uint64_t)(uintptr_t)a.R;
    const u32x4_t r_rsrc = {(uint32_t)r_base, (uint32_t)(r_base >> 32) & 0xFFFFu, (uint32_t)(a.M * a.ldr * 2), 0x00020000u};
    const uint32_t r_lane = (uint32_t)((lane >> 3) * a.ldr * 2 + f0 * 2);
    auto fetch_r = [&](int64_t m0, int buf) {
#pragma unroll
        for (int k = 0; k < RT / 8; ++k) {
            const int row = 8 * k + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            // (the whole offset in the per-lane part: the bounds check that zeroes the rows past the end is on it)
            bufl16_lds(r_rsrc, (uint32_t)(m0 * a.ldr * 2) + r_lane + (uint32_t)(8 * k * a.ldr * 2 + lc * 16), 0u, lds0 + (wave * 2 + buf) * R_BYTES + k * 1024);
        }
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };
    auto load_dz = [&](int64_t m0, s16x4_t (&d)[ST]) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            int64_t m = m0 + st * 16 + s16;
            if (m >= a.M) m = a.M - 1;                                       // (those rows are masked out of the sums and never stored)
            d[st] = *(const s16x4_t*)((const bf16_t*)a.A + m * a.lda + 4 * q4);
        }
    };

    float qs1[2] = {0.f, 0.f}, qs2[2] = {0.f, 0.f};
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    const int foff = ((q4 & 1) << 4) | ((q4 >> 1) << 3);
    const uint32_t c_lane = (uint32_t)(s16 * a.ldc + f0 + foff) * 2;

    s16x4_t dzf[ST], dzn[ST];
    fetch_r(row0(0), 0);
    load_dz(row0(0), dzf);
    for (int ti = 0; ti < ntile; ++ti) {
        const int buf = ti & 1;
        const int64_t m0 = row0(ti);
        // this tile's R sub-tile has landed (own DMA only).  PASS 1: the previous tile's 2*ST stores are younger than that fetch and
        // stay in flight (tools/vmcnt_order_probe.hip)
        if (PASS == 1 && ti > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ti + 1 < ntile) { fetch_r(row0(ti + 1), buf ^ 1); load_dz(row0(ti + 1), dzn); }
        f32x4_t acc[4][ST];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int st = 0; st < ST; ++st)
                acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfrag[ft], dzf[st], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const unsigned char* Rw = smem + (wave * 2 + buf) * R_BYTES;
        const uint32_t s_out = (uint32_t)(m0 * a.ldc * 2);
        float t1[8], t2[8];
#pragma unroll
        for (int u = 0; u < 2 * ST; ++u) {
            const int fp = u / ST, st = u % ST;
            const int row = st * 16 + s16;
            const bool live = m0 + row < a.M;
            const int rsw = (row >> 1) & 7;
            uint2 pk[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int ft = 2 * fp + o;
                const uint2 rr = *(const uint2*)(Rw + row * 128 + (((ft * 2 + (q4 >> 1)) ^ rsw) << 4) + 8 * (q4 & 1));
                const float r0 = __uint_as_float(rr.x << 16), r1 = __uint_as_float(rr.x & 0xffff0000u);
                const float r2 = __uint_as_float(rr.y << 16), r3 = __uint_as_float(rr.y & 0xffff0000u);
                float y0 = acc[ft][st][0], y1 = acc[ft][st][1], y2 = acc[ft][st][2], y3 = acc[ft][st][3];
                if (a.dp_thresh != 0) {
                    const uint32_t col = (uint32_t)(f0 + ft * 16 + 4 * q4);
                    const uint32_t m = (uint32_t)(m0 + row);
                    const uint32_t p0 = dropout_pair(dkey, m, (uint32_t)a.ldc, col);
                    const uint32_t p1 = dropout_pair(dkey, m, (uint32_t)a.ldc, col + 2);
                    y0 *= dropout_scale(p0, 0, a.dp_thresh, a.dp_inv_keep);
                    y1 *= dropout_scale(p0, 1, a.dp_thresh, a.dp_inv_keep);
                    y2 *= dropout_scale(p1, 0, a.dp_thresh, a.dp_inv_keep);
                    y3 *= dropout_scale(p1, 1, a.dp_thresh, a.dp_inv_keep);
                }
                // the masked gradient as the two-kernel path stores it (bf16), so both orders see the same numbers
                const uint32_t gx = cvt_pk_bf16<false>(y0, y1), gy = cvt_pk_bf16<false>(y2, y3);
                float g[4] = {__uint_as_float(gx << 16), __uint_as_float(gx & 0xffff0000u), __uint_as_float(gy << 16), __uint_as_float(gy & 0xffff0000u)};
                const float rv[4] = {r0, r1, r2, r3};
                if constexpr (PASS == 1) {
                    const int fl = fl0 + ft * 16 + 4 * q4;
                    const float4 ca = *(const float4*)(coef_s + fl), cb = *(const float4*)(coef_s + 256 + fl), cz = *(const float4*)(coef_s + 512 + fl);
                    g[0] = r0 > 0.f ? fmaf(ca.x, g[0], fmaf(cb.x, r0, cz.x)) : 0.f;
                    g[1] = r1 > 0.f ? fmaf(ca.y, g[1], fmaf(cb.y, r1, cz.y)) : 0.f;
                    g[2] = r2 > 0.f ? fmaf(ca.z, g[2], fmaf(cb.z, r2, cz.z)) : 0.f;
                    g[3] = r3 > 0.f ? fmaf(ca.w, g[3], fmaf(cb.w, r3, cz.w)) : 0.f;
                    pk[o].x = cvt_pk_bf16<false>(g[0], g[1]);
                    pk[o].y = cvt_pk_bf16<false>(g[2], g[3]);
                    g[0] = __uint_as_float(pk[o].x << 16); g[1] = __uint_as_float(pk[o].x & 0xffff0000u);
                    g[2] = __uint_as_float(pk[o].y << 16); g[3] = __uint_as_float(pk[o].y & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float w = live ? g[e] : 0.f;
                    if (st == 0) t1[o * 4 + e] = w; else t1[o * 4 + e] += w;
                    if constexpr (PASS == 0) {
                        if (st == 0) t2[o * 4 + e] = w * rv[e]; else t2[o * 4 + e] = fmaf(w, rv[e], t2[o * 4 + e]);
                    }
                }
            }
            if constexpr (PASS == 1) {
                const auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
                store_b128_settled(c, c_rsrc, c_lane, s_out + (uint32_t)(st * 16 * a.ldc + fp * 32) * 2, 0);
            }
            if (st == ST - 1) {
                qs1[fp] += row16_fold8(t1, lane);
                if constexpr (PASS == 0) qs2[fp] += row16_fold8(t2, lane);
            }
        }
#pragma unroll
        for (int st = 0; st < ST; ++st) dzf[st] = dzn[st];
    }
    if (s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
            const int f = f0 + fp * 32 + (s16 >> 2) * 16 + 4 * q4 + (s16 & 3);
            if constexpr (PASS == 0) {
                a.partials[(prow * 2 + 0) * a.F + f] = qs1[fp];
                a.partials[(prow * 2 + 1) * a.F + f] = qs2[fp];
            } else {
                a.partials[prow * a.F + f] = qs1[fp];
            }
        }
    }
}

// grid: 4 workgroups per CU (32 KiB of LDS and ~100 registers each); *stat_rows = partial rows written
template <int PASS>
static inline hipError_t launch_proj_dgrad(const GemmNTArgs& a, hipStream_t st, int* stat_rows) {
    if ((a.F & 255) || a.K < 16 || (a.K & 3) || !a.R || (PASS == 1 && !a.coef)) return hipErrorInvalidValue;
// workgroups of the projection's data-gradient launches (this one and fp8.cuh's): two per CU.  1024 / 768 / 512 / 384 / 256 on one box:
// 84-89 / 82-84 / 76-81 / 95 / 98 us (16-bit), 62-63 / 57 / 54-57 / 70 / 68 (8-bit)
#define PROJ_DGRAD_BLOCKS 512
    const int blocks = PROJ_DGRAD_BLOCKS, nwk = (blocks >> 3) / (a.F >> 8);
    const int64_t tiles = (a.M + PROJ_RT - 1) / PROJ_RT, workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
    hipLaunchKernelGGL(proj_dgrad_kernel<PASS>, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

static inline hipError_t launch_gemm_wsd_bn(const GemmNTArgs& a, hipStream_t st, int* stat_rows) {
    if (a.K != WS_K || (a.F & 255) || a.F > 768 || a.lda != WS_K || !a.R) return hipErrorInvalidValue;
    if (!a.coef) {
        // behind a dropout: mask + BatchNorm-backward sums (16x16x32 form only)
        const int nwk_ = 32 / (a.F >> 8);
        const int64_t tiles_ = (a.M + WSD16_RT - 1) / WSD16_RT, workers_ = (int64_t)nwk_ * 8;
        if (stat_rows) *stat_rows = (int)(tiles_ < workers_ ? tiles_ : workers_);
        hipLaunchKernelGGL(gemm_wsd16_kernel<1>, dim3(256), dim3(256), 0, st, a);
        return hipGetLastError();
    }
#ifdef CP_VARIANTS
    const bool m16 = !(a.dbg & 512) && !g_var.ws32 && !g_var.wsd32;          // (else the 32x32x16 form)
#else
    const bool m16 = true;
#endif
    const int nfb = a.F >> 8, nwk = 32 / nfb, rt = m16 ? WSD16_RT : WSD_RT;
    const int64_t tiles = (a.M + rt - 1) / rt;
    const int64_t workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
#ifdef CP_VARIANTS
    if (!m16) { hipLaunchKernelGGL(gemm_wsd_bn_kernel, dim3(256), dim3(256), 0, st, a); return hipGetLastError(); }
#endif
    hipLaunchKernelGGL(gemm_wsd16_kernel<0>, dim3(256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// static schedule only (a process that has the GPU to itself); *stat_rows = partial rows written
template <int EPI>
static inline hipError_t launch_gemm_ws(const GemmNTArgs& a, hipStream_t st, int* stat_rows) {
    if (a.K != WS_K || (a.F & 255) || a.F > 768 || a.lda != WS_K) return hipErrorInvalidValue;
    const int nfb = a.F >> 8, nwk = 32 / nfb;
#ifdef CP_VARIANTS
    const bool m16 = EPI == EPI_FWD && !(a.dbg & 512) && !g_var.ws32;
#else
    const bool m16 = true;
#endif
    const int rt = m16 ? WS16_RT : WS_RT;
    const int64_t tiles = (a.M + rt - 1) / rt;
    const int64_t workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
#ifndef WS_WAVES
#define WS_WAVES 4
#endif
#ifdef CP_VARIANTS
    // (dbg 512 / option ws32: the v_mfma_f32_32x32x16_bf16 form instead of the 16x16x32 one)
    if (!m16) { hipLaunchKernelGGL((gemm_ws_kernel<EPI, WS_WAVES>), dim3(256), dim3(64 * WS_WAVES), 0, st, a); return hipGetLastError(); }
#endif
    if (a.partials != nullptr) hipLaunchKernelGGL(gemm_ws16_kernel<true>, dim3(256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm_ws16_kernel<false>, dim3(256), dim3(256), 0, st, a);
    return hipGetLastError();
}

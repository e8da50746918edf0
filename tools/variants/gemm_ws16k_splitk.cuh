// Tools-only kernel (build/libcpnative_variants.so, make -C contrastiveprosthetics_amd/csrc variants): fc1's forward with the k range
// split over wave pairs -- rounds 2-3's product kernel, superseded in round 4 by gemm_ws16n_kernel (csrc/gemm_ws.cuh: one wave owns 32
// features x all 768 k, no exchange): 173-175 us against 149-157 at 167,936 rows (tools/fc1_fwd_ab.py, cp_debug_gemm dbg bit 1024).
// Included by csrc/gemm_ws.cuh under -DCP_VARIANTS only.
// fc1 forward (K = 768): 64 features x 768 k do not fit a wave's 256 weight registers, so TWO waves share a 64-feature group, each
// holding its k half (64 x 384 = 192 registers) -- a workgroup is 2 feature groups x 2 k halves = 128 features.  Both waves of a pair
// run the tile's k loop over their half of every row; each then finishes ONE of the tile's two 16-row sample tiles: it hands its 16 partial
// accumulator registers of the other sample tile to the partner through LDS in front of the tile barrier, adds the partner's for its own
// behind it, and runs that half's epilogue behind its next k loop as gemm_ws16_kernel does.  (With the whole tile finished by one wave of
// the pair in turn, the other idled at the barrier for an epilogue per tile: 186 us.)  32-row tiles: rows are 1536 bytes, a tile 48 KiB per buffer;
// LDS-DMA units of 1 KiB run across row boundaries, the 16-byte chunks are XOR-swizzled with the row inside 256-byte groups (applied to
// the DMA's per-lane source), so the 16 rows of a fragment read fall on 16 different chunk positions.  Partial rows: two per worker
// (one per k half: a wave only sums the tiles it owned).
// Measured against it: all 384 weight registers in ONE wave (256 in the accumulator file + 128 vector registers, MFMA takes its A operand
// from either; no exchange, 192 MFMAs per tile as in the K = 512 kernel) -- parity-green, but 24 spilled registers and 261 us.
__global__ __launch_bounds__(256, 1) void gemm_ws16k_kernel(GemmNTArgs a) {
    constexpr int K = WSK_K, KH = K / 2, KBH = KH / 32, RT = WSK_RT, ST = RT / 16, UPW = WSK_TILE_BYTES / 1024 / 4;
    constexpr int X_OFF = 2 * WSK_TILE_BYTES, X_BYTES = 64 * 32 * 4, BIAS_OFF = X_OFF + 2 * 2 * X_BYTES;
    static_assert(UPW == KBH, "one fetch unit per k block and wave");
    __shared__ __attribute__((aligned(16))) unsigned char smem[BIAS_OFF + 512 * 4];
    float* bias_s = (float*)(smem + BIAS_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fg = wave >> 1, kh = wave & 1;
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
    const int f0 = fb * 128 + fg * 64;

    s16x8 wreg[4][KBH];
    {
        const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)(f0 + s16) * K + kh * KH + 8 * q4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KBH; ++kb)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wreg[ft][kb]) : "v"(Wg + (int64_t)ft * 16 * K + kb * 32) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KBH; kb += 6)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(wreg[ft][kb]), "+a"(wreg[ft][kb + 1]), "+a"(wreg[ft][kb + 2]), "+a"(wreg[ft][kb + 3]),
                             "+a"(wreg[ft][kb + 4]), "+a"(wreg[ft][kb + 5]));
    }

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * WSK_ROWB), 0x00020000u};
    // fetch unit q of this wave = LDS bytes [(wave*UPW + q) * 1024, +1024) of the tile image: lane l lands on 16-byte slot g = unit*64 + l =
    // (row g / 96, physical chunk g % 96) and fetches the logical chunk that belongs there
    uint32_t fsrc[UPW];
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        const int g = (wave * UPW + q) * 64 + lane, row = g / 96, pc = g % 96;
        fsrc[q] = (uint32_t)(row * WSK_ROWB + (((pc & ~15) | ((pc ^ row) & 15)) << 4));
    }
    auto fetch_unit = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, fsrc[q], tile_soff, lds0 + buf * WSK_TILE_BYTES + (wave * UPW + q) * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };

    float qs1[2] = {0.f, 0.f}, qs2[2] = {0.f, 0.f};
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    const int foff = ((q4 & 1) << 4) | ((q4 >> 1) << 3);
    const uint32_t c_lane = (uint32_t)(s16 * a.ldc + f0 + foff) * 2;
    const int d16 = (q4 ^ s16) << 4;

    // epilogue of the wave's own sample tile (rows kh*16 + s16 of the tile), feature-tile pair fp
    auto epi_pair = [&](f32x4_t (&old)[4], int fp, uint32_t s_old, bool live) {
        float t1[8], t2[8];
        uint2 pk[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int ft = 2 * fp + o;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(old[ft][e]));
                const float w = live ? v[e] : 0.f;
                t1[o * 4 + e] = w;
                t2[o * 4 + e] = w * w;
            }
            pk[o].x = cvt_pk_bf16<false>(v[0], v[1]);
            pk[o].y = cvt_pk_bf16<false>(v[2], v[3]);
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
        const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
        store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(kh * 16 * a.ldc + fp * 32) * 2, 0);
        qs1[fp] += row16_fold8(t1, lane);
        qs2[fp] += row16_fold8(t2, lane);
    };

    // Both waves of a pair finish half of every tile: wave kh owns sample tile st == kh (rows kh*16 .. +15), sends its partial sums of the
    // OTHER sample tile to the partner through LDS in front of the tile barrier and adds the partner's for its own behind it.
    static_assert(ST == 2, "one sample tile per k half");
    // (element-wise selects: a conditional between two accumulator tiles makes hipcc index them through scratch memory)
    auto pick = [&](bool first, const f32x4_t& x, const f32x4_t& y) -> f32x4_t {
        return (f32x4_t){first ? x[0] : y[0], first ? x[1] : y[1], first ? x[2] : y[2], first ? x[3] : y[3]};
    };
    const bool k0 = kh == 0;
    f32x4_t acc[4][ST], old[4];
    bool pending = false;
    int64_t m_pending = 0;
    {
        const uint32_t soff0 = (uint32_t)(row0(0) * WSK_ROWB);
#pragma unroll
        for (int q = 0; q < UPW; ++q) fetch_unit(soff0, 0, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ti = 0; ti < ntile; ++ti) {
        const int buf = ti & 1;
        const uint32_t next_soff = ti + 1 < ntile ? (uint32_t)(row0(ti + 1) * WSK_ROWB) : 0xFFF00000u;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const float4 b4 = *(const float4*)(bias_s + f0 + ft * 16 + 4 * q4);
            const f32x4_t bv = {b4.x, b4.y, b4.z, b4.w}, zv = {0.f, 0.f, 0.f, 0.f};
            acc[ft][0] = pick(k0, bv, zv);                                   // the bias enters once: with the rows' owner
            acc[ft][1] = pick(k0, zv, bv);
        }
        const unsigned char* At = smem + buf * WSK_TILE_BYTES + s16 * WSK_ROWB + kh * (KH * 2);
        // fragments of k block kb + 1 are requested in front of the 8 MFMAs of block kb (two register sets): with only two sample tiles
        // a re-read right behind its consumer (gemm_ws16_kernel) would have 4 MFMAs = 64 cycles to return
        uint4 fa[2][ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[0][st] = *(const uint4*)(At + st * 16 * WSK_ROWB + (0 ^ d16));
#pragma unroll
        for (int kb = 0; kb < KBH; ++kb) {
#ifndef WSK_NO_FETCH
            fetch_unit(next_soff, buf ^ 1, kb);
#endif
            if (kb + 1 < KBH) {
#pragma unroll
                for (int st = 0; st < ST; ++st) fa[(kb + 1) & 1][st] = *(const uint4*)(At + st * 16 * WSK_ROWB + ((((kb + 1) * 4) << 4) ^ d16));
            }
#pragma unroll
            for (int st = 0; st < ST; ++st)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ft][kb], __builtin_bit_cast(s16x8, fa[kb & 1][st]), acc[ft][st], 0, 0, 0);
        }
        // this wave's sums of the partner's rows go to the partner: slot [feature group][buffer][receiving k half]
        unsigned char* Xo = smem + X_OFF + ((fg * 2 + buf) * 2 + (kh ^ 1)) * (X_BYTES / 2);
        const unsigned char* Xi = smem + X_OFF + ((fg * 2 + buf) * 2 + kh) * (X_BYTES / 2);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            *(f32x4_t*)(Xo + (ft * 64 + lane) * 16) = pick(k0, acc[ft][1], acc[ft][0]);
        }
        // the epilogue of the previous tile's half, behind the k loop (see gemm_ws16_kernel); then the tile barrier
        if (pending) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t s_old = (uint32_t)(m_pending * a.ldc * 2);
            epi_pair(old, 0, s_old, true);
            epi_pair(old, 1, s_old, true);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                 // the fetches are older than the 2 stores (gemm_ws16_kernel)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        pending = true;
        m_pending = row0(ti);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) old[ft] = pick(k0, acc[ft][0], acc[ft][1]) + *(const f32x4_t*)(Xi + (ft * 64 + lane) * 16);
    }
    if (pending) {                                                           // the last tile (possibly ragged)
        const uint32_t s_old = (uint32_t)(m_pending * a.ldc * 2);
        const bool live = m_pending + kh * 16 + s16 < a.M;
        epi_pair(old, 0, s_old, live);
        epi_pair(old, 1, s_old, live);
    }
    if (s16 < 8) {
        const int64_t prow = ((int64_t)wkr * 8 + xcd) * 2 + kh;
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
            const int f = f0 + fp * 32 + (s16 >> 2) * 16 + 4 * q4 + (s16 & 3);
            a.partials[(prow * 2 + 0) * a.F + f] = qs1[fp];
            a.partials[(prow * 2 + 1) * a.F + f] = qs2[fp];
        }
    }
}

static inline hipError_t launch_gemm_ws16k(const GemmNTArgs& a, hipStream_t st, int* stat_rows) {
    if (a.K != WSK_K || (a.F & 127) || a.F > 512 || a.lda != WSK_K || !a.bias) return hipErrorInvalidValue;
    const int nwk = 32 / (a.F >> 7);
    const int64_t tiles = (a.M + WSK_RT - 1) / WSK_RT, workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = 2 * (int)(tiles < workers ? tiles : workers);
    hipLaunchKernelGGL(gemm_ws16k_kernel, dim3(256), dim3(256), 0, st, a);
    return hipGetLastError();
}

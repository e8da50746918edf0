// Tools-only kernels (build/libcpnative_variants.so, make -C contrastiveprosthetics_amd/csrc variants): measured and superseded,
// kept for A/B runs (tools/ab_env.sh, tools/ws_bench.py, tools/gemm_bench.py).  Included by csrc/gemm_ws.cuh under -DCP_VARIANTS only;
// the product library does not contain them.  the 32x32x16 form of the BN-fused data gradient (superseded by gemm_wsd16_kernel<0>)

__global__ __launch_bounds__(256, 1) void gemm_wsd_bn_kernel(GemmNTArgs a) {
    constexpr int K = WS_K, KB = K / 16, RPW = WSD_RT / 4;
    constexpr int R_OFF = 2 * WSD_TILE_BYTES, COEF_OFF = R_OFF + 4 * 2 * 4096;
    __shared__ __attribute__((aligned(16))) unsigned char smem[COEF_OFF + 3 * 256 * 4];
    float* coef_s = (float*)(smem + COEF_OFF);                               // [3][256]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + WSD_RT - 1) / WSD_RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    if (ntile == 0) return;
    for (int q = tid; q < 3 * 256; q += 256) {
        const int c = q >> 8, f = fb * 256 + (q & 255);
        coef_s[q] = a.coef[c * a.coef_mod + f % a.coef_mod];
    }
    const int f0 = fb * 256 + wave * 64, fl0 = wave * 64;

    s16x8 wreg[2][KB];
    {
        const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)(f0 + r) * K + 8 * h;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wreg[i][kb]) : "v"(Wg + (int64_t)i * 32 * K + kb * 16) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kb = 0; kb < KB; kb += 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(wreg[i][kb]), "+a"(wreg[i][kb + 1]), "+a"(wreg[i][kb + 2]), "+a"(wreg[i][kb + 3]),
                             "+a"(wreg[i][kb + 4]), "+a"(wreg[i][kb + 5]), "+a"(wreg[i][kb + 6]), "+a"(wreg[i][kb + 7]));
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A, r_base = (uint64_t)(uintptr_t)a.R;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * (K * 2)), 0x00020000u};
    const u32x4_t r_rsrc = {(uint32_t)r_base, (uint32_t)(r_base >> 32) & 0xFFFFu, (uint32_t)(a.M * a.ldr * 2), 0x00020000u};
    // A: row q (0..7) of this wave's 8 rows of a tile; lane l fetches logical chunk l ^ (row & 15)
    auto fetch_a = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, (uint32_t)(((lane ^ ((wave * RPW + q) & 15)) << 4) + q * 1024), tile_soff, lds0 + buf * WSD_TILE_BYTES + (wave * RPW + q) * 1024);
    };
    // R: instruction k (0..3) = rows 8k .. 8k+7 of the tile x this wave's 64 features (128 bytes a row); lane l = row 8k + (l >> 3),
    // physical chunk l & 7 <- logical chunk (l & 7) ^ ((row >> 1) & 7)
    const uint32_t r_lane = (uint32_t)((lane >> 3) * a.ldr * 2 + f0 * 2);
    auto fetch_r = [&](uint32_t tile_soff, int buf, int k) {
        const int row = 8 * k + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        bufl16_lds(r_rsrc, r_lane + (uint32_t)(8 * k * a.ldr * 2 + lc * 16), tile_soff, lds0 + R_OFF + (wave * 2 + buf) * 4096 + k * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * WSD_RT; };

    float qs1[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) qs1[p] = 0.f;
    const bool o0 = lane & 1, o1 = (lane >> 1) & 1;
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    const uint32_t c_lane = (uint32_t)(r * a.ldc + f0 + 8 * h) * 2;
    const int d16 = (h ^ (r & 15)) << 4;
    const int rsw = (r >> 1) & 7;

    // one quad: features i*32 + 8q + 4h .. +3 of sample row r of the finished tile
    auto quad = [&](const f32x16& t, const unsigned char* Rw, int i, int q, bool live, uint2& pk) {
        const uint2 rr = *(const uint2*)(Rw + r * 128 + (((i * 4 + q) ^ rsw) << 4) + 8 * h);
        const int fl = fl0 + i * 32 + 8 * q + 4 * h;
        const float4 ca = *(const float4*)(coef_s + fl), cb = *(const float4*)(coef_s + 256 + fl), cz = *(const float4*)(coef_s + 512 + fl);
        const float r0 = __uint_as_float(rr.x << 16), r1 = __uint_as_float(rr.x & 0xffff0000u);
        const float r2 = __uint_as_float(rr.y << 16), r3 = __uint_as_float(rr.y & 0xffff0000u);
        float y0 = r0 > 0.f ? fmaf(ca.x, t[4 * q], fmaf(cb.x, r0, cz.x)) : 0.f;
        float y1 = r1 > 0.f ? fmaf(ca.y, t[4 * q + 1], fmaf(cb.y, r1, cz.y)) : 0.f;
        float y2 = r2 > 0.f ? fmaf(ca.z, t[4 * q + 2], fmaf(cb.z, r2, cz.z)) : 0.f;
        float y3 = r3 > 0.f ? fmaf(ca.w, t[4 * q + 3], fmaf(cb.w, r3, cz.w)) : 0.f;
        pk.x = cvt_pk_bf16<false>(y0, y1);
        pk.y = cvt_pk_bf16<false>(y2, y3);
        // column sums of the values as stored (the layer's bias gradient); rows past the end do not count
        float g0 = __uint_as_float(pk.x << 16), g1 = __uint_as_float(pk.x & 0xffff0000u);
        float g2 = __uint_as_float(pk.y << 16), g3 = __uint_as_float(pk.y & 0xffff0000u);
        if (!live) g0 = g1 = g2 = g3 = 0.f;
        qs1[4 * i + q] += quad_fold(g0, g1, g2, g3, o0, o1);
    };
    auto store16 = [&](const uint2& lo, const uint2& hi, uint32_t soff) {
        const auto sx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
        const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
        store_b128_settled(c, c_rsrc, c_lane, soff, 0);
    };
    // epilogue slot u = 0..11 of a finished tile: per feature half i: quads 0,1 | store kk=0 | quads 2,3 | store kk=1
    uint2 pk[2];
    auto epi_slot = [&](f32x16 (&old)[2], const unsigned char* Rw, int u, uint32_t s_old, bool live) {
        const int i = u / 6, w = u % 6, grp = w / 3, v = w % 3;
        if (v < 2) quad(old[i], Rw, i, 2 * grp + v, live, pk[v]);
        else store16(pk[0], pk[1], s_old + (uint32_t)(i * 32 + 16 * grp) * 2);
    };

    // K loop of tile ti (A buffer buf) into acc; the next tile's A rows and THIS tile's saved-activation rows are requested
    // along the way; WITH_EPI: the epilogue of the previous tile (accumulators `old`, its saved activation in R buffer buf ^ 1)
    auto step = [&](f32x16 (&acc)[2], f32x16 (&old)[2], int ti, int buf, bool has_next, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const uint32_t next_soff = has_next ? (uint32_t)((row0(ti + 1) + wave * RPW) * (K * 2)) : 0xFFF00000u;
        const uint32_t r_soff = (uint32_t)(row0(ti) * a.ldr * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
        const unsigned char* At = smem + buf * WSD_TILE_BYTES + r * 1024;
        const unsigned char* Rw = smem + R_OFF + (wave * 2 + (buf ^ 1)) * 4096;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        constexpr int PF = 2;
        uint4 fa[PF + 1];
#pragma unroll
        for (int p = 0; p < PF; ++p) fa[p] = *(const uint4*)(At + (((p * 2) << 4) ^ d16));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + PF < KB) fa[(kb + PF) % (PF + 1)] = *(const uint4*)(At + ((((kb + PF) * 2) << 4) ^ d16));
            if ((kb & 3) == 1) fetch_a(next_soff, buf ^ 1, kb >> 2);
            if ((kb & 7) == 3) fetch_r(r_soff, buf, kb >> 3);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[i][kb], __builtin_bit_cast(s16x8, fa[kb % (PF + 1)]), acc[i], 0, 0, 0);
            if constexpr (WITH_EPI)
                if (kb >= 8 && kb < 32 && (kb & 1) == 0) epi_slot(old, Rw, (kb - 8) >> 1, s_old, true);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto drain = [&](f32x16 (&old)[2], int buf, int64_t m_old) {
        const unsigned char* Rw = smem + R_OFF + (wave * 2 + buf) * 4096;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        const bool live = m_old + r < a.M;
#pragma unroll
        for (int u = 0; u < 12; ++u) epi_slot(old, Rw, u, s_old, live);
    };

    f32x16 accA[2], accB[2];
    {
        const uint32_t soff0 = (uint32_t)((row0(0) + wave * RPW) * (K * 2));
#pragma unroll
        for (int q = 0; q < RPW; ++q) fetch_a(soff0, 0, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                         // coefficient table + tile 0
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

#pragma unroll
    for (int s = 2, n = 8; n > 1; ++s, n >>= 1) {
        const bool odd = (lane >> s) & 1;
#pragma unroll
        for (int p = 0; p < n / 2; ++p) {
            const float k1 = odd ? qs1[2 * p + 1] : qs1[2 * p], g1 = odd ? qs1[2 * p] : qs1[2 * p + 1];
            qs1[p] = k1 + __shfl_xor(g1, 1 << s, 64);
        }
    }
    {
        const int f = f0 + (r >> 4) * 32 + ((r >> 2) & 3) * 8 + 4 * h + (r & 3);
        const int64_t prow = (int64_t)wkr * 8 + xcd;
        a.partials[prow * a.F + f] = qs1[0];                                 // bias gradient of the layer below: rows of F
    }
}


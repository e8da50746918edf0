// Tools-only kernels (build/libcpnative_variants.so, make -C contrastiveprosthetics_amd/csrc variants): measured and superseded,
// kept for A/B runs (tools/ab_env.sh, tools/ws_bench.py, tools/gemm_bench.py).  Included by csrc/gemm_ws.cuh under -DCP_VARIANTS only;
// the product library does not contain them.  the 32x32x16 form of the weight-stationary forward kernel (superseded by gemm_ws16_kernel)

template <int FI>
struct WsAcc {
    f32x16 t[FI][2];         // [32-feature tile i of the wave][sample half jj]
};

// WAVES = 4: one wave per SIMD, 64 features per wave (FI = 2 MFMA tiles);  WAVES = 8: two per SIMD, 32 features each
template <int EPI, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void gemm_ws_kernel(GemmNTArgs a) {
    static_assert(EPI == EPI_FWD, "forward epilogue only (so far)");
    constexpr int K = WS_K, KB = K / 16, FI = 8 / WAVES, NT = 64 * WAVES, RPW = WS_RT / WAVES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WS_TILE_BYTES + 768 * 4];
    float* bias_s = (float*)(smem + 2 * WS_TILE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + WS_RT - 1) / WS_RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    for (int q = tid; q < a.F; q += NT) bias_s[q] = a.bias[q];
    if (ntile == 0) return;
    const int f0 = fb * 256 + wave * (32 * FI);

    // ---- the wave's weights: fragment (i, kb) = rows f0 + i*32 + r, k = kb*16 + 8h .. +7 -------------------------
    // (loaded straight into ACCUMULATOR registers, "=a": MFMA reads its A operand from either half of the unified register
    //  file, VALU only from the architectural half.  Left to itself hipcc keeps these 256 registers in v0..v255, has no
    //  architectural register left for anything else and shuffles through the accumulator half with v_accvgpr copies and
    //  scratch spills; pinned there, the accumulators, fragments and epilogue live in v0..v255 and nothing is copied.)
    s16x8 wreg[FI][KB];
    {
        const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)(f0 + r) * K + 8 * h;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wreg[i][kb]) : "v"(Wg + (int64_t)i * 32 * K + kb * 16) : "memory");
        // hipcc does not know that these registers are still being written (guide 5.7 item 1): every one of them is named as
        // an operand of the wait (and of the empty statements after it), so nothing that reads them can be scheduled above it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int kb = 0; kb < KB; kb += 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(wreg[i][kb]), "+a"(wreg[i][kb + 1]), "+a"(wreg[i][kb + 2]), "+a"(wreg[i][kb + 3]),
                             "+a"(wreg[i][kb + 4]), "+a"(wreg[i][kb + 5]), "+a"(wreg[i][kb + 6]), "+a"(wreg[i][kb + 7]));
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // raw buffer descriptor of A (base, stride 0, bytes, flags as __builtin_amdgcn_make_buffer_rsrc sets them), built by
    // hand because the asm statement needs it as four plain SGPRs
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * (K * 2)), 0x00020000u};
    // row q of this wave's RPW rows of a tile (tile row wave*RPW + q): lane l fetches logical chunk l ^ (row & 15) (the swizzle, on the source side)
    auto fetch_row = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, (uint32_t)(((lane ^ ((wave * RPW + q) & 15)) << 4) + q * 1024), tile_soff, lds0 + buf * WS_TILE_BYTES + (wave * RPW + q) * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * WS_RT; };

    // BatchNorm sums of the wave's features after the two in-quad butterfly steps: lane (b1 b0) of a quad holds, in entry
    // [4i + q], the sum over its quad's samples of feature i*32 + 8q + 4h + (2 b1 + b0)
    float qs1[4 * FI], qs2[4 * FI];
#pragma unroll
    for (int p = 0; p < 4 * FI; ++p) qs1[p] = qs2[p] = 0.f;
    const bool o0 = lane & 1, o1 = (lane >> 1) & 1;

    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.ldc * 2), 0x00020000);
    const uint32_t c_lane = (uint32_t)(r * a.ldc + f0 + 8 * h) * 2;         // this lane's first chunk inside a tile: row r, features f0 + 8h
    const int d16 = (h ^ (r & 15)) << 4;                                     // per-lane part of the swizzled chunk offset

    // 4 consecutive features of one sample row -> ReLU, packed bf16; FIRST: start the tile-local sums, else add to them
    auto quad = [&](const f32x16& t, int q, bool first_jj, bool live, uint2& pk, float (&t1)[4], float (&t2)[4]) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (asm: in front of fmaxf hipcc puts a canonicalising v_max_f32 x, x -- 64 more VALU instructions per tile)
            asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(t[4 * q + e]));
            const float w = live ? v[e] : 0.f;
            if (first_jj) { t1[e] = w; t2[e] = w * w; }
            else { t1[e] += w; t2[e] = fmaf(w, w, t2[e]); }
        }
        pk.x = cvt_pk_bf16<false>(v[0], v[1]);
        pk.y = cvt_pk_bf16<false>(v[2], v[3]);
    };
    auto store16 = [&](const uint2& lo, const uint2& hi, uint32_t soff) {
        const auto sx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
        const u32x4_t c = {sx[0], sy[0], sx[1], sy[1]};
        store_b128_settled(c, c_rsrc, c_lane, soff, 0);
    };
    // epilogue slot u = 0..31 of a finished tile (rows from m_old): per feature half i sixteen slots --
    //   0,1: quad 0 of jj = 0,1 (+ fold)   2,3: quad 1   4,5: stores kk = 0   6,7: quad 2   8,9: quad 3   10,11: stores kk = 1
    uint2 pk[2][2];
    float t1[4], t2[4];
    auto epi_slot = [&](WsAcc<FI>& old, int u, uint32_t s_old, bool live0, bool live1) {
        const int i = u >> 4, v = u & 15;
        if (v >= 12 || i >= FI) return;
        const int grp = v / 6, w = v % 6;                                    // grp = kk
        if (w < 4) {
            const int q = 2 * grp + (w >> 1), jj = w & 1;
            quad(old.t[i][jj], q, jj == 0, jj ? live1 : live0, pk[jj][w >> 1], t1, t2);
            if (jj == 1) {
                qs1[4 * i + q] += quad_fold(t1[0], t1[1], t1[2], t1[3], o0, o1);
                qs2[4 * i + q] += quad_fold(t2[0], t2[1], t2[2], t2[3], o0, o1);
            }
        } else {
            const int jj = w - 4;
            store16(pk[jj][0], pk[jj][1], s_old + (uint32_t)(jj * 32 * a.ldc + i * 32 + 16 * grp) * 2);
        }
    };

    // K loop of tile `ti` (LDS buffer `buf`) into `acc`; WITH_EPI: the epilogue of the previous tile (`old`, rows from
    // m_old, all live) is woven in, one slot per k block; has_next: the next tile's 16 row fetches are spread over the loop
#ifdef WS_STAMP
    // diagnostic build only (tools/ws_bench.py): s_memtime at the start of a tile, at the end of its k loop and behind the
    // closing wait + barrier, wave 0 of block 0, into the unused tail of the partial-row buffer
    unsigned long long* stamps = (unsigned long long*)(a.partials + (size_t)200 * 2 * a.F);
    auto stamp = [&](int slot) {
        unsigned long long t, rt;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(rt) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // shader-clock cycles and the 100 MHz real-time counter side by side: their ratio is the clock the SIMD ran at
        if (blockIdx.x == 0 && tid == 0) { stamps[2 * slot] = t; stamps[2 * slot + 1] = rt; }
    };
#else
    auto stamp = [&](int) {};
#endif
    auto step = [&](WsAcc<FI>& acc, WsAcc<FI>& old, int ti, int buf, bool has_next, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        stamp(3 * ti);
        // (no next tile: an offset past the end of the buffer -- the fetches return zeros into the idle buffer -- instead of a
        //  branch in every other k block)
        const uint32_t next_soff = has_next ? (uint32_t)((row0(ti + 1) + wave * RPW) * (K * 2)) : 0xFFF00000u;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            f32x16 b0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *(const float4*)(bias_s + f0 + i * 32 + 8 * q + 4 * h);
                b0[4 * q] = b4.x; b0[4 * q + 1] = b4.y; b0[4 * q + 2] = b4.z; b0[4 * q + 3] = b4.w;
            }
            acc.t[i][0] = b0;
            acc.t[i][1] = b0;
        }
        const unsigned char* At = smem + buf * WS_TILE_BYTES + r * 1024;
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
#ifndef WS_PF
#define WS_PF 2
#endif
        constexpr int PF = WS_PF;                                            // fragment reads run PF k blocks ahead of the MFMAs
        uint4 fa[PF + 1][2];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fa[p][jj] = *(const uint4*)(At + jj * 32768 + (((p * 2) << 4) ^ d16));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + PF < KB) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) fa[(kb + PF) % (PF + 1)][jj] = *(const uint4*)(At + jj * 32768 + ((((kb + PF) * 2) << 4) ^ d16));
            }
#ifndef WS_NO_FETCH
            if (kb % (KB / RPW) == KB / RPW - 1) fetch_row(next_soff, buf ^ 1, kb / (KB / RPW));
#endif
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc.t[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[i][kb], __builtin_bit_cast(s16x8, fa[kb % (PF + 1)][jj]), acc.t[i][jj], 0, 0, 0);
#ifndef WS_NO_EPI
            if constexpr (WITH_EPI) epi_slot(old, kb, s_old, true, true);
#else
            if (kb == 31) {            // ablation: keep the finished accumulators alive so that their MFMAs are not removed
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) asm volatile("" :: "a"(old.t[i][jj]));
            }
#endif
        }
        stamp(3 * ti + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(3 * ti + 2);
    };
    // epilogue of the block's last tile (the only one that can be ragged: the stores are masked by the buffer's bounds,
    // the sums skip dead rows)
    auto drain = [&](WsAcc<FI>& old, int64_t m_old) {
        const uint32_t s_old = (uint32_t)(m_old * a.ldc * 2);
        const bool live0 = m_old + r < a.M, live1 = m_old + 32 + r < a.M;
#pragma unroll
        for (int u = 0; u < 32; ++u) epi_slot(old, u, s_old, live0, live1);
    };

    WsAcc<FI> accA, accB;
    {
        const uint32_t soff0 = (uint32_t)((row0(0) + wave * RPW) * (K * 2));
#pragma unroll
        for (int q = 0; q < RPW; ++q) fetch_row(soff0, 0, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                         // bias table + tile 0
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

    // ---- column sums: the remaining butterfly steps over lane bits 2..4 (as in gemm_nt256p.cuh).  FI = 2: 8 values, lane r of
    //      half h ends with feature (r >> 4) * 32 + ((r >> 2) & 3) * 8 + 4h + (r & 3);  FI = 1: 4 values over lane bits 2..3
    //      and a plain sum over bit 4, feature ((r >> 2) & 3) * 8 + 4h + (r & 3) in every lane
#pragma unroll
    for (int s = 2, n = 4 * FI; n > 1; ++s, n >>= 1) {
        const bool odd = (lane >> s) & 1;
#pragma unroll
        for (int p = 0; p < n / 2; ++p) {
            const float k1 = odd ? qs1[2 * p + 1] : qs1[2 * p], g1 = odd ? qs1[2 * p] : qs1[2 * p + 1];
            const float k2 = odd ? qs2[2 * p + 1] : qs2[2 * p], g2 = odd ? qs2[2 * p] : qs2[2 * p + 1];
            qs1[p] = k1 + __shfl_xor(g1, 1 << s, 64);
            qs2[p] = k2 + __shfl_xor(g2, 1 << s, 64);
        }
    }
    if constexpr (FI == 1) {
        qs1[0] += __shfl_xor(qs1[0], 16, 64);
        qs2[0] += __shfl_xor(qs2[0], 16, 64);
    }
    if (FI == 2 || (lane & 16) == 0) {
        const int f = f0 + (FI == 2 ? (r >> 4) * 32 : 0) + ((r >> 2) & 3) * 8 + 4 * h + (r & 3);
        const int64_t prow = (int64_t)wkr * 8 + xcd;
        a.partials[(prow * 2 + 0) * a.F + f] = qs1[0];
        a.partials[(prow * 2 + 1) * a.F + f] = qs2[0];
    }
}


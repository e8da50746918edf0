// Tools-only kernels (build/libcpnative_variants.so, make -C contrastiveprosthetics_amd/csrc variants): measured and superseded,
// kept for A/B runs (tools/ab_env.sh, tools/ws_bench.py, tools/gemm_bench.py).  Included by csrc/gemm_nt256.cuh under -DCP_VARIANTS only;
// the product library does not contain them.  the one-tile-per-block 256x256 NT kernel (superseded by gemm_nt256p.cuh / gemm_ws.cuh)

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmNTArgs a) {
    using T = bf16_t;
    using D = DT<T>;
    constexpr int BM = 256, BN = 256, BK = 64, EPC = 8;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;            // 32 KiB each
    constexpr int STAGE = A_BYTES + W_BYTES;
    constexpr int C_PITCH = BN * 2 + 16;
    constexpr int C_BYTES = BM * C_PITCH;                            // 135,168
    constexpr int RED_BYTES = 2 * 8 * BN * 4;                        // 16 KiB
    constexpr int LDS_BYTES = (2 * STAGE > C_BYTES + RED_BYTES) ? 2 * STAGE : (C_BYTES + RED_BYTES);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_f = a.F / BN;
    const int64_t tiles_m = (a.M + BM - 1) / BM;
    // bid = xcd + 8 * ((tile_m / 8) * tiles_f + tile_f),  tile_m = 8 * q + xcd
    const int xcd = blockIdx.x & 7;
    const int64_t j = blockIdx.x >> 3;
    const int tile_f = (int)(j % tiles_f);
    const int64_t tile_m = (j / tiles_f) * 8 + xcd;
    if (tile_m >= tiles_m) return;
    const int64_t m0 = tile_m * BM;
    const int f0 = tile_f * BN;
    const int ws = wave >> 2, wf = wave & 3;

    const T* __restrict__ Ag = (const T*)a.A;
    const T* __restrict__ Wg = (const T*)a.W;

    // per-lane source coordinates of the 4 + 4 row groups this wave stages per K-step
    const int lrow = lane >> 3, pch = lane & 7;
    const T* asrc[4];
    const T* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave + 8 * i) * 8 + lrow;                   // tile row 0..255
        const int lch = pch ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        if (m >= a.M) m = a.M - 1;                                   // clamp: such rows are never stored
        asrc[i] = Ag + m * a.lda + lch * EPC;
        wsrc[i] = Wg + (int64_t)(f0 + row) * a.K + lch * EPC;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage = [&](int buf, int kt) {
        const uint32_t As = lds0 + buf * STAGE;
        const uint32_t Ws = As + A_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = wave_u + 8 * i;
            glds16(asrc[i] + k0, As + rg * 1024);
            glds16(wsrc[i] + k0, Ws + rg * 1024);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][jj][g] = 0.f;

    const int nk = a.K / BK;
    const bool do_load = !(a.dbg & 4), do_mma = !(a.dbg & 1);
    if (do_load) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk && do_load) stage((kt + 1) & 1, kt + 1);
        const unsigned char* As = smem + (kt & 1) * STAGE;
        const unsigned char* Ws = As + A_BYTES;
        if (do_mma) {
            // fragments of sub-step ks+1 are read while the 8 MFMAs of sub-step ks run
            uint4 fw[2][2], fs[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[0][i] = *(const uint4*)(Ws + lds_tile_off(wf * 64 + i * 32 + r, h));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fs[0][jj] = *(const uint4*)(As + lds_tile_off(ws * 128 + jj * 32 + r, h));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        fw[nxt][i] = *(const uint4*)(Ws + lds_tile_off(wf * 64 + i * 32 + r, 2 * (ks + 1) + h));
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        fs[nxt][jj] = *(const uint4*)(As + lds_tile_off(ws * 128 + jj * 32 + r, 2 * (ks + 1) + h));
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) mma_chunk<T>(fw[cur][i], fs[cur][jj], acc[i][jj]);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (a.dbg & 2) {       // timing-only build of the main loop: keep the accumulators alive, store nothing
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int g = 0; g < 16; ++g) s += acc[i][jj][g];
        if (s == 12345.678f) a.partials[0] = s;
        return;
    }
    // ---- register-direct epilogue (forward, and data gradients without statistics) -------------
    // No LDS staging and no block barrier before the stores: a lane holds, per 32x32 tile, 4 runs of
    // 4 consecutive features of ONE sample row (8 bytes as bf16); v_permlane32_swap exchanges the
    // upper half-wave of run q with the lower half-wave of run q+1, after which every lane owns 8
    // consecutive features = one 16-byte store (guide T21).  The BatchNorm column sums are folded
    // over the wave's 4 sample tiles in registers (32 values per lane), then reduced across the 32
    // sample lanes with a halving butterfly (31 shuffles per statistic: at step s a lane keeps the
    // even/odd element of each pair according to bit s of its id and adds its partner's), which
    // leaves lane r with the total of value r.
    if ((EPI == EPI_FWD || (a.R == nullptr && a.dp_thresh == 0)) && !(a.dbg & 8)) {
        T* Cg = (T*)a.C;
        float ps1[32], ps2[32];
#pragma unroll
        for (int v = 0; v < 32; ++v) ps1[v] = ps2[v] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float bq[16];
            if constexpr (EPI == EPI_FWD) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *(const float4*)(a.bias + f0 + wf * 64 + i * 32 + 8 * q + 4 * h);
                    bq[4 * q] = b4.x; bq[4 * q + 1] = b4.y; bq[4 * q + 2] = b4.z; bq[4 * q + 3] = b4.w;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int64_t m = m0 + ws * 128 + jj * 32 + r;
                const bool live = m < a.M;
                uint2 pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][jj][4 * q + e];
                        if constexpr (EPI == EPI_FWD) {
                            x += bq[4 * q + e];
                            if (a.relu) x = fmaxf(x, 0.f);
                        }
                        v[e] = x;
                    }
                    pk[q] = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    if constexpr (EPI == EPI_FWD) {
                        // statistics of the values as stored (bf16-rounded), tail rows excluded
                        const float g0 = live ? __uint_as_float(pk[q].x << 16) : 0.f;
                        const float g1 = live ? __uint_as_float(pk[q].x & 0xffff0000u) : 0.f;
                        const float g2 = live ? __uint_as_float(pk[q].y << 16) : 0.f;
                        const float g3 = live ? __uint_as_float(pk[q].y & 0xffff0000u) : 0.f;
                        const int o = i * 16 + 4 * q;
                        ps1[o] += g0; ps1[o + 1] += g1; ps1[o + 2] += g2; ps1[o + 3] += g3;
                        ps2[o] = fmaf(g0, g0, ps2[o]); ps2[o + 1] = fmaf(g1, g1, ps2[o + 1]);
                        ps2[o + 2] = fmaf(g2, g2, ps2[o + 2]); ps2[o + 3] = fmaf(g3, g3, ps2[o + 3]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    uint2 lo = pk[k], hi = pk[k + 1];
                    const auto sx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
                    // lanes 0..31: features 8k..8k+7 of row r; lanes 32..63: features 8(k+1)..8(k+1)+7
                    if (live)
                        *(uint4*)(Cg + m * a.ldc + f0 + wf * 64 + i * 32 + 8 * (k + h)) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
        }
        if constexpr (EPI == EPI_FWD) {
            float* red = (float*)smem;           // [which][ws][BN]; the K loop's last barrier freed the LDS
#pragma unroll
            for (int s = 0, n = 32; s < 5; ++s, n >>= 1) {
                const bool odd = (lane >> s) & 1;
#pragma unroll
                for (int p = 0; p < n / 2; ++p) {
                    const float k1 = odd ? ps1[2 * p + 1] : ps1[2 * p], g1 = odd ? ps1[2 * p] : ps1[2 * p + 1];
                    const float k2 = odd ? ps2[2 * p + 1] : ps2[2 * p], g2 = odd ? ps2[2 * p] : ps2[2 * p + 1];
                    ps1[p] = k1 + __shfl_xor(g1, 1 << s, 64);
                    ps2[p] = k2 + __shfl_xor(g2, 1 << s, 64);
                }
            }
            // lane r of half h holds value r = i*16 + 4q + e  ->  feature wf*64 + i*32 + 8q + 4h + e
            const int fl = wf * 64 + (r >> 4) * 32 + ((r >> 2) & 3) * 8 + 4 * h + (r & 3);
            red[(0 * 2 + ws) * BN + fl] = ps1[0];
            red[(1 * 2 + ws) * BN + fl] = ps2[0];
            __syncthreads();
            const int which = tid / BN, col = tid % BN;
            a.partials[(tile_m * 2 + which) * a.F + f0 + col] = red[(which * 2 + 0) * BN + col] + red[(which * 2 + 1) * BN + col];
        }
        return;
    }
    // ---- staged epilogue (data gradients that also reduce statistics against the saved activation):
    //      accumulators -> bf16 tile in LDS -> 16-byte row segments to HBM ------------
    unsigned char* Cs = smem;
    float* red = (float*)(smem + C_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int srow = ws * 128 + jj * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fl = wf * 64 + i * 32 + 8 * q + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][jj][4 * q + e];
                    if constexpr (EPI == EPI_FWD) {
                        x += a.bias[f0 + fl + e];
                        if (a.relu) x = fmaxf(x, 0.f);
                    }
                    v[e] = x;
                }
                *(uint2*)(Cs + srow * C_PITCH + fl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
    __syncthreads();
    constexpr int CPR = BN / EPC;          // 32 chunks per row
    constexpr int RPP = 512 / CPR;         // 16 rows per pass
    const int cc = tid % CPR, rr = tid / CPR;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    T* Cg = (T*)a.C;
    const T* Rg = (const T*)a.R;
    // the saved-activation chunks of all 16 passes are independent loads: issue them together (the
    // accumulators are dead here, registers are plentiful) instead of one dependent load per pass
    // (R == nullptr: the caller derives the BN-backward sums from the weight gradient instead --
    //  bn_bwd_sums_from_wgrad_kernel -- and this launch neither reads R nor reduces anything)
    const bool with_stats = (EPI == EPI_FWD) || (a.R != nullptr);
    // a.coef: this launch produces the gradient that ENTERS the layer below and the layer's BN-backward coefficients
    // are already known (its sums were derived from the weight gradient, bn_bwd_sums_from_wgrad_kernel): apply
    // BatchNorm backward (step 2) and the ReLU mask here, on the tile that is in LDS anyway, against the saved
    // activation that this epilogue already knows how to fetch -- instead of a separate in-place pass over the
    // gradient (read 2 x N x F, write N x F).  The column sums of the result are the layer's bias gradient.
    const bool bnrelu = (EPI == EPI_DGRAD) && a.coef != nullptr && a.R != nullptr;
    float kca[EPC], kcb[EPC], kcz[EPC];
    if (bnrelu) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int ch = (f0 + cc * EPC + e) % a.coef_mod;
            kca[e] = a.coef[ch]; kcb[e] = a.coef[a.coef_mod + ch]; kcz[e] = a.coef[2 * a.coef_mod + ch];
        }
    }
    uint4 rpre[BM / RPP];
    if (EPI == EPI_DGRAD && with_stats) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            int64_t m = m0 + rr + p * RPP;
            if (m >= a.M) m = a.M - 1;
            rpre[p] = *(const uint4*)(Rg + m * a.ldr + f0 + cc * EPC);
        }
    }
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        const int row = rr + p * RPP;
        const int64_t m = m0 + row;
        if (m < a.M) {
            uint4 c = *(const uint4*)(Cs + row * C_PITCH + cc * 16);
            float v[EPC];
            D::unpack(c, v);
            const int f = f0 + cc * EPC;
            if constexpr (EPI == EPI_FWD) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
            } else {
                float rv[EPC];
                const uint4 rc = with_stats ? rpre[p] : make_uint4(0, 0, 0, 0);
                D::unpack(rc, rv);
                if (bnrelu) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = rv[e] > 0.f ? fmaf(kca[e], v[e], fmaf(kcb[e], rv[e], kcz[e])) : 0.f;
                    c = D::pack(v);
                }
                if (a.dp_thresh != 0) {
#pragma unroll
                    for (int e = 0; e < EPC; e += 2) {
                        const uint32_t pr = dropout_pair(a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key, (uint32_t)m, (uint32_t)a.ldc, (uint32_t)(f + e));
                        v[e] *= dropout_scale(pr, 0, a.dp_thresh, a.dp_inv_keep);
                        v[e + 1] *= dropout_scale(pr, 1, a.dp_thresh, a.dp_inv_keep);
                    }
                    c = D::pack(v);
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float g = D::round(v[e]);
                    s1[e] += g;
                    s2[e] = fmaf(g, rv[e], s2[e]);
                }
            }
            *(uint4*)(Cg + m * a.ldc + f) = c;
        }
    }
    if (!with_stats) return;
    // lanes l and l+32 of a wave own the same 8 columns: combine, then one row of sums per wave
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        s1[e] += __shfl_xor(s1[e], 32, 64);
        s2[e] += __shfl_xor(s2[e], 32, 64);
    }
    if (lane < 32) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[(0 * 8 + wave) * BN + cc * EPC + e] = s1[e];
            red[(1 * 8 + wave) * BN + cc * EPC + e] = s2[e];
        }
    }
    __syncthreads();
    {
        const int which = tid / BN, col = tid % BN;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[(which * 8 + q) * BN + col];
        if (bnrelu) {
            if (which == 0) a.partials[tile_m * a.F + f0 + col] = s;         // bias gradient: rows of F
        } else {
            a.partials[(tile_m * 2 + which) * a.F + f0 + col] = s;
        }
    }
}

template <int EPI>
static inline hipError_t launch_gemm_nt256(const GemmNTArgs& a, hipStream_t st) {
    const int64_t tiles_m = (a.M + 255) / 256;
    const int64_t groups = (tiles_m + 7) / 8;
    const int64_t blocks = groups * 8 * (a.F / 256);
    hipLaunchKernelGGL((gemm_nt256_kernel<EPI>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    return hipGetLastError();
}


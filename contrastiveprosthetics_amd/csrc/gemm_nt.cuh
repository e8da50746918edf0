// "NT" GEMM of the encoder:  C[m][f] = sum_k A[m][k] * W[f][k]
//   m = sample rows (windows, or (window,position) pairs for conv2)  -- huge
//   f = output features, k = input features, both K-contiguous in memory.
// One kernel template serves: fc forward (bias+ReLU+BN-statistics epilogue),
// fc / proj / conv2 data-gradient (dropout mask + BN-backward statistics epilogue),
// conv2 forward (3-tap sliding-window A loader with the BN affine applied in the
// staging registers), and the 512->16 projection (plain f32 epilogue).
//
// MFMA orientation: the weight tile is the MFMA "A" operand (rows = features) and the
// sample tile the "B" operand (columns = samples), so that each lane ends up with four
// CONSECUTIVE features of one sample per accumulator quad: the epilogue packs them into
// one 8/16-byte LDS write, and the tile leaves the block as full 16-byte row segments.
#pragma once
#include "common.cuh"

enum { ALOAD_PLAIN = 0, ALOAD_CONV = 1,
       ALOAD_BNDROP = 2,     // A = dropout(BatchNorm(saved activation)) formed while staging (a_scale/a_shift + the dp_* fields; K = row width)
       ALOAD_F8 = 3,         // A is stored as e4m3 (one byte per element, lda in bytes) with the scale 2^*a_exp: converted to T while staging
       ALOAD_BNDROP_F8 = 4 };// both
enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_PLAIN_F32 = 2,
       EPI_DGRAD_BN = 3,   // persistent kernel: data gradient + BN/ReLU backward of the layer below (GemmNTArgs::coef) against R
       EPI_DGRAD_ST = 4 }; // persistent kernel: data gradient (+ dropout mask) + BN-backward sums against R

struct GemmNTArgs {
    const void* A;       // [M][lda] T
    const void* W;       // [F][K] T   (F multiple of BN)
    void* C;             // [M][ldc] T   (EPI_PLAIN_F32: float, ldc floats)
    const float* bias;   // [F] (EPI_FWD) or nullptr
    const void* R;       // [M][ldr] T  saved post-ReLU activation (EPI_DGRAD)
    float* partials;     // [tiles_m][2][F] per-block column sums (EPI_FWD: v, v^2; EPI_DGRAD: g, g*r)
    const float* coef;   // EPI_DGRAD (256-tile staged kernel) with R: [3][coef_mod] BN-backward coefficients of the layer BELOW;
    int coef_mod;        //   the epilogue then writes  r > 0 ? ca*g + cb*r + cz : 0  (feature f uses entry f % coef_mod)
    const float* a_scale;  // ALOAD_CONV affine per input channel (64) or nullptr
    const float* a_shift;
    const int* a_exp;      // ALOAD_F8 / ALOAD_BNDROP_F8: device word holding the scale exponent of A (stored = value * 2^e)
    int64_t M;
    int lda, ldc, ldr;
    int K, F;
    int relu;            // EPI_FWD: apply ReLU
    int f_valid;         // EPI_PLAIN_F32: store only features < f_valid
    int dbg;             // ablation (tools/gemm_bench.py only): 1 skip MFMA, 2 skip epilogue, 4 skip staging loads
    // dropout on the gradient (EPI_DGRAD); thresh == 0 -> none
    uint32_t dp_thresh, dp_key;
    const uint32_t* dp_salt;   // optional device word XOR-ed into dp_key (graph replay: the per-step part of the key)
    float dp_inv_keep;
    int* sched;          // persistent kernel: its stream's tile counters (per XCD, and blocks finished; 32 ints apart); set by the launcher
};

template <typename T, int BM, int BN, int ALOAD, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNTArgs a) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    constexpr int BK = D::BK;
    constexpr int WAVES_F = (BN >= 128) ? 2 : 1;
    constexpr int WAVES_S = 4 / WAVES_F;
    constexpr int WS_T = BM / (WAVES_S * 32);        // sample tiles per wave
    constexpr int WF_T = BN / (WAVES_F * 32);        // feature tiles per wave
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int C_PITCH = BN * (int)sizeof(T) + 16;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + W_BYTES);
    constexpr int C_BYTES = (EPI == EPI_PLAIN_F32) ? 0 : BM * C_PITCH;
    constexpr int RED_BYTES = (EPI == EPI_PLAIN_F32) ? 0 : 2 * 256 * EPC * 4;
    constexpr int LDS_BYTES = (STAGE_BYTES > C_BYTES + RED_BYTES) ? STAGE_BYTES : (C_BYTES + RED_BYTES);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_f = a.F / BN;
    const int tile_f = blockIdx.x % tiles_f;
    const int64_t tile_m = blockIdx.x / tiles_f;
    const int64_t m0 = tile_m * BM;
    const int f0 = tile_f * BN;
    const int ws = wave / WAVES_F, wf = wave % WAVES_F;

    const T* __restrict__ Ag = (const T*)a.A;
    const T* __restrict__ Wg = (const T*)a.W;

    // ---- staging: each thread moves chunk (tid&7) of rows (tid>>3)+32*i -----------
    constexpr int A_IT = BM / 32, W_IT = BN / 32;
    uint4 areg[A_IT], wreg[W_IT];
    const int sc = tid & 7, sr = tid >> 3;
    float a_deq = 1.f;
    if constexpr (ALOAD == ALOAD_F8 || ALOAD == ALOAD_BNDROP_F8) a_deq = f8_exp2i(-*a.a_exp);

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int row = sr + 32 * i;
            wreg[i] = *(const uint4*)(Wg + (int64_t)(f0 + row) * a.K + k0 + sc * EPC);
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int64_t m = m0 + sr + 32 * i;
            if constexpr (ALOAD == ALOAD_PLAIN || ALOAD == ALOAD_BNDROP || ALOAD == ALOAD_F8 || ALOAD == ALOAD_BNDROP_F8) {
                const int64_t mc = m < a.M ? m : a.M - 1;
                uint4 v;
                if constexpr (ALOAD == ALOAD_F8 || ALOAD == ALOAD_BNDROP_F8) {
                    static_assert(sizeof(T) == 2 || (ALOAD != ALOAD_F8 && ALOAD != ALOAD_BNDROP_F8), "e4m3 operands feed the bf16 kernels");
                    v = f8_chunk_to_bf16(*(const uint2*)((const uint8_t*)a.A + mc * a.lda + k0 + sc * 8), a_deq);
                } else {
                    v = *(const uint4*)(Ag + mc * a.lda + k0 + sc * EPC);
                }
                if constexpr (ALOAD == ALOAD_BNDROP || ALOAD == ALOAD_BNDROP_F8)
                    v = bn_drop_chunk<T>(v, a.a_scale, a.a_shift, k0 + sc * EPC, a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key, (uint32_t)mc,
                                         (uint32_t)a.K, a.dp_thresh, a.dp_inv_keep);
                if (m >= a.M) v = make_uint4(0, 0, 0, 0);
                areg[i] = v;
            } else {
                // row m = (window, position w); k0 selects tap = k0/64 and channel offset
                const int tap = k0 >> 6, ch0 = (k0 & 63) + sc * EPC;
                const int w = (int)(m % 12) + tap - 1;
                const bool ok = (m < a.M) && (w >= 0) && (w < 12);
                const int64_t ms = ok ? (m + tap - 1) : 0;
                uint4 v = *(const uint4*)(Ag + ms * a.lda + ch0);
                if (a.a_scale != nullptr) {
                    float x[EPC];
                    D::unpack(v, x);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) x[e] = fmaf(x[e], a.a_scale[ch0 + e], a.a_shift[ch0 + e]);
                    v = D::pack(x);
                }
                if (!ok) v = make_uint4(0, 0, 0, 0);
                areg[i] = v;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* As = smem + buf * (A_BYTES + W_BYTES);
        unsigned char* Ws = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *(uint4*)(As + lds_tile_off(sr + 32 * i, sc)) = areg[i];
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *(uint4*)(Ws + lds_tile_off(sr + 32 * i, sc)) = wreg[i];
    };

    f32x16 acc[WF_T][WS_T];
#pragma unroll
    for (int i = 0; i < WF_T; ++i)
#pragma unroll
        for (int j = 0; j < WS_T; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    const int nk = a.K / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles(kt + 1);
        const unsigned char* As = smem + (kt & 1) * (A_BYTES + W_BYTES);
        const unsigned char* Ws = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 fw[WF_T], fs[WS_T];
#pragma unroll
            for (int i = 0; i < WF_T; ++i)
                fw[i] = *(const uint4*)(Ws + lds_tile_off((wf * WF_T + i) * 32 + r, 2 * ks + h));
#pragma unroll
            for (int j = 0; j < WS_T; ++j)
                fs[j] = *(const uint4*)(As + lds_tile_off((ws * WS_T + j) * 32 + r, 2 * ks + h));
#pragma unroll
            for (int i = 0; i < WF_T; ++i)
#pragma unroll
                for (int j = 0; j < WS_T; ++j) mma_chunk<T>(fw[i], fs[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tiles((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------
    if constexpr (EPI == EPI_PLAIN_F32) {
        float* Cg = (float*)a.C;
#pragma unroll
        for (int i = 0; i < WF_T; ++i)
#pragma unroll
            for (int j = 0; j < WS_T; ++j) {
                const int64_t m = m0 + (ws * WS_T + j) * 32 + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = f0 + (wf * WF_T + i) * 32 + 8 * q + 4 * h;
                    if (m < a.M && f < a.f_valid) {
                        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                        if (a.bias != nullptr) { v.x += a.bias[f]; v.y += a.bias[f + 1]; v.z += a.bias[f + 2]; v.w += a.bias[f + 3]; }
                        *(float4*)(Cg + m * a.ldc + f) = v;
                    }
                }
            }
        return;
    } else {
        unsigned char* Cs = smem;                       // staging buffers are dead now
        float* red = (float*)(smem + C_BYTES);
#pragma unroll
        for (int i = 0; i < WF_T; ++i)
#pragma unroll
            for (int j = 0; j < WS_T; ++j) {
                const int srow = (ws * WS_T + j) * 32 + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int fl = (wf * WF_T + i) * 32 + 8 * q + 4 * h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][4 * q + e];
                        if constexpr (EPI == EPI_FWD) {
                            x += a.bias[f0 + fl + e];
                            if (a.relu) x = fmaxf(x, 0.f);
                        }
                        v[e] = x;
                    }
                    unsigned char* dst = Cs + srow * C_PITCH + fl * (int)sizeof(T);
                    if constexpr (sizeof(T) == 2) {
                        *(uint2*)dst = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    } else {
                        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        __syncthreads();
        // store-out: thread owns chunk column cc, walks rows; 16-byte coalesced stores
        constexpr int CPR = BN / EPC;                   // chunks per row
        constexpr int RPP = 256 / CPR;                  // rows per pass
        const int cc = tid % CPR, rr = tid / CPR;
        float s1[EPC], s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
        T* Cg = (T*)a.C;
        const T* Rg = (const T*)a.R;
#pragma unroll 2
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
                    uint4 rc = make_uint4(0, 0, 0, 0);
                    if (a.R != nullptr) rc = *(const uint4*)(Rg + m * a.ldr + f);   // nullptr: sums come from the weight gradient
                    D::unpack(rc, rv);
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
        if (EPI == EPI_DGRAD && a.R == nullptr) return;
        // block column sums: red[which][rr][col]
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[(0 * RPP + rr) * BN + cc * EPC + e] = s1[e];
            red[(1 * RPP + rr) * BN + cc * EPC + e] = s2[e];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < RPP; ++q) s += red[(which * RPP + q) * BN + col];
            a.partials[(tile_m * 2 + which) * a.F + f0 + col] = s;
        }
    }
}

template <typename T, int BM, int BN, int ALOAD, int EPI>
static inline hipError_t launch_gemm_nt(const GemmNTArgs& a, hipStream_t st) {
    const int64_t tiles_m = (a.M + BM - 1) / BM;
    const int64_t blocks = tiles_m * (a.F / BN);
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, ALOAD, EPI>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

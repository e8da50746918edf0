// Conv front-end (code/models.py:255-261 on (N,1,1,12) input, so only kernel row 1 of each 3x3
// touches data).  conv1's input is 12 floats per window, so its 768-element output r1 is NEVER
// stored: every consumer (conv2 forward, conv2 weight gradient, the BN-backward sums of conv2's
// data gradient, conv1's own backward) recomputes  r1 = round_T(relu(b + sum_tap W[tap] x[w+tap-1]))
// from x while it stages its LDS image.  That removes one 258 MB write and four 258 MB reads per
// step at 167,936 windows.
//
// conv2 is an implicit GEMM over "strips" of 16 windows: the strip's input is held in LDS as an
// image of 14 rows per window (zero row, 12 positions, zero row) x 64 channels, so the three taps
// are the same image read at row offsets 0/1/2 and the zero padding is real zeros.  Blocks are
// persistent (weights stay in LDS, BN partial sums stay in registers across strips).
#pragma once
#include <type_traits>
#include "common.cuh"
#include "gemm_tn.cuh"

#define CONV_WPB 16                      // windows per strip
#define CONV_ROWS (CONV_WPB * 12)        // 192 output rows per strip
#define CONV_IMG_ROWS (CONV_WPB * 14)    // 224 image rows per strip

struct ConvArgs {
    const float* x;         // [N][12]
    const float* w1;        // conv_emg.0.weight (64,1,3,3)
    const float* b1;        // conv_emg.0.bias (64)
    const float* stats1;    // BN1 [4][64] mean, invstd, scale, shift (nullptr: image holds raw r1)
    const void* wc;         // conv2 weights in compute dtype [64][192] (k = tap*64 + in-channel)
    const float* bias2;     // conv_emg.3.bias (forward)
    const void* gin;        // [N*12][64] T gradient wrt conv2's pre-BN output (dgrad / wgrad); kernels with G8: e5m2 bytes, stored = value * 2^*gin_exp
    const int* gin_exp;
    const float* coef;      // conv2_dgrad_conv1_kernel: BatchNorm1-backward coefficients [3][64] (bn_bwd_finalize_kernel)
    void* out;              // forward: r2, dgrad: g_v1   [N*12][64] T
    uint8_t* out8;          // forward, CP_FP8: r2 as e4m3 [N*12][64] INSTEAD of `out`, scale 2^*out_exp, running maximum in *out_amax
    const int* out_exp;
    uint32_t* out_amax;
    float* partials;        // [grid][2][64] (fwd: sum, sumsq of r2; dgrad: sum g, sum g*r1) / wgrad: slabs [grid][64][192]
    int64_t n_windows;
    // small batches (small.cuh), forward only: BatchNorm1 finalised HERE from conv1_stats_kernel's fixed-point totals (bn1.acc != nullptr
    // replaces stats1; block 0 stores the statistics) and the output's totals added to acc_out instead of partial rows: no finalize launches
    SmBN bn1;
    long long* acc_out;     // [2][64] or nullptr
    // small batches, backward (conv_backward_tail): the consumer finalises, no launches in between.
    //   conv2_wgrad_kernel<T, false, true>: BatchNorm2 + ReLU backward applied to the gradient while it is staged (and written back
    //   in place for the data-gradient launch), its coefficients formed in the prologue from fc1's partial rows
    const float* bn2_rows;  // [bn2_nr][2][768] sums of (g, g r2) per (position, channel)
    int bn2_nr;
    const void* r2;         // [N*12][64] T conv2's saved output
    const float* stats2;    // [4][64]
    float* dgamma2;         // written by block 0
    float* dbeta2;
    float* gcols3;          // [grid][3][64]: column sums of the transformed gradient over all positions / position 0 / position 11
    //   conv2_dgrad_conv1_kernel<T, false, true>: BatchNorm1's backward coefficients from conv2_wgrad_finish_kernel's rows (stats1 = its table)
    const float* rows1;     // [rows1_nr][2][64]
    int rows1_nr;
    float* dgamma1;
    float* dbeta1;
};

// coefficients of  g_y = [r > 0] (ca g + cb r + cz)  for 64 channels from `nr` partial rows [2][nfold * 64] of (sum g, sum g r)
// (bn_bwd_finalize_kernel's arithmetic; 256 threads, every block of a launch computes the same numbers; `writer` stores dgamma / dbeta)
__device__ __forceinline__ void conv_bn_bwd_coef(const float* __restrict__ rows, int nr, int nfold, const float* __restrict__ stats, double count,
                                                 float* ca, float* cb, float* cz, double (*red)[4][64], bool writer, float* dgamma, float* dbeta) {
    const int tid = threadIdx.x, c = tid & 63, part = tid >> 6, W = nfold * 64;
    double s1 = 0, s2 = 0;
    const int total = nr * nfold;
    int i = part;
    for (; i + 28 < total; i += 32) {                  // eight (row, position) pairs of both sums in flight: this walk is pure latency
        float v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = i + 4 * u, rw = k / nfold, f = k % nfold;
            v1[u] = rows[((int64_t)rw * 2 + 0) * W + f * 64 + c];
            v2[u] = rows[((int64_t)rw * 2 + 1) * W + f * 64 + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
    }
    for (; i < total; i += 4) {
        const int rw = i / nfold, f = i % nfold;
        s1 += (double)rows[((int64_t)rw * 2 + 0) * W + f * 64 + c];
        s2 += (double)rows[((int64_t)rw * 2 + 1) * W + f * 64 + c];
    }
    red[0][part][c] = s1;
    red[1][part][c] = s2;
    __syncthreads();
    if (tid < 64) {
        s1 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        s2 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        const double mean = stats[c], invstd = stats[64 + c], sc = stats[128 + c];
        const double dot = (s2 - mean * s1) * invstd;
        const double c1 = s1 / count, c2 = dot / count;
        ca[c] = (float)sc;
        cb[c] = (float)(-sc * invstd * c2);
        cz[c] = (float)(-sc * (c1 - mean * invstd * c2));
        if (writer) { dgamma[c] = (float)dot; dbeta[c] = (float)s1; }
    }
    __syncthreads();
}

template <typename T> struct ConvGeo {
    using D = DT<T>;
    static constexpr int EPC = D::EPC;
    static constexpr int CPR = 64 / EPC;                 // 16-byte chunks per 64-channel row
    static constexpr int ROWB = 64 * (int)sizeof(T);     // bytes per image row
    static constexpr int RPP = 256 / CPR;                // rows handled per pass by 256 threads
    static constexpr int WPITCH = 192 * (int)sizeof(T) + 16;
    static constexpr int CPITCH = ROWB + 16;
    // The 16 lanes of a ds_read_b128 group read output rows that are distinct mod 16, but image rows
    // skip 2 at every window boundary, so the swizzle is keyed on the "dense" row index
    // d = row - 2*(row/14) (= output row + tap for every row a fragment read touches; d and row have
    // the same parity, which is what selects the bank half of a 128-byte row).
    __device__ static __forceinline__ int swz(int row) {
        const int d = row - 2 * (row / 14);
        return sizeof(T) == 2 ? ((d >> 1) & 7) : (d & 15);
    }
    __device__ static __forceinline__ int img_off(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};

// conv1 for one (window, position) and the EPC channels of one chunk: value exactly as stored
// activations are rounded (stats, BN-backward sums and conv2's input all see the same number)
// xr = the window's 12 input values (global memory, or the strip's copy in LDS)
// from the three input values under the taps (zero outside the window)
template <typename T>
__device__ __forceinline__ void conv1_chunk_vals(float xm, float x0, float xp, const float (*wt)[3], const float* bs, float* out) {
    using D = DT<T>;
    // channel pairs on the packed f32 pipe (v_pk_fma_f32: these kernels are bound by VALU issue, 4 cycles per wave64
    // instruction); bf16: round first, ReLU on the packed pair (v_pk_max_i16) -- the same value as round(relu(y))
    const f32x2_t xm2 = {xm, xm}, x02 = {x0, x0}, xp2 = {xp, xp};
#pragma unroll
    for (int e = 0; e < D::EPC; e += 2) {
        f32x2_t y = {bs[e], bs[e + 1]};
        y = __builtin_elementwise_fma((f32x2_t){wt[e][0], wt[e + 1][0]}, xm2, y);
        y = __builtin_elementwise_fma((f32x2_t){wt[e][1], wt[e + 1][1]}, x02, y);
        y = __builtin_elementwise_fma((f32x2_t){wt[e][2], wt[e + 1][2]}, xp2, y);
        if constexpr (sizeof(T) == 2) {
            const uint32_t pk = cvt_pk_bf16<true>(y.x, y.y);
            out[e] = __uint_as_float(pk << 16);
            out[e + 1] = __uint_as_float(pk & 0xffff0000u);
        } else {
            out[e] = fmaxf(y.x, 0.f);
            out[e + 1] = fmaxf(y.y, 0.f);
        }
    }
}
template <typename T>
__device__ __forceinline__ void conv1_chunk_row(const float* xr, int w, const float (*wt)[3], const float* bs, float* out) {
    // unconditional loads at clamped positions + selects: no divergent branch around a load, so the
    // loads of several unrolled callers are issued back to back
    const float x0 = xr[w];
    const float xl = xr[w > 0 ? w - 1 : 0];
    const float xh = xr[w < 11 ? w + 1 : 11];
    conv1_chunk_vals<T>(w > 0 ? xl : 0.f, x0, w < 11 ? xh : 0.f, wt, bs, out);
}
template <typename T>
__device__ __forceinline__ void conv1_chunk(const float* __restrict__ x, int64_t win, int w, const float (*wt)[3],
                                            const float* bs, float* out) {
    conv1_chunk_row<T>(x + win * 12, w, wt, bs, out);
}

// ------------------------------------------------------------------------------------------
// conv1 forward: BatchNorm2d statistics of r1 only (nothing is stored)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1_stats_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ partials,
                                                          int64_t rows, long long* __restrict__ acc = nullptr) {
    using G = ConvGeo<T>;
    constexpr int EPC = G::EPC, CPR = G::CPR, RPP = G::RPP;
    __shared__ float red[2][RPP][64];
    const int tid = threadIdx.x, cc = tid % CPR, rr = tid / CPR;
    float wt[EPC][3], bs[EPC];
    f32x2_t s1[EPC / 2], s2[EPC / 2];                  // channel pairs (packed f32 adds / fmas)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int k = 0; k < 3; ++k) wt[e][k] = w[(cc * EPC + e) * 9 + 3 + k];
        bs[e] = bias[cc * EPC + e];
    }
#pragma unroll
    for (int k = 0; k < EPC / 2; ++k) s1[k] = s2[k] = (f32x2_t){0.f, 0.f};
    // one window per thread and pass: its 12 inputs come as three 16-byte loads and the 12 positions are unrolled, so there
    // is no per-row index arithmetic (a 64-bit divide by 12) and no clamped neighbour loads (45 -> 22 us at 167,936 windows)
    const int64_t nwin = rows / 12;
    for (int64_t win = (int64_t)blockIdx.x * RPP + rr; win < nwin; win += (int64_t)gridDim.x * RPP) {
        float xv[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 t = *(const float4*)(x + win * 12 + 4 * q);
            xv[4 * q] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int w = 0; w < 12; ++w) {
            float v[EPC];
            conv1_chunk_vals<T>(w > 0 ? xv[w - 1] : 0.f, xv[w], w < 11 ? xv[w + 1] : 0.f, wt, bs, v);
#pragma unroll
            for (int k = 0; k < EPC / 2; ++k) {
                const f32x2_t vp = {v[2 * k], v[2 * k + 1]};
                s1[k] += vp;
                s2[k] = __builtin_elementwise_fma(vp, vp, s2[k]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[0][rr][cc * EPC + e] = s1[e / 2][e & 1]; red[1][rr][cc * EPC + e] = s2[e / 2][e & 1]; }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float s = 0.f;
        for (int q = 0; q < RPP; ++q) s += red[which][q][c];
        if (acc != nullptr) sm_acc_add(acc + which * 64 + c, s, SM_ACT_SHIFT);       // (small batches: order-independent integer totals)
        else partials[((int64_t)blockIdx.x * 2 + which) * 64 + c] = s;
    }
}

// debug/test only: materialise r1 [N*12][64] as f32
template <typename T>
__global__ __launch_bounds__(256) void conv1_materialize_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int64_t rows) {
    using G = ConvGeo<T>;
    constexpr int EPC = G::EPC, CPR = G::CPR, RPP = G::RPP;
    const int tid = threadIdx.x, cc = tid % CPR, rr = tid / CPR;
    float wt[EPC][3], bs[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int k = 0; k < 3; ++k) wt[e][k] = w[(cc * EPC + e) * 9 + 3 + k];
        bs[e] = bias[cc * EPC + e];
    }
    for (int64_t m = (int64_t)blockIdx.x * RPP + rr; m < rows; m += (int64_t)gridDim.x * RPP) {
        float v[EPC];
        conv1_chunk<T>(x, m / 12, (int)(m % 12), wt, bs, v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) out[m * 64 + cc * EPC + e] = v[e];
    }
}

// ------------------------------------------------------------------------------------------
// conv2 forward (MODE 0) and conv2 data gradient (MODE 1), persistent strip kernel.
//   MODE 0: image = BN1(r1) recomputed from x;  out = relu(conv + bias2);  sums: out, out^2
//   MODE 1: image = g_y2 strip from HBM;        out = g_v1;                sums: g, g * r1(recomputed)
// 4 waves: wave w computes feature tile (w&1) x sample tiles 3*(w>>1)+{0,1,2} of the 192 x 64 strip.
// ------------------------------------------------------------------------------------------
// LDS: image/output region + weights = 54 KB in bf16 (the final block reduction reuses the region).  Two blocks per CU:
// the kernel needs 198-234 VGPRs; capping it at 168 for a third block (__launch_bounds__(256, 3)) spilled 30-61
// registers and ran 2x slower (0.39 / 0.32 ms against 0.15 / 0.18 ms for forward / data gradient at 167,936 windows).
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void conv2_strip_kernel(ConvArgs a) {
    using D = DT<T>;
    using G = ConvGeo<T>;
    constexpr int EPC = G::EPC, CPR = G::CPR, ROWB = G::ROWB, RPP = G::RPP, WPITCH = G::WPITCH, CPITCH = G::CPITCH;
    constexpr int KSTEP = D::KSTEP, KS_PER_TAP = 64 / KSTEP, NKS = 3 * KS_PER_TAP;
    constexpr int IMG_BYTES = CONV_IMG_ROWS * ROWB, C_BYTES = CONV_ROWS * CPITCH;
    constexpr int REGION = IMG_BYTES > C_BYTES ? IMG_BYTES : C_BYTES;
    constexpr int W_BYTES = 64 * WPITCH;
    static_assert(REGION >= 2 * RPP * 64 * 4, "the end-of-kernel reduction reuses the image region");
    __shared__ __attribute__((aligned(16))) unsigned char smem[REGION + W_BYTES];
    // MODE 0: the strip's raw input (16 windows x 12 values), double-buffered: the next strip's copy is requested at the
    // start of a strip and parked here behind its MFMAs, so the image is built from LDS (per-row global loads of x sat,
    // three at a time, in front of every image row's arithmetic: 153 -> 137 us).  MODE 1 recomputes r1 in its store-out
    // loop, where the same change cost 38 us (128 -> 166): it keeps reading x from global memory.
    __shared__ float xs[MODE == 0 ? 2 : 1][MODE == 0 ? CONV_WPB * 12 : 1];
    // CP_FP8 forward: scale of the e4m3 output and the largest scaled value this thread stored
    float out_scale = 1.f, out_max = 0.f;
    if constexpr (MODE == 0 && sizeof(T) == 2) {
        if (a.out8 != nullptr) {
            out_scale = f8_exp2i(*a.out_exp);
            __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // MODE.FP16_OVFL: the e4m3 conversions below saturate (fp8.cuh, f8_saturating_conversions)
        }
    }
    unsigned char* img = smem;
    unsigned char* Cs = smem;                      // aliases the image once the MFMAs are done
    unsigned char* Wl = smem + REGION;
    float* red = (float*)smem;                     // used after the strip loop only (which ends with a barrier)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cc = tid % CPR, rr = tid / CPR;
    const int ft = wave & 1, st0 = 3 * (wave >> 1);
    const int64_t nstrips = (a.n_windows + CONV_WPB - 1) / CONV_WPB;
    const int64_t total_rows = a.n_windows * 12;

    // conv2 weights -> LDS (padded pitch: 16 rows x one chunk land on 16 distinct 16-byte slots)
    for (int i = tid; i < 64 * (192 / EPC); i += 256) {
        const int row = i / (192 / EPC), ch = i % (192 / EPC);
        *(uint4*)(Wl + row * WPITCH + ch * 16) = *(const uint4*)((const T*)a.wc + row * 192 + ch * EPC);
    }
    // conv1 parameters of this thread's channel chunk (image build in MODE 0, r1 recompute in MODE 1)
    float wt[EPC][3], bs[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = cc * EPC + e;
#pragma unroll
        for (int k = 0; k < 3; ++k) wt[e][k] = a.w1[c * 9 + 3 + k];
        bs[e] = a.b1[c];
        sc[e] = (MODE == 0 && a.stats1) ? a.stats1[2 * 64 + c] : 1.f;
        sh[e] = (MODE == 0 && a.stats1) ? a.stats1[3 * 64 + c] : 0.f;
        if (MODE == 0 && a.bn1.acc != nullptr) sm_bn_channel(a.bn1, c, blockIdx.x == 0 && rr == 0, sc[e], sh[e]);
    }
    float b2v[4][4];                               // conv2 bias of this lane's 16 output features
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) b2v[q][e] = (MODE == 0) ? a.bias2[ft * 32 + 8 * q + 4 * h + e] : 0.f;
    f32x2_t s1[EPC / 2], s2[EPC / 2];                // channel pairs (packed f32 adds / fmas)
#pragma unroll
    for (int k = 0; k < EPC / 2; ++k) s1[k] = s2[k] = (f32x2_t){0.f, 0.f};
    const bool want_sums = MODE != 0 || a.partials != nullptr || a.acc_out != nullptr;

    auto load_x = [&](int64_t strip) {
        const int64_t i = strip * (CONV_WPB * 12) + tid;
        return (tid < CONV_WPB * 12 && strip < nstrips && i < total_rows) ? a.x[i] : 0.f;
    };
    if (MODE == 0 && tid < CONV_WPB * 12) xs[0][tid] = load_x(blockIdx.x);
    int xb = 0;
    // MODE 1: the strip's gradient rows (7 x 16 bytes per thread) are requested one strip ahead, behind the barrier that
    // publishes the current image, and land under its MFMAs and epilogue
    constexpr int NITG = CONV_IMG_ROWS / RPP;
    uint4 pg[MODE == 1 ? NITG : 1];
    auto prefetch_g = [&](int64_t strip) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int it = 0; it < NITG; ++it) {
                const int ir = rr + it * RPP;
                const int nl = ir / 14, wp = ir % 14;
                const int64_t win = strip * CONV_WPB + nl;
                const bool ok = strip < nstrips && wp >= 1 && wp <= 12 && win < a.n_windows;
                pg[it] = *(const uint4*)((const T*)a.gin + ((ok ? win : 0) * 12 + (ok ? wp - 1 : 0)) * 64 + cc * EPC);
            }
        }
    };
    prefetch_g(blockIdx.x);
    __syncthreads();
    for (int64_t strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int64_t win0 = strip * CONV_WPB;
        const float x_next = MODE == 0 ? load_x(strip + gridDim.x) : 0.f;
        const float* xcur = xs[MODE == 0 ? xb : 0];
        // ---- stage the image: 16 windows x 14 rows x 64 channels ------------------------------
        // (fully unrolled: the CONV_IMG_ROWS / RPP independent global loads are issued together)
        {
            constexpr int NIT = CONV_IMG_ROWS / RPP;
            uint4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int ir = rr + it * RPP;
                const int nl = ir / 14, wp = ir % 14;
                const int64_t win = win0 + nl;
                const bool ok = wp >= 1 && wp <= 12 && win < a.n_windows;
                const int64_t winc = ok ? win : 0;                 // clamped: the load itself is unconditional
                const int wpos = ok ? wp - 1 : 0;
                if constexpr (MODE == 0) {
                    float t[EPC];
                    conv1_chunk_row<T>(xcur + nl * 12, wpos, wt, bs, t);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) t[e] = fmaf(t[e], sc[e], sh[e]);
                    v[it] = D::pack(t);
                } else {
                    v[it] = pg[it];
                }
                if (!ok) v[it] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) *(uint4*)(img + G::img_off(rr + it * RPP, cc)) = v[it];
        }
        __syncthreads();
        prefetch_g(strip + gridDim.x);
        // ---- implicit GEMM: out[m][f] = sum_tap sum_c image[row(m)+tap][c] * W[f][tap*64+c] ----
        f32x16 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
        int ir0[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ml = (st0 + j) * 32 + r;
            ir0[j] = (ml / 12) * 14 + (ml % 12);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int tap = ks / KS_PER_TAP, chunk = (ks % KS_PER_TAP) * 2 + h;
            const uint4 fw = *(const uint4*)(Wl + (ft * 32 + r) * WPITCH + ks * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint4 fs = *(const uint4*)(img + G::img_off(ir0[j] + tap, chunk));
                mma_chunk<T>(fw, fs, acc[j]);
            }
        }
        if (MODE == 0 && tid < CONV_WPB * 12) xs[xb ^ 1][tid] = x_next;     // read from the next strip on, two barriers away
        __syncthreads();                           // image dead: the region becomes the output tile
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int srow = (st0 + j) * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fl = ft * 32 + 8 * q + 4 * h;
                f32x2_t v0 = {acc[j][4 * q], acc[j][4 * q + 1]}, v1 = {acc[j][4 * q + 2], acc[j][4 * q + 3]};
                if constexpr (MODE == 0) {
                    v0 += (f32x2_t){b2v[q][0], b2v[q][1]};
                    v1 += (f32x2_t){b2v[q][2], b2v[q][3]};
                }
                unsigned char* dst = Cs + srow * CPITCH + fl * (int)sizeof(T);
                if constexpr (sizeof(T) == 2) {
                    // bf16: round, then ReLU on the packed pair -- the same value as round(relu(.))
                    *(uint2*)dst = make_uint2(cvt_pk_bf16<MODE == 0>(v0.x, v0.y), cvt_pk_bf16<MODE == 0>(v1.x, v1.y));
                } else {
                    if constexpr (MODE == 0) { v0 = __builtin_elementwise_max(v0, (f32x2_t){0.f, 0.f}); v1 = __builtin_elementwise_max(v1, (f32x2_t){0.f, 0.f}); }
                    *(float4*)dst = make_float4(v0.x, v0.y, v1.x, v1.y);
                }
            }
        }
        __syncthreads();
        // ---- store-out: 16-byte row segments + BatchNorm partial sums ----------------------------
#pragma unroll
        for (int p = 0; p < CONV_ROWS / RPP; ++p) {
            const int row = rr + p * RPP;
            const int64_t m = win0 * 12 + row;
            const bool ok = m < total_rows;
            const int64_t mc = ok ? m : 0;                      // loads unconditional, only the store is guarded
            const uint4 c = *(const uint4*)(Cs + row * CPITCH + cc * 16);
            float v[EPC];
            D::unpack(ok ? c : make_uint4(0, 0, 0, 0), v);      // rows past the end count as zeros
            if constexpr (MODE == 0) {
                if (want_sums) {                           // (uniform: evaluation with the running statistics has no reader for them)
#pragma unroll
                    for (int k = 0; k < EPC / 2; ++k) {
                        const f32x2_t vp = {v[2 * k], v[2 * k + 1]};
                        s1[k] += vp;
                        s2[k] = __builtin_elementwise_fma(vp, vp, s2[k]);
                    }
                }
            } else {
                float r1[EPC];
                conv1_chunk<T>(a.x, mc / 12, (int)(mc % 12), wt, bs, r1);
#pragma unroll
                for (int k = 0; k < EPC / 2; ++k) {
                    const f32x2_t vp = {v[2 * k], v[2 * k + 1]}, rp = {r1[2 * k], r1[2 * k + 1]};
                    s1[k] += vp;
                    s2[k] = __builtin_elementwise_fma(vp, rp, s2[k]);
                }
            }
            if constexpr (MODE == 0 && sizeof(T) == 2) {
                if (a.out8 != nullptr) {
                    // CP_FP8: the eight values of the chunk as e4m3 with the tensor's scale (v >= 0: clamp from above only)
                    float w8[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) { w8[e] = v[e] * out_scale; out_max = fmaxf(out_max, w8[e]); }          // (v >= 0; the conversion saturates at 448)
                    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[0], w8[1], 0, false), p1 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[4], w8[5], 0, false);
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[2], w8[3], p0, true);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[6], w8[7], p1, true);
                    if (ok) *(uint2*)(a.out8 + m * 64 + cc * EPC) = make_uint2((uint32_t)p0, (uint32_t)p1);
                    continue;
                }
            }
            if (ok) *(uint4*)((T*)a.out + m * 64 + cc * EPC) = c;
        }
        __syncthreads();                           // output tile dead before the next image is staged
        xb ^= 1;
    }
    if constexpr (MODE == 0 && sizeof(T) == 2) {
        if (a.out8 != nullptr) {
            out_max = wave_max(out_max);               // (fp8.cuh, f8_atomic_amax: the atomic only where it raises the word)
            if ((tid & 63) == 0 && out_max > 0.f && __float_as_uint(out_max) > __hip_atomic_load(a.out_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(a.out_amax, __float_as_uint(out_max));
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[(0 * RPP + rr) * 64 + cc * EPC + e] = s1[e / 2][e & 1]; red[(1 * RPP + rr) * 64 + cc * EPC + e] = s2[e / 2][e & 1]; }
    __syncthreads();
    if (tid < 128 && want_sums) {
        const int which = tid >> 6, c = tid & 63;
        float s = 0.f;
        for (int q = 0; q < RPP; ++q) s += red[(which * RPP + q) * 64 + c];
        if (MODE == 0 && a.acc_out != nullptr) sm_acc_add(a.acc_out + which * 64 + c, s, SM_ACT_SHIFT);
        else a.partials[((int64_t)blockIdx.x * 2 + which) * 64 + c] = s;
    }
}

// ------------------------------------------------------------------------------------------
// conv2 weight gradient, persistent:  dW[o][tap*64+i] = sum_m g_y2[m][o] * u1[m + tap - 1][i]
// Both operands are held as padded images (zero row, 12 positions, zero row per window) with one
// extra zero guard row at each end, so the tap is a constant row shift of the X image and the
// reduction simply runs over all image rows of the strip (8 windows).  4 waves: wave w owns
// output-channel tile (w&1), in-channel tile (w>>1) and all three taps of the 64 x 192 result, kept
// in registers across strips; one f32 slab per block at the end (reduce_slabs_kernel mode 2 sums them).
// ------------------------------------------------------------------------------------------
#define CONV_WG_WPB 8                                  // windows per strip of the weight-gradient kernel
#define CONV_WG_IMG (CONV_WG_WPB * 14)                 // 112 image rows (a multiple of the 16-row k-step)
template <typename T, bool G8 = false, bool BN2 = false>
__global__ __launch_bounds__(256, BN2 ? 1 : 2) void conv2_wgrad_kernel(ConvArgs a) {
    static_assert(!G8 || sizeof(T) == 2, "e5m2 gradients are expanded into a bf16 image");
    static_assert(!(G8 && BN2), "the small-batch form reads T gradients");
    using GV = std::conditional_t<G8, uint2, uint4>;         // one 8-channel chunk of the gradient as it sits in memory
    using D = DT<T>;
    using G = ConvGeo<T>;
    constexpr int EPC = G::EPC, CPR = G::CPR, RPP = G::RPP, KSTEP = D::KSTEP;
    constexpr int PITCH = TNPitch<T, 64>::value;              // row pitch that keeps the transposed reads conflict-free
    constexpr int ROWS = CONV_WG_IMG + 2;                     // guard row before and after
    constexpr int NIT = (CONV_WG_IMG + RPP - 1) / RPP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ROWS * PITCH];
    unsigned char* Xi = smem;                                 // g_y2 image
    unsigned char* Yi = smem + ROWS * PITCH;                  // u1 image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cc = tid % CPR, rr = tid / CPR;
    const int ot = wave & 1, it = wave >> 1;                  // this wave: out-channel tile, in-channel tile, all 3 taps
    const int64_t nstrips = (a.n_windows + CONV_WG_WPB - 1) / CONV_WG_WPB;

    float wt[EPC][3], bs[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = cc * EPC + e;
#pragma unroll
        for (int k = 0; k < 3; ++k) wt[e][k] = a.w1[c * 9 + 3 + k];
        bs[e] = a.b1[c];
        sc[e] = a.stats1 ? a.stats1[2 * 64 + c] : 1.f;         // nullptr: the image holds the raw r1 (conv2_wgrad_finish_kernel
        sh[e] = a.stats1 ? a.stats1[3 * 64 + c] : 0.f;         // applies BatchNorm1's scale and shift to the 64 x 192 product)
    }
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
    // BN2 (small batches): BatchNorm2 + ReLU backward of the staged gradient, coefficients of this thread's channels in registers
    uint4 xr2[BN2 ? NIT : 1];
    float c2a[BN2 ? EPC : 1], c2b[BN2 ? EPC : 1], c2z[BN2 ? EPC : 1], cs_all[BN2 ? EPC : 1], cs_first[BN2 ? EPC : 1], cs_last[BN2 ? EPC : 1];
    if constexpr (BN2) {
        __shared__ float ca_s[64], cb_s[64], cz_s[64];
        conv_bn_bwd_coef(a.bn2_rows, a.bn2_nr, 12, a.stats2, (double)a.n_windows * 12, ca_s, cb_s, cz_s, (double (*)[4][64])smem, blockIdx.x == 0,
                         a.dgamma2, a.dbeta2);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            c2a[e] = ca_s[cc * EPC + e]; c2b[e] = cb_s[cc * EPC + e]; c2z[e] = cz_s[cc * EPC + e];
            cs_all[e] = cs_first[e] = cs_last[e] = 0.f;
        }
        __syncthreads();                               // (the reduction scratch is the image region)
    }
    // guard rows stay zero for the whole kernel
    if (rr == 0) {
        *(uint4*)(Xi + 0 * PITCH + cc * 16) = make_uint4(0, 0, 0, 0);
        *(uint4*)(Xi + (ROWS - 1) * PITCH + cc * 16) = make_uint4(0, 0, 0, 0);
        *(uint4*)(Yi + 0 * PITCH + cc * 16) = make_uint4(0, 0, 0, 0);
        *(uint4*)(Yi + (ROWS - 1) * PITCH + cc * 16) = make_uint4(0, 0, 0, 0);
    }
    // The strip's global loads (the gradient rows and the three inputs under conv1's taps, per image row of this thread)
    // are requested one strip ahead, right behind the barrier that publishes the current images, and land under the
    // MFMA phase: 28 more live registers (two blocks per CU instead of three), no exposed load latency.
    GV xg[NIT];
    float xr[NIT][3];
    float gd = 1.f;
    if constexpr (G8) gd = f8_exp2i(-*a.gin_exp);
    auto prefetch = [&](int64_t strip) {
        const int64_t win0 = strip * CONV_WG_WPB;
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int ir = rr + q * RPP;
            const int nl = ir / 14, wp = ir % 14;
            const int64_t win = win0 + nl;
            const bool ok = strip < nstrips && ir < CONV_WG_IMG && wp >= 1 && wp <= 12 && win < a.n_windows;
            const int64_t winc = ok ? win : 0;                     // clamped: loads are unconditional
            const int wpos = ok ? wp - 1 : 0;
            xg[q] = *(const GV*)((const unsigned char*)a.gin + ((winc * 12 + wpos) * 64 + cc * EPC) * (int64_t)sizeof(GV) / EPC);
            if constexpr (BN2) xr2[q] = *(const uint4*)((const T*)a.r2 + (winc * 12 + wpos) * 64 + cc * EPC);
            const float* xw = a.x + winc * 12;
            xr[q][0] = xw[wpos > 0 ? wpos - 1 : 0];
            xr[q][1] = xw[wpos];
            xr[q][2] = xw[wpos < 11 ? wpos + 1 : 11];
        }
    };
    prefetch(blockIdx.x);
    for (int64_t strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int64_t win0 = strip * CONV_WG_WPB;
        {
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int ir = rr + q * RPP;
                const int nl = ir / 14, wp = ir % 14;
                const bool ok = ir < CONV_WG_IMG && wp >= 1 && wp <= 12 && win0 + nl < a.n_windows;
                const int wpos = ok ? wp - 1 : 0;
                float t[EPC];
                conv1_chunk_vals<T>(wpos > 0 ? xr[q][0] : 0.f, xr[q][1], wpos < 11 ? xr[q][2] : 0.f, wt, bs, t);
                if (a.stats1 != nullptr) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) t[e] = fmaf(t[e], sc[e], sh[e]);
                }
                uint4 yv = D::pack(t);
                uint4 xv;
                if constexpr (G8) xv = f8_chunk5_to_bf16(xg[q], gd);
                else xv = xg[q];
                if constexpr (BN2) {
                    float gv[EPC], rv[EPC];
                    D::unpack(xv, gv);
                    D::unpack(xr2[q], rv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) gv[e] = rv[e] > 0.f ? fmaf(c2a[e], gv[e], fmaf(c2b[e], rv[e], c2z[e])) : 0.f;
                    xv = D::pack(gv);
                    if (ok) {
                        *(uint4*)((T*)const_cast<void*>(a.gin) + ((win0 + nl) * 12 + wpos) * 64 + cc * EPC) = xv;      // (each element belongs to one thread of one block)
                        D::unpack(xv, gv);                                                            // the sums are of the values as stored
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            cs_all[e] += gv[e];
                            cs_first[e] += wpos == 0 ? gv[e] : 0.f;
                            cs_last[e] += wpos == 11 ? gv[e] : 0.f;
                        }
                    }
                }
                if (!ok) { xv = make_uint4(0, 0, 0, 0); yv = make_uint4(0, 0, 0, 0); }
                if (ir < CONV_WG_IMG) {
                    *(uint4*)(Xi + (ir + 1) * PITCH + cc * 16) = xv;
                    *(uint4*)(Yi + (ir + 1) * PITCH + cc * 16) = yv;
                }
            }
        }
        __syncthreads();
        prefetch(strip + gridDim.x);
#pragma unroll
        for (int k0 = 0; k0 < CONV_WG_IMG; k0 += KSTEP) {
            // image row j lives at LDS row j+1; the X row paired with Y image row j under tap t is image row j - t + 1
            uint4 fy;
            if constexpr (sizeof(T) == 2) fy = tn_frag_bf16<PITCH>(Yi, k0 + 1, it * 32, lane);
            else fy = tn_frag_f32<PITCH>(Yi, k0 + 1, it * 32, lane);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                uint4 fx;
                if constexpr (sizeof(T) == 2) fx = tn_frag_bf16<PITCH>(Xi, k0 + 2 - tap, ot * 32, lane);
                else fx = tn_frag_f32<PITCH>(Xi, k0 + 2 - tap, ot * 32, lane);
                mma_chunk<T>(fx, fy, acc[tap]);
            }
        }
        __syncthreads();
    }
    float* slab = a.partials + (int64_t)blockIdx.x * 64 * 192;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int q = tap * 64 + it * 32 + r;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int p = ot * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
            slab[p * 192 + q] = acc[tap][g];
        }
    }
    if constexpr (BN2) {
        // column sums of the transformed gradient (all positions / position 0 / position 11): the row slots folded in a fixed order
        float* red = (float*)smem;                         // [3][RPP][64]  (the strip loop ended with a barrier)
        static_assert(2 * ROWS * PITCH >= 3 * RPP * 64 * 4, "the column-sum reduction reuses the image region");
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[(0 * RPP + rr) * 64 + cc * EPC + e] = cs_all[e];
            red[(1 * RPP + rr) * 64 + cc * EPC + e] = cs_first[e];
            red[(2 * RPP + rr) * 64 + cc * EPC + e] = cs_last[e];
        }
        __syncthreads();
        if (tid < 192) {
            const int which = tid >> 6, c = tid & 63;
            float s = 0.f;
            for (int q = 0; q < RPP; ++q) s += red[(which * RPP + q) * 64 + c];
            a.gcols3[((int64_t)blockIdx.x * 3 + which) * 64 + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv2 data gradient + BatchNorm1 / ReLU backward + conv1's weight and bias gradient in ONE pass over g_y2 (round 4).
// dL/d(BN1 output) = g_v1 is never written: the BatchNorm1-backward coefficients exist before the launch (their two sums
// follow from conv2's weight gradient: conv2_wgrad_finish_kernel), so each 32 x 32 accumulator tile is consumed where it
// lands.  The tile is formed TRANSPOSED against conv2_strip_kernel (image rows as the MFMA's A operand, weights as B): a lane
// holds ONE conv1 channel (its three taps, bias and three coefficients: 7 registers) and 16 rows, in groups of four
// consecutive positions of one window; the six inputs under such a group's taps come from a zero-padded copy of the strip's
// x in LDS (16 floats per window: one 16-byte and one 8-byte read).  Per element:
//     r1 = round_T(relu(b + w . x))          (conv1_chunk_vals' order of operations: the value every other consumer sees)
//     gy = r1 > 0 ? ca g_v1 + cb r1 + cz : 0
//     dW1[c][tap] += gy x[pos + tap - 1],  db1[c] += gy
// partials[block][4][64]: dW tap 0..2, db (conv1_bwd_finalize_kernel).  Two barriers per strip (image published, image dead);
// no output tile in LDS.  Replaces conv2_strip_kernel<T, 1> + conv1_bwd_kernel: 258 MB less written and 266 MB less read per
// step at 167,936 windows.
// ------------------------------------------------------------------------------------------
template <typename T, bool G8 = false, bool ROWS = false>
__global__ __launch_bounds__(256, 2) void conv2_dgrad_conv1_kernel(ConvArgs a) {
    static_assert(!G8 || sizeof(T) == 2, "e5m2 gradients are expanded into a bf16 image");
    using GV = std::conditional_t<G8, uint2, uint4>;
    using D = DT<T>;
    using G = ConvGeo<T>;
    constexpr int EPC = G::EPC, CPR = G::CPR, ROWB = G::ROWB, RPP = G::RPP, WPITCH = G::WPITCH;
    constexpr int KSTEP = D::KSTEP, KS_PER_TAP = 64 / KSTEP, NKS = 3 * KS_PER_TAP;
    constexpr int IMG_BYTES = CONV_IMG_ROWS * ROWB, W_BYTES = 64 * WPITCH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[IMG_BYTES + W_BYTES];
    __shared__ __attribute__((aligned(16))) float xpad[2][CONV_WPB * 16];      // [window][0, x0..x11, 0, -, -], double-buffered
    static_assert(IMG_BYTES >= 4 * 8 * 32 * 4, "the end-of-kernel reduction reuses the image region");
    unsigned char* img = smem;
    unsigned char* Wl = smem + IMG_BYTES;
    float* red = (float*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cc = tid % CPR, rr = tid / CPR;
    const int ft = wave & 1, st0 = 3 * (wave >> 1);
    const int64_t nstrips = (a.n_windows + CONV_WPB - 1) / CONV_WPB;
    const int64_t total_rows = a.n_windows * 12;

    for (int i = tid; i < 64 * (192 / EPC); i += 256) {
        const int row = i / (192 / EPC), ch = i % (192 / EPC);
        *(uint4*)(Wl + row * WPITCH + ch * 16) = *(const uint4*)((const T*)a.wc + row * 192 + ch * EPC);
    }
    for (int i = tid; i < 2 * CONV_WPB * 16; i += 256) (&xpad[0][0])[i] = 0.f;
    // this lane's conv1 channel
    const int ch1 = ft * 32 + r;
    const float w0 = a.w1[ch1 * 9 + 3], w1 = a.w1[ch1 * 9 + 4], w2 = a.w1[ch1 * 9 + 5], b1 = a.b1[ch1];
    float ca, cb, cz;
    if constexpr (ROWS) {
        // small batches: BatchNorm1's backward coefficients from conv2_wgrad_finish_kernel's rows, here (no finalize launch in between)
        __shared__ float ca_s[64], cb_s[64], cz_s[64];
        conv_bn_bwd_coef(a.rows1, a.rows1_nr, 1, a.stats1, (double)a.n_windows * 12, ca_s, cb_s, cz_s, (double (*)[4][64])smem, blockIdx.x == 0,
                         a.dgamma1, a.dbeta1);
        ca = ca_s[ch1]; cb = cb_s[ch1]; cz = cz_s[ch1];
        __syncthreads();                               // (the reduction scratch is the image region; the weights come after it)
    } else {
        ca = a.coef[ch1]; cb = a.coef[64 + ch1]; cz = a.coef[128 + ch1];
    }
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;

    auto load_x = [&](int64_t strip) {
        const int64_t i = strip * (CONV_WPB * 12) + tid;
        return (tid < CONV_WPB * 12 && strip < nstrips && i < total_rows) ? a.x[i] : 0.f;
    };
    const int xslot = (tid / 12) * 16 + 1 + tid % 12;          // (tid < 192)
    constexpr int NITG = CONV_IMG_ROWS / RPP;
    GV pg[NITG];
    float gd = 1.f;
    if constexpr (G8) gd = f8_exp2i(-*a.gin_exp);
    auto prefetch_g = [&](int64_t strip) {
#pragma unroll
        for (int it = 0; it < NITG; ++it) {
            const int ir = rr + it * RPP;
            const int nl = ir / 14, wp = ir % 14;
            const int64_t win = strip * CONV_WPB + nl;
            const bool ok = strip < nstrips && wp >= 1 && wp <= 12 && win < a.n_windows;
            pg[it] = *(const GV*)((const unsigned char*)a.gin + (((ok ? win : 0) * 12 + (ok ? wp - 1 : 0)) * 64 + cc * EPC) * (int64_t)sizeof(GV) / EPC);
            if (!ok) pg[it] = GV{};
        }
    };
    __syncthreads();                                   // (the zeroed pads before the first x values)
    if (tid < CONV_WPB * 12) xpad[0][xslot] = load_x(blockIdx.x);
    prefetch_g(blockIdx.x);
    int xb = 0;
    int ir0[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int ml = (st0 + j) * 32 + r;
        ir0[j] = (ml / 12) * 14 + (ml % 12);
    }
    for (int64_t strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int64_t win0 = strip * CONV_WPB;
        const float x_next = load_x(strip + gridDim.x);
#pragma unroll
        for (int it = 0; it < NITG; ++it) {
            if constexpr (G8) *(uint4*)(img + G::img_off(rr + it * RPP, cc)) = f8_chunk5_to_bf16(pg[it], gd);
            else *(uint4*)(img + G::img_off(rr + it * RPP, cc)) = pg[it];
        }
        __syncthreads();
        prefetch_g(strip + gridDim.x);
        f32x16 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int tap = ks / KS_PER_TAP, chunk = (ks % KS_PER_TAP) * 2 + h;
            const uint4 fw = *(const uint4*)(Wl + (ft * 32 + r) * WPITCH + ks * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint4 fs = *(const uint4*)(img + G::img_off(ir0[j] + tap, chunk));
                mma_chunk<T>(fs, fw, acc[j]);          // D[row = sample][col = channel]
            }
        }
        if (tid < CONV_WPB * 12) xpad[xb ^ 1][xslot] = x_next;          // read from the next strip on, two barriers away
        __syncthreads();                               // image dead: the next strip may be staged
        const float* xs = xpad[xb];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mb = (st0 + j) * 32 + 8 * q + 4 * h;           // four consecutive rows of one window
                const int nl = mb / 12, p0 = mb % 12;
                const bool valid = win0 + nl < a.n_windows;
                const float4 xa = *(const float4*)(xs + nl * 16 + p0);
                const float2 xc = *(const float2*)(xs + nl * 16 + p0 + 4);
                const float X[6] = {xa.x, xa.y, xa.z, xa.w, xc.x, xc.y};
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    float y0 = fmaf(w2, X[e + 2], fmaf(w1, X[e + 1], fmaf(w0, X[e], b1)));
                    float y1 = fmaf(w2, X[e + 3], fmaf(w1, X[e + 2], fmaf(w0, X[e + 1], b1)));
                    float r0, r1;
                    if constexpr (sizeof(T) == 2) {
                        const uint32_t pk = cvt_pk_bf16<true>(y0, y1);
                        r0 = __uint_as_float(pk << 16);
                        r1 = __uint_as_float(pk & 0xffff0000u);
                    } else {
                        r0 = fmaxf(y0, 0.f);
                        r1 = fmaxf(y1, 0.f);
                    }
                    const float t0 = fmaf(ca, acc[j][4 * q + e], fmaf(cb, r0, cz));
                    const float t1 = fmaf(ca, acc[j][4 * q + e + 1], fmaf(cb, r1, cz));
                    const float g0 = (r0 > 0.f && valid) ? t0 : 0.f;
                    const float g1 = (r1 > 0.f && valid) ? t1 : 0.f;
                    d0 = fmaf(g0, X[e], d0);     d1 = fmaf(g0, X[e + 1], d1); d2 = fmaf(g0, X[e + 2], d2); d3 += g0;
                    d0 = fmaf(g1, X[e + 1], d0); d1 = fmaf(g1, X[e + 2], d1); d2 = fmaf(g1, X[e + 3], d2); d3 += g1;
                }
                __builtin_amdgcn_sched_barrier(0);     // (one group's six inputs live at a time: hoisted together they spill)
            }
        }
        xb ^= 1;
    }
    __syncthreads();
    // red[k][wave * 2 + h][r]: the four (wave pair, h) parts of each channel
    red[(0 * 8 + wave * 2 + h) * 32 + r] = d0;
    red[(1 * 8 + wave * 2 + h) * 32 + r] = d1;
    red[(2 * 8 + wave * 2 + h) * 32 + r] = d2;
    red[(3 * 8 + wave * 2 + h) * 32 + r] = d3;
    __syncthreads();
    {
        const int k = tid >> 6, c = tid & 63, f = c >> 5, cl = c & 31;
        float s = 0.f;
#pragma unroll
        for (int wq = 0; wq < 2; ++wq)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) s += red[(k * 8 + (f + 2 * wq) * 2 + hh) * 32 + cl];
        a.partials[((int64_t)blockIdx.x * 4 + k) * 64 + c] = s;
    }
}

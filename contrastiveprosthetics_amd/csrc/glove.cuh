// Glove-angle class encoder (BASELINE config 3):
//   zg = last( relu( BN( Linear(20 -> 256, no bias)(glove) ) ) ),  last = Linear(256 -> 16, no bias)
// The reference holds these layers as commented-out lines (code/models.py:386-391, 461) plus the built but
// unused `self.last` (code/models.py:425-428); this is their un-commented form.  One row per (group, class):
// R = B * 41 rows, 2.6 MFLOP per group -- small next to the sEMG encoder, so the two GEMMs reuse the generic
// 128-row-tile kernels and only the element-wise passes that are specific to the Linear -> BN -> ReLU order
// (the sEMG encoder is Linear -> ReLU -> BN) live here.
#pragma once
#include "common.cuh"

constexpr int GL_IN = 20;           // glove sensors kept by the reference (22 minus 2)
constexpr int GL_KP = 64;           // input width padded to one K step of either dtype
constexpr int GL_H = 256;           // hidden width (512 // 2)

// dst[r][c] = (r < src_rows && c < cols) ? src[r][c] : 0   (float32 -> T, `rows` rows of pitch ld)
template <typename T>
__global__ __launch_bounds__(256) void pad_cast_kernel(const float* __restrict__ src, int64_t src_rows, int cols, T* __restrict__ dst,
                                                       int64_t rows, int ld) {
    using D = DT<T>;
    const int64_t total = rows * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ld;
        const int c = (int)(i % ld);
        D::store(dst + i, (c < cols && r < src_rows) ? src[r * cols + c] : 0.f);
    }
}

// a = relu(scale * h + shift), 16-byte chunks; stats = [mean, invstd, scale, shift][C]
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ h, const float* __restrict__ stats, T* __restrict__ a,
                                                            int64_t rows, int C) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    const int cpr = C / EPC;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cc = (int)(i % cpr);
        float v[EPC];
        D::unpack(*(const uint4*)(h + i * EPC), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fmaf(stats[2 * C + cc * EPC + e], v[e], stats[3 * C + cc * EPC + e]), 0.f);
        *(uint4*)(a + i * EPC) = D::pack(v);
    }
}

// ReLU backward in place, g = (a > 0) ? g : 0, with the two BN-backward sums of the result against the BN input h:
// partials[block][2][C] = (sum g, sum g*h) in the layout bn_bwd_finalize_kernel reads.  C = 256, 16-byte chunks.
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_colsum_kernel(T* __restrict__ g, const T* __restrict__ a, const T* __restrict__ h,
                                                              float* __restrict__ partials, int64_t rows, int C) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    extern __shared__ float dyn_red[];                  // [rpp][2][C]
    const int cpr = C / EPC, rpp = 256 / cpr;
    const int tid = threadIdx.x, cc = tid % cpr, rr = tid / cpr;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    for (int64_t m = (int64_t)blockIdx.x * rpp + rr; m < rows; m += (int64_t)gridDim.x * rpp) {
        float gv[EPC], av[EPC], hv[EPC];
        D::unpack(*(const uint4*)(g + m * C + cc * EPC), gv);
        D::unpack(*(const uint4*)(a + m * C + cc * EPC), av);
        D::unpack(*(const uint4*)(h + m * C + cc * EPC), hv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float y = av[e] > 0.f ? gv[e] : 0.f;
            gv[e] = y;
            s1[e] += y;
            s2[e] = fmaf(y, hv[e], s2[e]);
        }
        *(uint4*)(g + m * C + cc * EPC) = D::pack(gv);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        dyn_red[(rr * 2 + 0) * C + cc * EPC + e] = s1[e];
        dyn_red[(rr * 2 + 1) * C + cc * EPC + e] = s2[e];
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        float s = 0.f;
        for (int q = 0; q < rpp; ++q) s += dyn_red[q * 2 * C + i];
        partials[(int64_t)blockIdx.x * 2 * C + i] = s;
    }
}

// BatchNorm backward, step 2, in place: g = ca*g + cb*h + cz  (coefficients of bn_bwd_finalize_kernel)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(T* __restrict__ g, const T* __restrict__ h, const float* __restrict__ coef,
                                                           int64_t rows, int C) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    const int cpr = C / EPC;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cpr) * EPC;
        float gv[EPC], hv[EPC];
        D::unpack(*(const uint4*)(g + i * EPC), gv);
        D::unpack(*(const uint4*)(h + i * EPC), hv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) gv[e] = fmaf(coef[c0 + e], gv[e], fmaf(coef[C + c0 + e], hv[e], coef[2 * C + c0 + e]));
        *(uint4*)(g + i * EPC) = D::pack(gv);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 4, 16-bit storage: the hidden layer is never stored.  Its input is 20 numbers per row, so
// h = x W1^T (256 values) is ONE v_mfma_f32_16x16x32_bf16 per 16 features x 16 rows (K = 20 zero-padded to 32) and is recomputed by
// every kernel that needs it -- four kernels instead of the generic K-padded GEMM launches and the element-wise passes between them:
//   glove_stats_kernel   h -> column sums (sum h, sum h^2) for BatchNorm; writes the padded bf16 copy of x the weight gradient reads
//   glove_fwd_kernel     h -> a = relu(BN(h)) -> zg = a W2^T (a stays in registers: it IS the second product's B operand)
//   glove_bwd_kernel<0>  h, da = dzg W2, g = [BN(h) > 0] da -> column sums (sum g, sum g h) and dW2 = dzg^T relu(BN(h)), on the matrix pipe
//   glove_bwd_kernel<1>  the same again with the coefficients final: dh = ca g + cb h + cz -> dW1 = dh^T x, on the matrix pipe
// Lane map of every 16 x 16 tile (gemm_ws16_kernel's): weights are the MFMA A operand, rows the B operand; a lane (q4 = lane >> 4,
// s16 = lane & 15) ends with features 16 ft + 4 q4 + e (e = 0..3) of row 16 tile + s16.  The contraction index of a product may be
// permuted as long as both operands agree, so the second product takes a lane's own 8 values of feature tiles 2 kb, 2 kb + 1 as its
// k block and W2 is loaded in that order -- no cross-lane traffic between the two products.
// stats / bwd kernels: the 4 waves of a workgroup split the FEATURES (4 tiles each) over the same rows; fwd: they split the ROWS.
// ------------------------------------------------------------------------------------------------------------------------------
struct GloveFusedArgs {
    const float* x;          // [R][20] f32
    const float* w1;         // glove_net.linear.1.weight [256][20] f32
    const float* w2;         // glove_net.last.0.weight [16][256] f32
    const float* stats;      // [4][256] mean, invstd, scale, shift of the BatchNorm (fwd, bwd)
    const float* coef;       // [3][256] BatchNorm-backward coefficients (bwd<1>)
    const bf16_t* dzg;       // [R][64] dL/dzg, 16 live columns (bwd)
    float* zg;               // [R][16] f32 (fwd)
    bf16_t* xp;              // [R][64] bf16 copy of x, zero padded (stats writes it -- nullptr: not wanted; the bwd kernels READ it: the same rounded
                             // inputs as the forward pass saw, and cp_glove_backward is not handed x again)
    float* slabs;            // (bwd) one weight-gradient slab per workgroup: <0> [16][256] (dW2), <1> [256][32] (dW1, k padded)
    float* partials;         // [gridDim.x][2][256]
    const uint4* frags;      // the weights in MFMA-fragment order, bf16, one 16-byte chunk per lane (glove_prep_kernel): [GLF_W1 + ft][64] =
                             // W1 tile ft, [GLF_W2 + kb][64] = W2 k block kb (fwd), [GLF_W2T + ft][64] = W2^T tile ft (bwd)
    int64_t R;
};
enum { GLF_W1 = 0, GLF_W2 = 16, GLF_W2T = 24, GLF_COUNT = 40 };

__device__ __forceinline__ s16x8 glove_pack8(const float (&v)[8]) {
    const uint4 c = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    return __builtin_bit_cast(s16x8, c);
}
// B operand of h's product for one 16-row tile: lane (row s16, k group q4) supplies x[row][8 q4 .. +7] (k >= 20, rows >= R: 0)
__device__ __forceinline__ s16x8 glove_x_frag(const float* __restrict__ x, int64_t row, int64_t R, int q4) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < R && q4 < 3) {
        const float4 lo = *(const float4*)(x + row * GL_IN + 8 * q4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
        if (q4 < 2) {
            const float4 hi = *(const float4*)(x + row * GL_IN + 8 * q4 + 4);
            v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
    }
    return glove_pack8(v);
}
// A operand of h's product for feature tile ft: lane (feature s16, k group q4) supplies W1[16 ft + s16][8 q4 .. +7]
__device__ __forceinline__ s16x8 glove_w1_frag(const float* __restrict__ w1, int ft, int s16, int q4) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (8 * q4 + i < GL_IN) ? w1[(ft * 16 + s16) * GL_IN + 8 * q4 + i] : 0.f;
    return glove_pack8(v);
}
typedef __attribute__((ext_vector_type(4))) float gl_f32x4;
__device__ __forceinline__ gl_f32x4 glove_mfma(const s16x8& a, const s16x8& b, const gl_f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// the three weight operands in fragment order, once per pass (the kernels' own prologues then are 4-16 16-byte loads per lane instead
// of 64-192 scalar loads + conversions in every one of thousands of waves): grid GLF_COUNT blocks of 64 threads
__global__ __launch_bounds__(64) void glove_prep_kernel(const float* __restrict__ w1, const float* __restrict__ w2, uint4* __restrict__ frags) {
    const int lane = threadIdx.x, s16 = lane & 15, q4 = lane >> 4, b = blockIdx.x;
    s16x8 f;
    if (b < GLF_W2) {
        f = glove_w1_frag(w1, b - GLF_W1, s16, q4);
    } else if (b < GLF_W2T) {           // lane (output j = s16, k group q4): W2[j][the 8 features lane group q4 holds of tiles 2 kb, 2 kb + 1]
        const int kb = b - GLF_W2;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w2[s16 * GL_H + (2 * kb + (i >> 2)) * 16 + 4 * q4 + (i & 3)];
        f = glove_pack8(v);
    } else {                            // lane (feature s16 of tile ft, k group q4): W2[j = 8 q4 .. +7][16 ft + s16], j < 16
        const int ft = b - GLF_W2T;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = q4 < 2 ? w2[(8 * q4 + i) * GL_H + ft * 16 + s16] : 0.f;
        f = glove_pack8(v);
    }
    frags[b * 64 + lane] = __builtin_bit_cast(uint4, f);
}
__device__ __forceinline__ s16x8 glove_frag(const uint4* __restrict__ frags, int idx, int lane) {
    return __builtin_bit_cast(s16x8, frags[idx * 64 + lane]);
}

__global__ __launch_bounds__(256) void glove_stats_kernel(GloveFusedArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s16 = lane & 15, q4 = lane >> 4;
    s16x8 w1f[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w1f[t] = glove_frag(a.frags, GLF_W1 + 4 * wave + t, lane);
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[t][e] = s2[t][e] = 0.f;
    const int64_t ntile = (a.R + 15) / 16;
    const gl_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    s16x8 xnext = glove_x_frag(a.x, (int64_t)blockIdx.x * 16 + s16, a.R, q4);       // (the next tile's inputs are in flight under this tile's arithmetic)
    for (int64_t rt = blockIdx.x; rt < ntile; rt += gridDim.x) {
        const int64_t row = rt * 16 + s16;
        const s16x8 xf = xnext;
        xnext = glove_x_frag(a.x, row + (int64_t)gridDim.x * 16, a.R, q4);
        if (a.xp != nullptr && row < a.R && wave < 2)           // wave 0: columns 0..31 = this fragment; wave 1: columns 32..63 = 0
            *(uint4*)(a.xp + row * GL_KP + 32 * wave + 8 * q4) = wave == 0 ? __builtin_bit_cast(uint4, xf) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const gl_f32x4 h = glove_mfma(w1f[t], xf, zero);      // rows past the end have x = 0: h = 0, nothing added
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[t][e] += h[e]; s2[t][e] = fmaf(h[e], h[e], s2[t][e]); }
        }
    }
    // totals over the 16 rows of the lane group; lane s16 < 8 of group q4 ends with value s16 = 4 (tile & 1) + e
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float v1[8], v2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { v1[i] = s1[2 * half + (i >> 2)][i & 3]; v2[i] = s2[2 * half + (i >> 2)][i & 3]; }
        const float r1 = row16_fold8(v1, lane), r2 = row16_fold8(v2, lane);
        if (s16 < 8) {
            const int f = (4 * wave + 2 * half + (s16 >> 2)) * 16 + 4 * q4 + (s16 & 3);
            a.partials[((int64_t)blockIdx.x * 2 + 0) * GL_H + f] = r1;
            a.partials[((int64_t)blockIdx.x * 2 + 1) * GL_H + f] = r2;
        }
    }
}

__global__ __launch_bounds__(256) void glove_fwd_kernel(GloveFusedArgs a) {
    __shared__ float sc_s[GL_H], sh_s[GL_H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s16 = lane & 15, q4 = lane >> 4;
    sc_s[tid] = a.stats[2 * GL_H + tid];
    sh_s[tid] = a.stats[3 * GL_H + tid];
    s16x8 w1f[16], w2f[8];
#pragma unroll
    for (int t = 0; t < 16; ++t) w1f[t] = glove_frag(a.frags, GLF_W1 + t, lane);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) w2f[kb] = glove_frag(a.frags, GLF_W2 + kb, lane);
    __syncthreads();
    const int64_t ntile = (a.R + 15) / 16;
    const gl_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    s16x8 xnext = glove_x_frag(a.x, ((int64_t)blockIdx.x * 4 + wave) * 16 + s16, a.R, q4);
    for (int64_t rt = (int64_t)blockIdx.x * 4 + wave; rt < ntile; rt += (int64_t)gridDim.x * 4) {
        const int64_t row = rt * 16 + s16;
        const s16x8 xf = xnext;
        xnext = glove_x_frag(a.x, row + (int64_t)gridDim.x * 64, a.R, q4);
        gl_f32x4 z = zero;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            float v[8];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int t = 2 * kb + o;
                const gl_f32x4 h = glove_mfma(w1f[t], xf, zero);
                const float4 sc = *(const float4*)(sc_s + t * 16 + 4 * q4), sh = *(const float4*)(sh_s + t * 16 + 4 * q4);
                v[4 * o + 0] = fmaxf(fmaf(sc.x, h[0], sh.x), 0.f);
                v[4 * o + 1] = fmaxf(fmaf(sc.y, h[1], sh.y), 0.f);
                v[4 * o + 2] = fmaxf(fmaf(sc.z, h[2], sh.z), 0.f);
                v[4 * o + 3] = fmaxf(fmaf(sc.w, h[3], sh.w), 0.f);
            }
            z = glove_mfma(w2f[kb], glove_pack8(v), z);
        }
        if (row < a.R) *(float4*)(a.zg + row * 16 + 4 * q4) = make_float4(z[0], z[1], z[2], z[3]);     // zg[row][j = 4 q4 + e]
    }
}

// Backward, both passes, in the TRANSPOSED tile orientation: rows are the MFMA's A operand and the weights its B operand, so a lane
// (q4, s16) ends with rows 4 q4 + e (e = 0..3) of a 16-row tile for ONE feature 16 ft + s16.  In that orientation a lane's own 8 values of
// two consecutive row tiles (32 rows) are exactly one A / B fragment of a product that contracts over ROWS -- the two weight gradients
//     dW2[j][f]  = sum_rows dzg[row][j] a[row][f]      (PASS 0: a = relu(BN(h)), its B operand; A = 8 scalars of column j of dzg)
//     dW1[f][k]  = sum_rows dh[row][f] x[row][k]       (PASS 1: dh, its A operand; B = 8 scalars of column k of the padded x)
// run on the matrix pipe inside the recompute kernels and NO row-sized tensor is written by the backward pass at all (the first form of
// this round wrote a and dL/dh, 86 MB each, for two generic TN launches: 41 + 30 + 2 x 31 + 2 x 11 us; this one: see DESIGN.md 7e).
// PASS 0 also reduces the two BatchNorm-backward sums (sum g, sum g h; g = [BN(h) > 0] dzg W2).  Per workgroup: the 4 waves split the
// features (4 tiles each), all walk the same row pairs; f32 slabs per workgroup, summed by glove_reduce_kernel (fixed order: reproducible).
// rows past the end: x = 0 and dzg = 0 there, so h = 0, g = 0 and both products receive zeros.
template <int PASS>
__global__ __launch_bounds__(256) void glove_bwd_kernel(GloveFusedArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s16 = lane & 15, q4 = lane >> 4;
    s16x8 w1b[4], w2b[4];
    float sc[4], sh[4], ca[4], cb[4], cz[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int f = (4 * wave + t) * 16 + s16;
        w1b[t] = glove_frag(a.frags, GLF_W1 + 4 * wave + t, lane);         // lane (feature s16, k group q4): W1[f][8 q4 .. +7]
        w2b[t] = glove_frag(a.frags, GLF_W2T + 4 * wave + t, lane);        // lane (feature s16, k group q4): W2[j = 8 q4 .. +7][f]
        sc[t] = a.stats[2 * GL_H + f];
        sh[t] = a.stats[3 * GL_H + f];
        if (PASS == 1) { ca[t] = a.coef[f]; cb[t] = a.coef[GL_H + f]; cz[t] = a.coef[2 * GL_H + f]; }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const gl_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    gl_f32x4 acc[4][PASS == 0 ? 1 : 2];        // PASS 0: dW2 tile [16 j][16 f] per feature tile; PASS 1: dW1 tiles [16 f][16 k] x 2 k tiles
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < (PASS == 0 ? 1 : 2); ++q) acc[t][q] = zero;
    const int64_t npair = (a.R + 31) / 32;
    // A operands of the two recompute products for row tile u of a pair: lane (row s16, k group q4) -> 16 bytes of its row
    auto load_x = [&](int64_t row) { return row < a.R ? *(const uint4*)(a.xp + row * GL_KP + 8 * q4) : make_uint4(0, 0, 0, 0); };
    auto load_dz = [&](int64_t row) { return (row < a.R && q4 < 2) ? *(const uint4*)(a.dzg + row * 64 + 8 * q4) : make_uint4(0, 0, 0, 0); };
    // The gathered operand of the weight-gradient product -- for the lane's column c the 8 rows {4 q4 + e, 16 + 4 q4 + e} of the pair --
    // is a transpose of what the wave already holds (its 16-byte row chunks): through a wave-private LDS tile (rows padded to 68 bytes,
    // so the four lane groups' rows fall on different banks).  (First version: 8-16 two-byte global loads per lane and pair -- 40-44 us.)
    __shared__ unsigned short tr_s[4][32][34];
    unsigned short (*tr)[34] = tr_s[wave];
    auto put_rows = [&](const uint4 (&v)[2]) {          // lane (row s16, k group q4) stores columns 8 q4 .. +7 of rows s16, 16 + s16
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t* d = (uint32_t*)&tr[16 * u + s16][8 * q4];
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
    };
    auto get_col = [&](int c) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r0 = (i >> 1) * 16 + 4 * q4 + 2 * (i & 1);
            w[i] = (uint32_t)tr[r0][c] | ((uint32_t)tr[r0 + 1][c] << 16);
        }
        return __builtin_bit_cast(s16x8, make_uint4(w[0], w[1], w[2], w[3]));
    };
    int64_t rp = blockIdx.x;
    uint4 xn[2], dn[2];
    auto fetch = [&](int64_t pair) {
        const int64_t row0 = pair * 32;
#pragma unroll
        for (int u = 0; u < 2; ++u) { xn[u] = load_x(row0 + 16 * u + s16); dn[u] = load_dz(row0 + 16 * u + s16); }
    };
    if (rp < npair) fetch(rp);
    for (; rp < npair; rp += gridDim.x) {
        s16x8 xa[2], dza[2], gop[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { xa[u] = __builtin_bit_cast(s16x8, xn[u]); dza[u] = __builtin_bit_cast(s16x8, dn[u]); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (the previous pair's column reads are done before the tile is rewritten)
        if (PASS == 0) { if (q4 < 2) put_rows(dn); }                       // dzg: 16 columns
        else put_rows(xn);                                                  // x: 32 columns
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // a wave's LDS operations are in order: its own writes have landed
        gop[0] = get_col(s16);                                              // PASS 0: column j = s16 of dzg; PASS 1: column k = s16 of x
        if (PASS == 1) gop[1] = get_col(16 + s16);
        if (rp + gridDim.x < npair) fetch(rp + gridDim.x);                 // the next pair's inputs are in flight under this pair's arithmetic
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float o[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const gl_f32x4 h = glove_mfma(xa[u], w1b[t], zero);          // rows 16 u + 4 q4 + e, feature 16 (4 wave + t) + s16
                const gl_f32x4 da = glove_mfma(dza[u], w2b[t], zero);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bn = fmaf(sc[t], h[e], sh[t]);
                    const float g = bn > 0.f ? da[e] : 0.f;
                    if (PASS == 0) {
                        s1[t] += g;
                        s2[t] = fmaf(g, h[e], s2[t]);
                        o[4 * u + e] = fmaxf(bn, 0.f);
                    } else {
                        o[4 * u + e] = fmaf(ca[t], g, fmaf(cb[t], h[e], cz[t]));
                    }
                }
            }
            const s16x8 own = glove_pack8(o);            // 8 rows of the pair for this lane's feature: a fragment of the row-contracting product
            if (PASS == 0) {
                acc[t][0] = glove_mfma(gop[0], own, acc[t][0]);             // A = dzg^T (j = s16), B = a: [16 j][16 f]
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[t][PASS == 0 ? 0 : q] = glove_mfma(own, gop[q], acc[t][PASS == 0 ? 0 : q]);      // A = dh^T (f = s16), B = x: [16 f][16 k]
            }
        }
    }
    // this workgroup's slab
    if (PASS == 0) {
        float* slab = a.slabs + (int64_t)blockIdx.x * (16 * GL_H);           // [16 j][256 f]: element (j = 4 q4 + e, f = tile * 16 + s16)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[(4 * q4 + e) * GL_H + (4 * wave + t) * 16 + s16] = acc[t][0][e];
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                    // column sums: this lane's rows, then the 4 lane groups
            float v1 = s1[t], v2 = s2[t];
            v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
            v2 += __shfl_xor(v2, 16, 64); v2 += __shfl_xor(v2, 32, 64);
            if (q4 == 0) {
                const int f = (4 * wave + t) * 16 + s16;
                a.partials[((int64_t)blockIdx.x * 2 + 0) * GL_H + f] = v1;
                a.partials[((int64_t)blockIdx.x * 2 + 1) * GL_H + f] = v2;
            }
        }
    } else {
        float* slab = a.slabs + (int64_t)blockIdx.x * (GL_H * 32);                  // [256 f][32 k]: element (f = tile * 16 + 4 q4 + e, k = 16 q + s16)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < (PASS == 0 ? 1 : 2); ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) slab[((4 * wave + t) * 16 + 4 * q4 + e) * 32 + 16 * q + s16] = acc[t][q][e];
    }
}

// out[p * q_valid + q] = sum over S slabs of slab[p * Q + q], q < q_valid, in a fixed order (reproducible).  dW2: P = 16, Q = q_valid = 256;
// dW1: P = 256, Q = 32, q_valid = 20.  A block folds 64 outputs: 4 slab lanes x 64 outputs, four loads in flight per thread, then LDS.
// grid (P * q_valid + 63) / 64 blocks of 256 threads.
__global__ __launch_bounds__(256) void glove_reduce_kernel(const float* __restrict__ slabs, int S, int P, int Q, int q_valid, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float acc = 0.f;
    if (i < P * q_valid) {
        const int p = i / q_valid, q = i % q_valid;
        const float* src = slabs + (int64_t)p * Q + q;
        const int64_t stride = (int64_t)P * Q;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = g;
        for (; k + 12 < S; k += 16) { a0 += src[(k + 0) * stride]; a1 += src[(k + 4) * stride]; a2 += src[(k + 8) * stride]; a3 += src[(k + 12) * stride]; }
        for (; k < S; k += 4) a0 += src[k * stride];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[g][o] = acc;
    __syncthreads();
    if (g == 0 && i < P * q_valid) out[i] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
}

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

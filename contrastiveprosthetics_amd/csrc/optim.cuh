// L2 regulariser (sum of Frobenius norms, NOT squared -- code/models.py:344-349, 467-472) fused
// with the two Adam optimisers of code/train.py:72-73,107-108, as three launches over one flat
// parameter buffer: per-chunk sums of squares -> per-tensor norms -> Adam with the regulariser's
// gradient reg * p / |p| added on the fly.
#pragma once
#include "common.cuh"

#define CP_MAX_TENSORS 64
#define OPT_CHUNK 2048

struct OptTensor {
    int64_t offset;   // into the flat buffers (floats)
    int64_t numel;
    int32_t chunk0;   // first chunk index of this tensor
    int32_t nchunks;
    int32_t group;    // 0 = emg_net, 1 = glove_net
    int32_t l2;       // takes part in the regulariser
};

struct OptArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* norm_partials;   // [total_chunks]
    float* norms;           // [n_tensors]
    float* l2_out;          // scalar: reg_glove * sum|q| + reg_emg * sum|p|
    int n_tensors, total_chunks;
    float lr[2], reg[2];
    float beta1, beta2, eps, bc1, bc2, grad_scale;
    const float* state;     // optional device cp_step_state: [1] bc1, [2] bc2, [3] lr_emg, [4] lr_glove override the values above
    OptTensor t[CP_MAX_TENSORS];
};

// the tensor a chunk belongs to: last i with t[i].chunk0 <= chunk (chunk0 ascends; a binary search, six probes of the
// kernel-argument table per block instead of a scan over all of it)
__device__ __forceinline__ int opt_find_tensor(const OptArgs& a, int chunk) {
    int lo = 0, hi = a.n_tensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.t[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void l2_sumsq_kernel(OptArgs a) {
    __shared__ float red[4];
    const int chunk = blockIdx.x;
    const int ti = opt_find_tensor(a, chunk);
    const OptTensor& t = a.t[ti];
    const int64_t base = (int64_t)(chunk - t.chunk0) * OPT_CHUNK;
    float s = 0.f;
    if (t.l2) {
        if (((t.offset | t.numel) & 3) == 0 && ((uintptr_t)a.p & 15) == 0) {
            // 16-byte loads where the tensor allows (every tensor of the model does): eight 4-byte loads per thread were the launch's time
            // (round 4, third part); the order of the additions per thread changes with it, the result stays run-to-run exact
            for (int i = threadIdx.x * 4; i < OPT_CHUNK; i += 1024) {
                const int64_t e = base + i;
                if (e < t.numel) {
                    const float4 x = *(const float4*)(a.p + t.offset + e);
                    s = fmaf(x.x, x.x, s); s = fmaf(x.y, x.y, s); s = fmaf(x.z, x.z, s); s = fmaf(x.w, x.w, s);
                }
            }
        } else {
            for (int i = threadIdx.x; i < OPT_CHUNK; i += 256) {
                const int64_t e = base + i;
                if (e < t.numel) { const float x = a.p[t.offset + e]; s = fmaf(x, x, s); }
            }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.norm_partials[chunk] = red[0] + red[1] + red[2] + red[3];
}

// one block of 16 * CP_MAX_TENSORS threads: 16 lanes per tensor walk its chunk sums (a weight matrix has 128-192 of
// them; one thread per tensor took 18-20 us), folded in a fixed order
#define L2_FIN_LANES 16
__global__ __launch_bounds__(L2_FIN_LANES * CP_MAX_TENSORS) void l2_finalize_kernel(OptArgs a) {
    __shared__ double contrib[CP_MAX_TENSORS];
    const int i = threadIdx.x / L2_FIN_LANES, l = threadIdx.x % L2_FIN_LANES;
    double s = 0;
    if (i < a.n_tensors)
        for (int k = l; k < a.t[i].nchunks; k += L2_FIN_LANES) s += (double)a.norm_partials[a.t[i].chunk0 + k];
#pragma unroll
    for (int m = 1; m < L2_FIN_LANES; m <<= 1) s += __shfl_xor(s, m, 64);
    if (l == 0) {
        double c = 0;
        if (i < a.n_tensors) {
            const float n = (float)sqrt(s);
            a.norms[i] = n;
            if (a.t[i].l2) c = (double)a.reg[a.t[i].group] * (double)n;
        }
        contrib[i] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int k = 0; k < a.n_tensors; ++k) s += contrib[k];
        *a.l2_out = (float)s;
    }
}

// FUSED: the per-tensor norms are folded from l2_sumsq_kernel's chunk sums HERE, by 16 lanes in l2_finalize_kernel's order (the same
// numbers), so the step needs no l2_finalize launch (5 us of timeline per step on every path; it is 1.5 % of a step at the reference's
// batch sizes); block 0 also writes norms[] and the regulariser value.  cp_l2_norms (no update) keeps the separate kernel.
template <bool FUSED>
__global__ __launch_bounds__(256) void adam_kernel(OptArgs a) {
    const int chunk = blockIdx.x;
    const int ti = opt_find_tensor(a, chunk);
    const OptTensor& t = a.t[ti];
    const int64_t base = (int64_t)(chunk - t.chunk0) * OPT_CHUNK;
    float norm;
    if constexpr (FUSED) {
        __shared__ float norm_s;
        __shared__ double contrib[CP_MAX_TENSORS];
        auto tensor_norm = [&](int i, int l) -> double {          // lanes l = 0..15 of one 16-lane group: sum of tensor i's chunk sums
            double s = 0;
            if (i < a.n_tensors)
                for (int k = l; k < a.t[i].nchunks; k += L2_FIN_LANES) s += (double)a.norm_partials[a.t[i].chunk0 + k];
#pragma unroll
            for (int m = 1; m < L2_FIN_LANES; m <<= 1) s += __shfl_xor(s, m, 64);
            return s;
        };
        const int grp = threadIdx.x / L2_FIN_LANES, l = threadIdx.x % L2_FIN_LANES;
        if (grp == 0) {
            const double s = tensor_norm(ti, l);
            if (l == 0) norm_s = (float)sqrt(s);
        }
        if (blockIdx.x == 0) {
            for (int i0 = 0; i0 < a.n_tensors; i0 += 256 / L2_FIN_LANES) {
                const int i = i0 + grp;
                const double s = tensor_norm(i, l);
                if (l == 0 && i < a.n_tensors) {
                    const float n = (float)sqrt(s);
                    a.norms[i] = n;
                    contrib[i] = a.t[i].l2 ? (double)a.reg[a.t[i].group] * (double)n : 0.0;
                }
            }
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            double s = 0;
            for (int k = 0; k < a.n_tensors; ++k) s += contrib[k];
            *a.l2_out = (float)s;
        }
        norm = norm_s;
    } else {
        norm = a.norms[ti];
    }
    const float lr = a.state ? a.state[3 + t.group] : a.lr[t.group];
    const float l2c = t.l2 ? a.reg[t.group] / norm : 0.f;
    const float step = lr / (a.state ? a.state[1] : a.bc1);
    const float rs2 = 1.0f / sqrtf(a.state ? a.state[2] : a.bc2);
    auto upd = [&](float p, float gr, float& m, float& v) -> float {
        float g = gr * a.grad_scale;
        if (t.l2) g = fmaf(l2c, p, g);
        m = a.beta1 * m + (1.f - a.beta1) * g;
        v = a.beta2 * v + (1.f - a.beta2) * g * g;
        return p - step * (m / (sqrtf(v) * rs2 + a.eps));
    };
    if (((t.offset | t.numel) & 3) == 0 && ((((uintptr_t)a.p | (uintptr_t)a.g | (uintptr_t)a.m | (uintptr_t)a.v)) & 15) == 0) {
        // four elements per thread and load (element-wise: the same numbers as the scalar form)
        for (int i = threadIdx.x * 4; i < OPT_CHUNK; i += 1024) {
            const int64_t e = base + i;
            if (e >= t.numel) break;
            const int64_t o = t.offset + e;
            float4 p4 = *(const float4*)(a.p + o), m4 = *(const float4*)(a.m + o), v4 = *(const float4*)(a.v + o);
            const float4 g4 = *(const float4*)(a.g + o);
            p4.x = upd(p4.x, g4.x, m4.x, v4.x); p4.y = upd(p4.y, g4.y, m4.y, v4.y);
            p4.z = upd(p4.z, g4.z, m4.z, v4.z); p4.w = upd(p4.w, g4.w, m4.w, v4.w);
            *(float4*)(a.m + o) = m4;
            *(float4*)(a.v + o) = v4;
            *(float4*)(a.p + o) = p4;
        }
        return;
    }
    for (int i = threadIdx.x; i < OPT_CHUNK; i += 256) {
        const int64_t e = base + i;
        if (e >= t.numel) break;
        const int64_t o = t.offset + e;
        float m = a.m[o], v = a.v[o];
        const float pn = upd(a.p[o], a.g[o], m, v);
        a.m[o] = m;
        a.v[o] = v;
        a.p[o] = pn;
    }
}

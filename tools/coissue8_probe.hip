// Do vector instructions hide in the gaps of the block-scaled 8-bit MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, the instruction of
// gemm_ws8 / gemm_wsd8 / gemm_tn8)?  One workgroup per CU, every CU busy, operands from a hash (not constants: the clock the
// chip holds depends on the data), timed with hipEvents and stamped with s_memtime / s_memrealtime for the in-kernel clock.
//   F = 0..8 independent v_fma_f32 placed behind EACH MFMA of a chain-free loop of 16 MFMAs (one wave per SIMD);
//   split: 8 waves (two per SIMD): waves 0-3 the bare MFMA loop, waves 4-7 the same number of v_fma_f32 as F gives.
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue8_probe.hip -o build/coissue8_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
#define ITER 2048
#define NM 16

__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// ROLE 0: MFMA + F fillers per MFMA in one wave; ROLE 1: split (waves 0-3 MFMA, waves 4-7 F * NM fillers per iteration)
template <int F, int ROLE>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* stamps) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = ROLE != 1 || wave < 4;
    const bool do_v = ROLE != 1 || wave >= 4;
    i32x8_t a, b;
    for (int i = 0; i < 8; ++i) {
        // e4m3 bytes with exponents in the middle of the range (no NaN: 0x7f / 0xff never appear with the top exponent bit pair cleared)
        a[i] = (int)(h32(threadIdx.x * 8 + i + blockIdx.x * 4096) & 0xB7B7B7B7u);
        b[i] = (int)(h32(threadIdx.x * 8 + i + 77777 + blockIdx.x * 4096) & 0xB7B7B7B7u);
    }
    f32x4_t acc[NM];
    for (int i = 0; i < NM; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(h32(threadIdx.x + i) & 0xFFFF) * 1e-4f;
    const float m = 0.99991f, c = 0.5f;
    const int sc = 127;
    float w[8], o[8], t[2] = {0.f, 0.f}, mx = 0.f, lim = 448.f;
    int pk = 0;
    for (int i = 0; i < 8; ++i) { w[i] = 0.f; o[i] = (float)(h32(threadIdx.x * 9 + i) & 0xFFFF) * 1e-2f - 100.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
        if (ROLE == 0) {
            // (placement pinned with inline assembly: sched_group_barrier left clusters of 12 bare MFMAs between filler groups)
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[i]) : "v"(a), "v"(b), "v"(sc));
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * F + f) & 7]) : "s"(m), "v"(c));
            }
        } else if (ROLE == 2) {
            // the paced epilogue's instruction mix behind each MFMA: clamp -> sum, sum of squares (a dependent chain of two), every other
            // MFMA also the maximum and the conversion: F = 4 average
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[i]) : "v"(a), "v"(b), "v"(sc));
                asm volatile("v_med3_f32 %0, %3, 0, %4\n\tv_add_f32 %1, %1, %0\n\tv_fmac_f32 %2, %0, %0" : "=&v"(t[i & 1]), "+v"(v[i & 7]), "+v"(w[i & 7]) : "v"(o[i & 7]), "v"(lim));
                if (i & 1) asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_cvt_pk_fp8_f32 %1, %4, %5" : "+v"(mx), "+v"(pk) : "v"(o[i & 7]), "v"(o[(i - 1) & 7]), "v"(t[0]), "v"(t[1]));
            }
        } else if (ROLE == 3) {
            // the same instructions, two outputs' chains interleaved behind every other MFMA (no instruction waits for its neighbour)
#pragma unroll
            for (int i = 0; i < NM; i += 2) {
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[i]) : "v"(a), "v"(b), "v"(sc));
                asm volatile("v_med3_f32 %0, %2, 0, %4\n\tv_med3_f32 %1, %3, 0, %4\n\tv_max3_f32 %5, %5, %2, %3" : "=&v"(t[0]), "=&v"(t[1]), "+v"(o[i & 7]), "+v"(o[(i + 1) & 7]), "+v"(lim), "+v"(mx));
                asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(v[i & 7]), "+v"(v[(i + 1) & 7]) : "v"(t[0]), "v"(t[1]));
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[i + 1]) : "v"(a), "v"(b), "v"(sc));
                asm volatile("v_fmac_f32 %0, %2, %2\n\tv_fmac_f32 %1, %3, %3\n\tv_cvt_pk_fp8_f32 %4, %2, %3" : "+v"(w[i & 7]), "+v"(w[(i + 1) & 7]) : "v"(t[0]), "v"(t[1]), "v"(pk));
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < NM; ++i) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[i]) : "v"(a), "v"(b), "v"(sc));
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < NM * F; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "s"(m), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + o[i];
    s += mx + (float)pk + t[0] + t[1] + lim;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int F, int ROLE>
static void run(float* out, unsigned long long* stamps, const char* name) {
    const int threads = ROLE == 1 ? 512 : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<F, ROLE>), dim3(256), dim3(threads), 0, 0, out, stamps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<F, ROLE>), dim3(256), dim3(threads), 0, 0, out, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), stamps, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int b = 0; b < 256; ++b) { cyc.push_back((double)h[2 * b]); clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double us = ms / 10 * 1e3;
    printf("%-28s F=%d  %8.1f us   %6.1f cycles per MFMA (stamped, median)   clock %5.0f MHz   %5.0f TF/s\n", name, F, us,
           cyc[128] / ((double)ITER * NM), clk[128], 256.0 * 4 * ITER * NM * 2.0 * 16 * 16 * 128 / (us * 1e-6) * 1e-12);
}

int main() {
    float* out; unsigned long long* stamps;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&stamps, 512 * sizeof(unsigned long long));
    run<0, 0>(out, stamps, "one wave, MFMA + F fma");
    run<1, 0>(out, stamps, "one wave, MFMA + F fma");
    run<2, 0>(out, stamps, "one wave, MFMA + F fma");
    run<3, 0>(out, stamps, "one wave, MFMA + F fma");
    run<4, 0>(out, stamps, "one wave, MFMA + F fma");
    run<5, 0>(out, stamps, "one wave, MFMA + F fma");
    run<6, 0>(out, stamps, "one wave, MFMA + F fma");
    run<8, 0>(out, stamps, "one wave, MFMA + F fma");
    run<4, 2>(out, stamps, "epilogue mix, chains");
    run<4, 3>(out, stamps, "epilogue mix, interleaved");
    run<2, 1>(out, stamps, "two waves, MFMA | F fma");
    run<4, 1>(out, stamps, "two waves, MFMA | F fma");
    run<6, 1>(out, stamps, "two waves, MFMA | F fma");
    run<8, 1>(out, stamps, "two waves, MFMA | F fma");
    return 0;
}

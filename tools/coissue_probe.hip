// Do vector and matrix instructions overlap on one SIMD?  Four kernels, one workgroup per CU, timed with hipEvents:
//   mfma:  4 waves (one per SIMD), each a chain-free loop of v_mfma_f32_16x16x32_bf16
//   valu:  4 waves, each a loop of independent v_fma_f32 (or v_pk_fma_f32 with -DPK)
//   same:  4 waves, each wave issues both streams interleaved
//   split: 8 waves (two per SIMD): waves 0-3 run the mfma loop, waves 4-7 the valu loop
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o build/coissue_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#define ITER 4096
template <int MODE>   // 0 mfma, 1 valu, 2 same wave, 3 split waves
__global__ __launch_bounds__(512) void probe(float* out, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#ifdef PK
    f32x2_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = (f32x2_t){(float)threadIdx.x, 1.f};
    const f32x2_t m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
#else
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x + i;
    const float m = 1.0001f, c = 0.5f;
#endif
    for (int it = 0; it < ITER; ++it) {
        if (do_m) {
#pragma unroll
            for (int r = 0; r < 1; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#ifdef PK
                    v[i] = __builtin_elementwise_fma(v[i], m, c);
#else
                    v[i] = fmaf(v[i], m, c);
#endif
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#ifdef PK
        s += v[i].x + v[i].y;
#else
        s += v[i];
#endif
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static float run(float* out, int threads) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, 0, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, 0, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const float tm = run<0>(out, 256), tv = run<1>(out, 256), ts = run<2>(out, 256), tp = run<3>(out, 512);
    // per loop iteration and wave: 8 MFMAs (16x16x32 bf16) and 32 vector FMAs
    printf("mfma only  %8.1f us  (%.1f cycles per MFMA at 2.4 GHz)\n", tm, tm * 2400.0 / (ITER * 8.0));
    printf("valu only  %8.1f us  (%.1f cycles per FMA instruction at 2.4 GHz)\n", tv, tv * 2400.0 / (ITER * 32.0));
    printf("same wave  %8.1f us  (sum %.1f, max %.1f)\n", ts, tm + tv, tm > tv ? tm : tv);
    printf("two waves  %8.1f us  (sum %.1f, max %.1f)\n", tp, tm + tv, tm > tv ? tm : tv);
    return 0;
}

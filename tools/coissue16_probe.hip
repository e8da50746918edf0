// The bf16 question behind gemm_ws16 / gemm_wsd16's "epilogue BEHIND the k loop": do F vector instructions pinned behind EACH
// v_mfma_f32_16x16x32_bf16 (16 cycles, 8 of them issue) cost wall time when every CU runs the loop on hashed data -- i.e. at the
// clock the chip then holds?  round 2's probe (coissue_probe.hip) interleaved 4 per MFMA and left the placement to the compiler.
//   F = 0, 1, 2, 3, 4 v_fma_f32 behind each MFMA of a chain-free loop of 16 (one wave per SIMD), fragments re-read from LDS
//   (one ds_read_b128 per 4 MFMAs, as the kernels do) with -DLDS.
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue16_probe.hip -o build/coissue16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8;
#define ITER 4096
#define NM 16
__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int F, bool LDS>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[64 * 1024];
    s16x8 a, b[4];
    for (int i = 0; i < 8; ++i) {
        a[i] = (short)((h32(threadIdx.x * 8 + i + blockIdx.x * 4096) & 0x3FFF) | 0x3000);        // bf16 in [0.5, 2) or so, hashed mantissas
        for (int q = 0; q < 4; ++q) b[q][i] = (short)((h32(threadIdx.x * 8 + i + 77777 * (q + 1) + blockIdx.x * 4096) & 0xBFFF) | 0x3000);
    }
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) ((uint32_t*)sm)[i] = (h32(i + blockIdx.x) & 0xBFFFBFFFu) | 0x30003000u;
    __syncthreads();
    const unsigned char* base = sm + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 16384;
    f32x4_t acc[NM];
    for (int i = 0; i < NM; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(h32(threadIdx.x + i) & 0xFFFF) * 1e-4f;
    const float m = 0.99991f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b[i >> 2]));
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * F + f) & 7]) : "s"(m), "v"(c));
            if (LDS && (i & 3) == 3) b[i >> 2] = *(const s16x8*)(base + (((it * 4 + (i >> 2)) & 15) << 10));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int F, bool LDS>
static void run(float* out, unsigned long long* stamps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((probe<F, LDS>), dim3(256), dim3(256), 0, 0, out, stamps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<F, LDS>), dim3(256), dim3(256), 0, 0, out, stamps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    (void)hipMemcpy(h.data(), stamps, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int b = 0; b < 256; ++b) { cyc.push_back((double)h[2 * b]); clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double us = ms / 10 * 1e3;
    printf("bf16 16x16x32 %s F=%d  %8.1f us   %6.1f cycles per MFMA   clock %5.0f MHz   %5.0f TF/s\n", LDS ? "+ LDS fragment reads" : "registers only      ", F, us,
           cyc[128] / ((double)ITER * NM), clk[128], 256.0 * 4 * ITER * NM * 2.0 * 16 * 16 * 32 / (us * 1e-6) * 1e-12);
}
int main() {
    float* out; unsigned long long* stamps;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&stamps, 512 * sizeof(unsigned long long));
    run<0, false>(out, stamps); run<1, false>(out, stamps); run<2, false>(out, stamps); run<3, false>(out, stamps); run<4, false>(out, stamps);
    run<0, true>(out, stamps); run<1, true>(out, stamps); run<2, true>(out, stamps); run<3, true>(out, stamps);
    return 0;
}

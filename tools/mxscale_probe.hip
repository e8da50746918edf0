// What does the SCALE operand of v_mfma_scale_f32_16x16x128_f8f6f4 cost?  gemm_ws8's MFMAs + fragment reads alone run at ~53 cycles per
// MFMA (DESIGN 7f ablations) where a bare loop of the instruction runs at 32.8 (tools/coissue8_probe.hip).  Variants, all chain-free
// loops of 16 MFMAs on hashed operands, one wave per SIMD, every CU busy:
//   same     one scale register, op_sel 0 throughout
//   cycle    the scale byte selected by op_sel cycles 0,1,2,3 from one MFMA to the next (what a loop "for st: for ft:" issues)
//   by4      the scale byte changes every 4 MFMAs (loop "for ft: for st:")
//   noscale  v_mfma_f32_16x16x128_f8f6f4 (no scale operands)
//   agpr     srcA in AGPRs, op_sel cycling (the kernel's form)
//   2regs    two scale registers alternating, op_sel 0
// build: hipcc --offload-arch=gfx950 -O3 tools/mxscale_probe.hip -o build/mxscale_probe
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
#define T "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 "
template <int SEL>
__device__ __forceinline__ void mf(f32x4_t& c, const i32x8_t& a, const i32x8_t& b, int sa, int sb) {
    if constexpr (SEL == 0) asm volatile(T "op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    if constexpr (SEL == 1) asm volatile(T "op_sel:[1,0,0] op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    if constexpr (SEL == 2) asm volatile(T "op_sel_hi:[1,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    if constexpr (SEL == 3) asm volatile(T "op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
template <int SEL>
__device__ __forceinline__ void mfa(f32x4_t& c, const i32x8_t& a, const i32x8_t& b, int sa, int sb) {
    if constexpr (SEL == 0) asm volatile(T "op_sel_hi:[0,0,0]" : "+v"(c) : "a"(a), "v"(b), "v"(sa), "v"(sb));
    if constexpr (SEL == 1) asm volatile(T "op_sel:[1,0,0] op_sel_hi:[0,0,0]" : "+v"(c) : "a"(a), "v"(b), "v"(sa), "v"(sb));
    if constexpr (SEL == 2) asm volatile(T "op_sel_hi:[1,0,0]" : "+v"(c) : "a"(a), "v"(b), "v"(sa), "v"(sb));
    if constexpr (SEL == 3) asm volatile(T "op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(c) : "a"(a), "v"(b), "v"(sa), "v"(sb));
}
template <int V>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* stamps) {
    i32x8_t a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (int)(h32(threadIdx.x * 8 + i + blockIdx.x * 4096) & 0xB7B7B7B7u);
        b[i] = (int)(h32(threadIdx.x * 8 + i + 77777 + blockIdx.x * 4096) & 0xB7B7B7B7u);
    }
    i32x8_t aa = a;
    asm volatile("" : "+a"(aa));
    f32x4_t acc[NM];
    for (int i = 0; i < NM; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int s0 = 0x7f7f7f7f, s1 = 0x7f7f7f7f, sb = 127;
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(sb));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#define G4(i0, A, B, C, D, MF, AR, S0, S1, S2, S3) MF<A>(acc[i0], AR, b, S0, sb); MF<B>(acc[i0 + 1], AR, b, S1, sb); MF<C>(acc[i0 + 2], AR, b, S2, sb); MF<D>(acc[i0 + 3], AR, b, S3, sb);
        if constexpr (V == 0) { G4(0, 0, 0, 0, 0, mf, a, s0, s0, s0, s0) G4(4, 0, 0, 0, 0, mf, a, s0, s0, s0, s0) G4(8, 0, 0, 0, 0, mf, a, s0, s0, s0, s0) G4(12, 0, 0, 0, 0, mf, a, s0, s0, s0, s0) }
        if constexpr (V == 1) { G4(0, 0, 1, 2, 3, mf, a, s0, s0, s0, s0) G4(4, 0, 1, 2, 3, mf, a, s0, s0, s0, s0) G4(8, 0, 1, 2, 3, mf, a, s0, s0, s0, s0) G4(12, 0, 1, 2, 3, mf, a, s0, s0, s0, s0) }
        if constexpr (V == 2) { G4(0, 0, 0, 0, 0, mf, a, s0, s0, s0, s0) G4(4, 1, 1, 1, 1, mf, a, s0, s0, s0, s0) G4(8, 2, 2, 2, 2, mf, a, s0, s0, s0, s0) G4(12, 3, 3, 3, 3, mf, a, s0, s0, s0, s0) }
        if constexpr (V == 3) {
#pragma unroll
            for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        if constexpr (V == 4) { G4(0, 0, 1, 2, 3, mfa, aa, s0, s0, s0, s0) G4(4, 0, 1, 2, 3, mfa, aa, s0, s0, s0, s0) G4(8, 0, 1, 2, 3, mfa, aa, s0, s0, s0, s0) G4(12, 0, 1, 2, 3, mfa, aa, s0, s0, s0, s0) }
        if constexpr (V == 5) { G4(0, 0, 0, 0, 0, mf, a, s0, s1, s0, s1) G4(4, 0, 0, 0, 0, mf, a, s0, s1, s0, s1) G4(8, 0, 0, 0, 0, mf, a, s0, s1, s0, s1) G4(12, 0, 0, 0, 0, mf, a, s0, s1, s0, s1) }
        if constexpr (V == 6) { G4(0, 0, 0, 0, 0, mfa, aa, s0, s0, s0, s0) G4(4, 0, 0, 0, 0, mfa, aa, s0, s0, s0, s0) G4(8, 0, 0, 0, 0, mfa, aa, s0, s0, s0, s0) G4(12, 0, 0, 0, 0, mfa, aa, s0, s0, s0, s0) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int V>
static void run(float* out, unsigned long long* stamps, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<V>), dim3(256), dim3(256), 0, 0, out, stamps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<V>), dim3(256), dim3(256), 0, 0, out, stamps);
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
    printf("%-10s %8.1f us   %6.1f cycles per MFMA (stamped, median)   clock %5.0f MHz   %5.0f TF/s\n", name, us, cyc[128] / ((double)ITER * NM), clk[128],
           256.0 * 4 * ITER * NM * 2.0 * 16 * 16 * 128 / (us * 1e-6) * 1e-12);
}
int main() {
    float* out; unsigned long long* stamps;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&stamps, 512 * sizeof(unsigned long long));
    run<0>(out, stamps, "same");
    run<1>(out, stamps, "cycle");
    run<2>(out, stamps, "by4");
    run<3>(out, stamps, "noscale");
    run<4>(out, stamps, "agpr+cycle");
    run<6>(out, stamps, "agpr+same");
    run<5>(out, stamps, "2regs");
    return 0;
}

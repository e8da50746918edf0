// Standalone probe: verifies on a real MI355X the lane maps the GEMM kernels rely on.
//   1. v_mfma_f32_32x32x16_bf16 A/B/C maps
//   2. v_mfma_f32_32x32x2_f32  A/B/C maps
//   3. ds_read_b64_tr_b16 (transposed LDS read) as a [k][col] -> fragment loader
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// C[32][32] = A[32][16] * B[16][32], A row-major [i][k], Bt row-major [j][k]
__global__ void k_bf16(const uint16_t* A, const uint16_t* Bt, float* C) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[r * 16 + 8 * h + j]; b[j] = (short)Bt[r * 16 + 8 * h + j]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int g = 0; g < 16; ++g) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; C[row * 32 + r] = acc[g]; }
}

// f32: C[32][32] = A[32][8] * B[8][32] using 4 MFMAs of K=2; lane (r,h) holds k = 4h..4h+3
__global__ void k_f32(const float* A, const float* Bt, float* C) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {0};
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 8 + 4 * h + e], Bt[r * 8 + 4 * h + e], acc, 0, 0, 0);
    for (int g = 0; g < 16; ++g) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; C[row * 32 + r] = acc[g]; }
}

// transposed read: X stored [k=16][col=32] bf16 row-major in LDS (64-B rows).
// Want fragment: lane (r,h): elements j=0..7 = X[8h + j][r].  C = X^T(32x16) * Y(16x32)
__global__ void k_tr(const uint16_t* X, const uint16_t* Y, float* C) {
    __shared__ __attribute__((aligned(16))) uint16_t sx[16 * 32], sy[16 * 32];
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    for (int i = l; i < 512; i += 64) { sx[i] = X[i]; sy[i] = Y[i]; }
    __syncthreads();
    // 16-lane group g = l>>4 : columns 16*(g&1).., rows 8*(g>>1)..; lane 4q+p in group: row q, cols 4p..4p+3
    int g = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
    int row0 = 8 * (g >> 1), col0 = 16 * (g & 1);
    uint32_t ax = (uint32_t)(uintptr_t)(&sx[(row0 + q) * 32 + col0 + 4 * p]);
    uint32_t ay = (uint32_t)(uintptr_t)(&sy[(row0 + q) * 32 + col0 + 4 * p]);
    bf16x4 x0, x1, y0, y1;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n" : "=v"(x0) : "v"(ax));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256\n" : "=v"(x1) : "v"(ax));   // rows +4 (4*64 B)
    asm volatile("ds_read_b64_tr_b16 %0, %1\n" : "=v"(y0) : "v"(ay));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256\n" : "=v"(y1) : "v"(ay));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = x0[j]; a[j + 4] = x1[j]; b[j] = y0[j]; b[j + 4] = y1[j]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int gg = 0; gg < 16; ++gg) { int row = (gg & 3) + 8 * (gg >> 2) + 4 * h; C[row * 32 + r] = acc[gg]; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

int main() {
    int bad = 0;
    srand(1);
    {   // bf16
        std::vector<uint16_t> A(32 * 16), Bt(32 * 16); std::vector<float> C(1024), R(1024, 0.f);
        for (auto& v : A) v = f2bf((float)(rand() % 17 - 8));
        for (auto& v : Bt) v = f2bf((float)(rand() % 13 - 6));
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += bf2f(A[i * 16 + k]) * bf2f(Bt[j * 16 + k]);
        uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
        k_bf16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int e = 0; for (int i = 0; i < 1024; ++i) e += (C[i] != R[i]);
        printf("mfma_f32_32x32x16_bf16 map: %s (%d mismatches)\n", e ? "FAIL" : "ok", e); bad += e;
    }
    {   // f32
        std::vector<float> A(32 * 8), Bt(32 * 8), C(1024), R(1024, 0.f);
        for (auto& v : A) v = (float)(rand() % 17 - 8);
        for (auto& v : Bt) v = (float)(rand() % 13 - 6);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 8; ++k) R[i * 32 + j] += A[i * 8 + k] * Bt[j * 8 + k];
        float *dA, *dB, *dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
        k_f32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int e = 0; for (int i = 0; i < 1024; ++i) e += (C[i] != R[i]);
        printf("mfma_f32_32x32x2f32 map:    %s (%d mismatches)\n", e ? "FAIL" : "ok", e); bad += e;
    }
    {   // transposed read
        std::vector<uint16_t> X(512), Y(512); std::vector<float> C(1024), R(1024, 0.f);
        for (auto& v : X) v = f2bf((float)(rand() % 17 - 8));
        for (auto& v : Y) v = f2bf((float)(rand() % 13 - 6));
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += bf2f(X[k * 32 + i]) * bf2f(Y[k * 32 + j]);
        uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, X.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Y.data(), 1024, hipMemcpyHostToDevice));
        k_tr<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int e = 0; for (int i = 0; i < 1024; ++i) e += (C[i] != R[i]);
        printf("ds_read_b64_tr_b16 loader:  %s (%d mismatches)\n", e ? "FAIL" : "ok", e); bad += e;
    }
    return bad ? 1 : 0;
}

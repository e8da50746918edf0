// Does MODE.FP16_OVFL (bit 23 of the MODE register) make v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 saturate instead of returning NaN / inf?
// hipcc --offload-arch=gfx950 -O2 tools/fp8_sat_probe.hip -o /tmp/fp8_sat_probe && /tmp/fp8_sat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(const float* in, unsigned* out, int n, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = 1
    const int i = threadIdx.x;
    if (i < n) {
        const float v = in[i];
        const int a = __builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
        const int b = __builtin_amdgcn_cvt_pk_bf8_f32(v, v, 0, false);
        out[2 * i] = (unsigned)a & 0xFF;
        out[2 * i + 1] = (unsigned)b & 0xFF;
    }
    if (ovfl) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 0);
}
int main() {
    const float h[] = {1.f, 447.f, 448.f, 464.f, 465.f, 480.f, 1000.f, 1e9f, -1000.f, 57344.f, 57345.f, 61440.f, 65536.f, 1e6f, -1e6f, INFINITY, -INFINITY, NAN, 1e-10f, -0.f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned ho[2 * 32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, n, ovfl);
        hipMemcpy(ho, o, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
        printf("FP16_OVFL = %d\n", ovfl);
        for (int i = 0; i < n; ++i) printf("  cvt %12g -> fp8 0x%02x  bf8 0x%02x\n", h[i], ho[2 * i], ho[2 * i + 1]);
    }
    return 0;
}

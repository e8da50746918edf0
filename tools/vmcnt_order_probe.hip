// Does s_waitcnt vmcnt(N) order an LDS-DMA load before a YOUNGER store on gfx950?  Each wave: poison its LDS slot, issue one
// buffer_load_dwordx4 ... lds from a far-apart (HBM-miss) address, then `nstores` stores to a small hot buffer, s_waitcnt vmcnt(nstores),
// read the LDS slot back and compare with the source.  A mismatch = the wait was satisfied while the load was still in flight.
// build: hipcc --offload-arch=gfx950 -O3 tools/vmcnt_order_probe.hip -o build/vmcnt_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void bufl16_lds(const u32x4_t& rsrc, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int NST>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ src, size_t n16, uint4* __restrict__ hot, unsigned long long* bad, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 slot[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t base = (uint64_t)(uintptr_t)src;
    const u32x4_t rsrc = {(uint32_t)base, (uint32_t)(base >> 32) & 0xFFFFu, 0xFFFFFFF0u, 0x00020000u};
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)slot) + wave * 1024;
    const auto h_rsrc = __builtin_amdgcn_make_buffer_rsrc(hot, 0, 1 << 20, 0x00020000);
    unsigned long long nbad = 0;
    uint32_t x = blockIdx.x * 2654435761u + wave * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        // a 1 KiB row somewhere in the (large) source: every wave and iteration another one -> cache misses
        const size_t row = ((size_t)x * 977u + (size_t)blockIdx.x * 131071u + it * 7919u) % (n16 / 64);
        slot[tid] = make_uint4(0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bufl16_lds(rsrc, (uint32_t)(lane * 16), (uint32_t)0, lds0 - 0 + 0);   // placeholder: replaced below
        (void)row;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // real sequence
        slot[tid] = make_uint4(0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t off = (uint64_t)row * 1024;
        const uint64_t rb = base + (off & ~0xFFFFFFFFull);
        const u32x4_t r2 = {(uint32_t)rb, (uint32_t)(rb >> 32) & 0xFFFFu, 0xFFFFFFF0u, 0x00020000u};
        bufl16_lds(r2, (uint32_t)(off & 0xFFFFFFFFull) + lane * 16, 0u, lds0);
        const u32x4_t v = {x, (uint32_t)it, (uint32_t)tid, 7u};
#pragma unroll
        for (int s = 0; s < NST; ++s)
            __builtin_amdgcn_raw_buffer_store_b128(v, h_rsrc, (uint32_t)(((blockIdx.x & 15) * 256 + tid) * 16 + s * 65536), 0, 0);
        if (NST == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (NST == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (NST == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // control: one too many -- the load itself may still fly
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        const uint4 got = slot[tid];
        const uint4 want = src[row * 64 + lane];
        if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) ++nbad;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const size_t bytes = (size_t)6 << 30;          // 6 GiB source: far beyond L2 + MALL
    uint4 *src, *hot; unsigned long long* bad;
    hipMalloc(&src, bytes); hipMalloc(&hot, 1 << 20); hipMalloc(&bad, 8);
    hipMemset(src, 0x5A, bytes);
    // make rows distinguishable
    {
        const size_t n = bytes / 16;
        uint4* h = (uint4*)malloc(1 << 24);
        for (size_t c = 0; c < bytes; c += (1 << 24)) {
            for (size_t i = 0; i < (1 << 24) / 16; ++i) { const uint32_t g = (uint32_t)((c / 16 + i) * 2654435761u); h[i] = make_uint4(g, g ^ 0x1234567u, (uint32_t)(c >> 20), (uint32_t)i); }
            hipMemcpy((char*)src + c, h, 1 << 24, hipMemcpyHostToDevice);
        }
        free(h); (void)n;
    }
    const int iters = 2000;
    for (int nst : {0, 1, 4, 6, 2}) {
        hipMemset(bad, 0, 8);
        if (nst == 0) hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(256), 0, 0, src, bytes / 16, hot, bad, iters);
        if (nst == 1) hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(256), 0, 0, src, bytes / 16, hot, bad, iters);
        if (nst == 4) hipLaunchKernelGGL(probe<4>, dim3(1024), dim3(256), 0, 0, src, bytes / 16, hot, bad, iters);
        if (nst == 2) hipLaunchKernelGGL(probe<2>, dim3(1024), dim3(256), 0, 0, src, bytes / 16, hot, bad, iters);
        if (nst == 6) hipLaunchKernelGGL(probe<6>, dim3(1024), dim3(256), 0, 0, src, bytes / 16, hot, bad, iters);
        hipDeviceSynchronize();
        unsigned long long b = 0;
        hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
        printf("LDS-DMA load, then %d stores, s_waitcnt vmcnt(%d)%s: %llu stale lane-reads of %llu\n", nst, nst == 2 ? 3 : nst,
               nst == 2 ? " (control: one too many)" : "", b, (unsigned long long)1024 * 256 * iters);
    }
    return 0;
}

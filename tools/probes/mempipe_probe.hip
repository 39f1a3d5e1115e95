// Ceilings of one CU's vector-memory path on gfx950, measured the way the GEMM kernels use it (tools/probes, not product code):
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 8 rows x 128 B per instruction) from an L2-resident panel, NW waves of 8 issuing, vmcnt-throttled
//   mode 1: the same while the other 8 - NW waves issue back-to-back MFMA 16x16x32 on random registers
//   mode 2: 16-byte non-temporal stores (8 rows x 128 B per instruction) streaming to a large buffer, NW waves
//   mode 3: mode 0 with the source panel streamed from a large buffer (HBM)
//   mode 4: global_load_dwordx4 to VGPRs from the L2-resident panel
// build: hipcc --offload-arch=gfx950 -O3 -o mempipe_probe mempipe_probe.hip ; run: ./mempipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_wave_base) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* src, char* dst, size_t src_bytes, int nw, int iters, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem) + wave * 16384;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < nw) {
        if constexpr (MODE == 0 || MODE == 1 || MODE == 3) {
            // each instruction: 8 rows x 128 B of a [rows][768 B] panel (K = 384 bf16), rows advance
            // rows is a power of two (32-bit masks: a 64-bit modulo per lane and piece would make the probe VALU-bound)
            const unsigned row_bytes = 768;
            const unsigned rows = (unsigned)src_bytes;  // number of rows, power of two
            unsigned r = ((blockIdx.x * 8 + wave) * 977u) & (rows - 1);
            for (int it = 0; it < iters; ++it) {
                const unsigned koff = (it % 6) * 128 + (lane & 7) * 16;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned row = (r + 8 * q + (lane >> 3)) & (rows - 1);
                    glds16(src + (size_t)row * row_bytes + koff, base + (q & 15) * 1024);
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                r = (r + 128) & (rows - 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 2) {
            // stores: 8 rows x 128 B per instruction into a [rows][2304 B] output (N = 1152 bf16)
            const unsigned row_bytes = 2304;
            const unsigned rows = (unsigned)src_bytes;
            unsigned r = ((blockIdx.x * 8 + wave) * 4099u) & (rows - 1);
            u32x4_t v = {(unsigned)tid, 1u, 2u, 3u};
            for (int it = 0; it < iters; ++it) {
                const unsigned coff = (it % 9) * 256 + (lane & 7) * 16;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned row = (r + 8 * (q >> 1) + (lane >> 3)) & (rows - 1);
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(dst + (size_t)row * row_bytes + (q & 1) * 128 + coff));
                }
                r = (r + 64) & (rows - 1);
            }
        } else if constexpr (MODE == 4) {
            const unsigned row_bytes = 768;
            const unsigned rows = (unsigned)src_bytes;
            unsigned r = ((blockIdx.x * 8 + wave) * 977u) & (rows - 1);
            u32x4_t acc = {0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
                const unsigned koff = (it % 6) * 128 + (lane & 7) * 16;
                u32x4_t x[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned row = (r + 8 * q + (lane >> 3)) & (rows - 1);
                    x[q] = *reinterpret_cast<const u32x4_t*>(src + (size_t)row * row_bytes + koff);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) acc ^= x[q];
                r = (r + 128) & (rows - 1);
            }
            if (acc.x == 0x12345678u) sink[tid] = 1.f;
        }
    } else if constexpr (MODE == 1) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.37f * ((tid * 7 + j * 13) % 17) - 3.f); b[j] = (__bf16)(0.21f * ((tid * 5 + j * 11) % 19) - 2.f); }
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        if (s == 1.2345f) sink[tid] = s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, const char* src, char* dst, size_t bytes, int nw, int iters, int grid, unsigned long long* cyc, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(512), 0, 0, src, dst, bytes, nw, iters, cyc, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double cmax = 0, mfma_cyc = 0;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 8; ++w) { if (w < nw) cmax += (double)h[b * 8 + w] / (grid * nw); else mfma_cyc += (double)h[b * 8 + w] / (grid * (8 - nw)); }
    const double kb_per_cu = (double)nw * iters * 16;  // KB moved per CU
    printf("%-44s nw=%d grid=%3d: %8.1f us  %6.1f GB/s/CU  %6.2f TB/s chip  | memory waves %9.0f cycles -> %5.1f B/clk/CU", name, nw, grid, ms * 1e3,
           kb_per_cu * 1024 / (ms * 1e-3) / 1e9, kb_per_cu * 1024 * grid / (ms * 1e-3) / 1e12, cmax, kb_per_cu * 1024 / cmax);
    if (MODE == 1) printf("  | MFMA waves %9.0f cycles for %d MFMAs each (%.1f cyc/MFMA/SIMD)", mfma_cyc, iters * 64, mfma_cyc / (iters * 64.0) / ((8 - nw) > 4 ? 0.5 : 1.0) / ((8 - nw) > 4 ? 1 : 1));
    printf("\n");
}

int main() {
    const size_t small = 2048;              // rows of 768 B: a 1.5 MB panel, L2-resident  (the kernels take the ROW COUNT, a power of two)
    const size_t big = (size_t)1 << 20;     // 2^20 rows of 768 B = 805 MB: HBM
    const size_t big_st = (size_t)1 << 18;  // store target: 2^18 rows of 2304 B = 604 MB
    char *src, *dst;
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&src, (size_t)1 << 30); hipMalloc(&dst, (size_t)1 << 30); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 4096);
    hipMemset(src, 0x3c, (size_t)1 << 30); hipMemset(dst, 0, (size_t)1 << 30);
    const int it = 400;
    for (int grid : {256, 64}) {
        for (int nw : {8, 4, 2}) run<0>("LDS-DMA from an L2-resident panel", src, dst, small, nw, it, grid, cyc, sink);
        for (int nw : {8, 4, 2}) run<3>("LDS-DMA from an 805 MB buffer (HBM)", src, dst, big, nw, it, grid, cyc, sink);
        for (int nw : {4, 2}) run<1>("LDS-DMA (L2 panel) beside MFMA waves", src, dst, small, nw, it, grid, cyc, sink);
        for (int nw : {8, 4, 2}) run<2>("16-B nt stores, 8 rows x 128 B per instr", src, dst, big_st, nw, it, grid, cyc, sink);
        for (int nw : {8, 4}) run<4>("global_load_dwordx4 to VGPRs (L2 panel)", src, dst, small, nw, it, grid, cyc, sink);
    }
    return 0;
}

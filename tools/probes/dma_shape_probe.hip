// How fast does a CU's LDS-DMA path (global_load_lds_dwordx4, 1 KB per wave instruction) take in an L2-resident panel, as a function of the SHAPE of
// the piece (rows x contiguous bytes per row) and of the row stride?  (tools/probes, not product code.)  gemm_nt* use 8 rows x 128 B at a 768-byte
// stride (~16 cycles per piece, profiles/r03_x3); gemm_tn384 uses 4 rows x 256 B at 3072 / 2304 / 768-byte strides and its waves spend 560-960 cycles
// per stage issuing 4 pieces each (tools/tn_stamp.py).
// build: hipcc --offload-arch=gfx950 -O3 -o dma_shape_probe dma_shape_probe.hip ; run: ./dma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_wave_base) : "memory");
}

// every wave walks its own row range of a [rows][stride] panel: piece = R rows x (1024 / R) bytes; lanes_per_row = 64 / R
__global__ __launch_bounds__(512) void probe(const char* src, unsigned rows_mask, unsigned stride, int R, unsigned col_span, int iters, int perm,
                                             unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem) + wave * 16384;
    const int lpr = 64 / R;                       // lanes per row
    const unsigned lrow = lane / lpr;
    unsigned lcol = (lane % lpr) * 16;
    if (perm) {  // gemm_tn384's swizzle: the four 64-byte groups of a 256-byte row segment permuted by (row & 3)
        const unsigned pc = lane % lpr;
        lcol = ((((pc >> 2) ^ (lrow & 3)) << 2) | (pc & 3)) * 16;
    }
    const unsigned bpr = 1024 / R;                // bytes per row of a piece
    unsigned r = ((blockIdx.x * 8 + wave) * 977u) & rows_mask;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned coff = ((it * bpr) % col_span) + lcol;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned row = (r + R * q + lrow) & rows_mask;
            glds16(src + (size_t)row * stride + coff, base + q * 1024);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        r = (r + 16 * R) & rows_mask;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    char* src;
    unsigned long long* cyc;
    hipMalloc(&src, (size_t)64 << 20);
    hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(src, 0x3c, (size_t)64 << 20);
    const int iters = 400, grid = 256;
    struct Case { const char* name; int R; unsigned stride; unsigned col_span; int perm; } cases[] = {
        {"8 rows x 128 B, stride  768 (NT operands)      ", 8, 768, 768, 0},
        {"8 rows x 128 B, stride 3072                    ", 8, 3072, 768, 0},
        {"8 rows x 128 B, stride 2304                    ", 8, 2304, 768, 0},
        {"4 rows x 256 B, stride  768                    ", 4, 768, 768, 0},
        {"4 rows x 256 B, stride 3072 (TN, Y of fc1)     ", 4, 3072, 768, 0},
        {"4 rows x 256 B, stride 3072, swizzled chunks   ", 4, 3072, 768, 1},
        {"4 rows x 256 B, stride 2304                    ", 4, 2304, 768, 0},
        {"2 rows x 512 B, stride 3072                    ", 2, 3072, 1024, 0},
        {"16 rows x 64 B, stride  768                    ", 16, 768, 768, 0},
    };
    for (auto& c : cases) {
        // panel of ~1.5 MB: rows = power of two with rows * stride <= 4 MB (L2-resident)
        unsigned rows = 1;
        while ((size_t)rows * 2 * c.stride <= ((size_t)3 << 19)) rows *= 2;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, src, rows - 1, c.stride, c.R, c.col_span, iters, c.perm, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h(grid * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double c_avg = 0;
        for (auto v : h) c_avg += (double)v / h.size();
        const double kb = 8.0 * iters * 16;  // KB per CU
        printf("%s rows %5u: %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU  = %5.1f cycles per 1 KB piece and CU\n", c.name, rows, ms * 1e3, kb * 1024 / (ms * 1e-3) / 1e9,
               kb * 1024 / c_avg, c_avg / (8.0 * iters * 16) );
    }
    return 0;
}

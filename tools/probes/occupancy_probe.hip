// How many workgroups of a given LDS size / VGPR count are REALLY co-resident on a gfx950 CU?  (tools/probes, not product code.)
// Every workgroup adds 1 to a counter of its CU (XCC id, SE, CU id from HW_REG_HW_ID / XCC_ID), keeps the largest value it sees while it spins for
// ~30 us, then subtracts 1.  Printed: the maximum over CUs and time, beside what hipOccupancyMaxActiveBlocksPerMultiprocessor predicts.
// build: hipcc --offload-arch=gfx950 -O3 -o occupancy_probe occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int THREADS, int VG>
__global__ __launch_bounds__(THREADS) void probe(int* cu_count, int* cu_max, float* sink, float seed) {
    __shared__ char lds[LDS_BYTES];
    float keep[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = seed * i + threadIdx.x;
    if (seed == 123.f) lds[threadIdx.x] = 1;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7, sh = (hw >> 12) & 1;
    const unsigned id = ((xcc & 15) * 8 + se) * 32 + sh * 16 + cu;  // < 4096
    if (threadIdx.x == 0) {
        const int v = atomicAdd(cu_count + id, 1) + 1;
        atomicMax(cu_max + id, v);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) {  // 100 MHz ticks: 30 us
#pragma unroll
        for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += keep[i];
    if (s == 0.123f) sink[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(cu_count + id, 1);
}

template <int LDS_BYTES, int THREADS, int VG>
static void run(int* d_cnt, int* d_max, float* d_sink) {
    hipMemset(d_cnt, 0, 4096 * 4);
    hipMemset(d_max, 0, 4096 * 4);
    int pred = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&pred, probe<LDS_BYTES, THREADS, VG>, THREADS, 0);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)probe<LDS_BYTES, THREADS, VG>);
    hipLaunchKernelGGL((probe<LDS_BYTES, THREADS, VG>), dim3(4096), dim3(THREADS), 0, 0, d_cnt, d_max, d_sink, 1.0f);
    hipDeviceSynchronize();
    std::vector<int> h(4096);
    hipMemcpy(h.data(), d_max, 4096 * 4, hipMemcpyDeviceToHost);
    int mx = 0, cus = 0;
    long sum = 0;
    for (int v : h)
        if (v > 0) { ++cus; sum += v; mx = v > mx ? v : mx; }
    printf("LDS %6d B  %4d threads  %3d VGPRs (compiler)  runtime predicts %2d workgroups/CU;  measured: max %2d, mean of per-CU maxima %.2f over %d CUs\n", LDS_BYTES,
           THREADS, fa.numRegs, pred, mx, cus ? (double)sum / cus : 0.0, cus);
}

int main() {
    int *d_cnt, *d_max;
    float* d_sink;
    hipMalloc(&d_cnt, 4096 * 4);
    hipMalloc(&d_max, 4096 * 4);
    hipMalloc(&d_sink, 4096);
    run<16384, 256, 8>(d_cnt, d_max, d_sink);
    run<32768, 256, 8>(d_cnt, d_max, d_sink);
    run<40960, 256, 8>(d_cnt, d_max, d_sink);
    run<49152, 256, 8>(d_cnt, d_max, d_sink);
    run<53248, 256, 8>(d_cnt, d_max, d_sink);
    run<65536, 256, 8>(d_cnt, d_max, d_sink);
    run<69632, 256, 8>(d_cnt, d_max, d_sink);
    run<81920, 256, 8>(d_cnt, d_max, d_sink);
    run<49152, 256, 100>(d_cnt, d_max, d_sink);
    run<49152, 256, 120>(d_cnt, d_max, d_sink);
    run<32768, 512, 100>(d_cnt, d_max, d_sink);
    run<49152, 512, 100>(d_cnt, d_max, d_sink);
    return 0;
}

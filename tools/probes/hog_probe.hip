// Measurement aid for tools/hog_probe.py (NOT part of libdcv_hip.so): `wgs` workgroups that each hold a CU slot (256 threads, `lds_bytes`
// of LDS) for about `kcycles` thousand clocks, like a communication kernel running beside the step.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/libhog_probe.so tools/probes/hog_probe.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hog_kernel(int kcycles, unsigned* sink) {
    extern __shared__ char hog_lds[];
    const long long t0 = clock64();
    unsigned x = threadIdx.x;
    while (clock64() - t0 < (long long)kcycles * 1000) {
        __builtin_amdgcn_s_sleep(32);
        x = x * 1664525u + 1013904223u;
    }
    if (x == 0xFFFFFFFFu && sink) sink[0] = hog_lds[threadIdx.x];
}
extern "C" int hog_launch(int wgs, int lds_bytes, int kcycles, void* stream) {
    if (wgs <= 0 || lds_bytes < 0 || kcycles <= 0) return -1;
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, kcycles, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

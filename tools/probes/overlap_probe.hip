// Micro-probe: do the matrix pipe and the vector ALU of one SIMD overlap ACROSS waves (gfx950)?
// Each wave runs ITER iterations of a phase pattern; phases inside a wave depend on each other (as softmax depends on the score MFMAs),
// so only OTHER waves of the SIMD can fill the gaps.  mode 0: MFMA phase only (16 x v_mfma_f32_32x32x16_bf16, two chains of 8),
// mode 1: vector phase only (NV fma + NV/4 exp, dependent on a per-iteration value), mode 2: both, MFMA phase then vector phase.
// Launch with 256 CUs x (waves per SIMD x 4) waves; the host divides cycles by iterations.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = 0.01f * r + lane;
    const unsigned long long t0 = clock64();
    if (MODE == 3) {
        // the same work with the vector phase of iteration i-1 interleaved BY HAND between the MFMAs of iteration i (2 MFMAs, then 1/8 of
        // the vector work): what intra-wave software pipelining could reach
        float seed = v[0];
        for (int it = 0; it < iters; ++it) {
            const float seed_next = acc0[0] + acc1[0];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k < 6) {  // 6 of the 8 gaps carry NV/6 fma (as NV/96 rounds over the 16 chains)
#pragma unroll
                    for (int j = 0; j < NV / 96; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], 0.999f, seed * 1e-9f);
                } else {      // the last two carry the exps
#pragma unroll
                    for (int j = 0; j < NV / 128; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_exp2f(v[r] * 1e-3f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            a[0] = (__bf16)(v[0] * 1e-6f);
            seed = seed_next;
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
            }
        }
        if (MODE != 0) {
            // depends on the MFMA results of THIS iteration (mode 2) so the compiler cannot hoist it above them
            const float seed = (MODE == 2) ? acc0[0] + acc1[0] : v[0];
#pragma unroll
            for (int j = 0; j < NV / 16; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], 0.999f, seed * 1e-9f);
#pragma unroll
            for (int j = 0; j < NV / 64; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_exp2f(v[r] * 1e-3f);
            if (MODE == 2) {  // and the next MFMA phase depends on the vector phase
                a[0] = (__bf16)(v[0] * 1e-6f);
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

extern "C" int run_probe(int mode, int nv, float* out, unsigned long long* cyc, int iters, int blocks, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define L(M, V) hipLaunchKernelGGL((probe<M, V>), dim3(blocks), dim3(256), 0, s, out, cyc, iters)
    if (nv == 128) { if (mode == 0) L(0, 128); else if (mode == 1) L(1, 128); else if (mode == 2) L(2, 128); else L(3, 128); }
    else if (nv == 192) { if (mode == 0) L(0, 192); else if (mode == 1) L(1, 192); else if (mode == 2) L(2, 192); else L(3, 192); }
    else return -1;
    return (int)hipGetLastError();
}

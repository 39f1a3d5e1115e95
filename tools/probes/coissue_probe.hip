// Can a gfx950 SIMD run its matrix pipe and its vector ALU at the same time — from ONE wave's instruction stream, and from TWO / THREE waves that
// are in different phases?  (tools/probes, not product code; the attention kernels' matrix and vector times ADD per wave: DESIGN 3.3.)
//   mode 0: every wave: [ MFMA 32x32x16 bf16 ; K x v_fma_f32 ] x 4 accumulators per iteration              (fine interleave inside one stream)
//   mode 1: the same with K x v_exp_f32
//   mode 2: waves 0-3 (one per SIMD) pure MFMA, the other waves pure vector work (KIND fma / exp), K vector instructions per MFMA of the others
//   mode 3: every wave: burst of 8 MFMAs, then burst of 8 K vector instructions (E of every 8 are v_exp) — the phase structure of the attention loops;
//           waves unsynchronised (no barrier), 1 / 2 / 3 waves per SIMD
//   mode 4: mode 3 with one s_barrier per iteration (what a shared LDS ring does to the phases)
//   mode 5: mode 3 with the attention loops' DEPENDENCIES: every vector instruction reads an accumulator of the burst's MFMAs, and the next
//           iteration's MFMAs read an operand the vector burst produced (MFMA -> softmax -> MFMA); waves unsynchronised
// Output: cycles (s_memtime) per iteration and per wave, against the issue-time model: MFMA 32 cycles, v_fma 4, v_exp 8 (or 16).
// build: hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip ; run: ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <int MODE, int K, int E, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(int iters, unsigned long long* cyc, float* sink, float seed) {
    __shared__ char hold[100 * 1024];  // one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (seed == 12345.f) hold[tid] = 1;  // keeps the allocation
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int u = 0; u < 4; ++u)
        for (int i = 0; i < 16; ++i) acc[u][i] = seed * (u + i);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed + 0.01f * (lane + i);
    float c1 = 0.999f + seed * 1e-9f, c2 = 1e-7f;
    const bool mfma_wave = (MODE != 2) || wave < 4;
    const bool vec_wave = (MODE != 2) || wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                MFMA(acc[u]);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if constexpr (MODE == 0) VFMA(f[k & 7]);
                    else VEXP(f[k & 7]);
                }
            }
        } else if constexpr (MODE == 2) {
            if (mfma_wave) {
#pragma unroll
                for (int u = 0; u < 4; ++u) MFMA(acc[u]);
            }
            if (vec_wave) {
#pragma unroll
                for (int k = 0; k < 4 * K; ++k) {
                    if constexpr (E == 0) VFMA(f[k & 7]);
                    else VEXP(f[k & 7]);
                }
            }
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 8; ++u) MFMA(acc[u & 3]);
#pragma unroll
            for (int k = 0; k < 8 * K; ++k) {
                if ((k % 8) < E) VEXP(f[k & 7]);
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k & 7]) : "v"(acc[k & 3][(k >> 2) & 15]), "v"(c2));
            }
            unsigned bw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(bw[i]) : "v"(f[2 * i]), "v"(f[2 * i + 1]));
            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
            const u32x4v bv = {bw[0], bw[1], bw[2], bw[3]};
            b = __builtin_bit_cast(bf16x8, bv);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) MFMA(acc[u & 3]);
#pragma unroll
            for (int k = 0; k < 8 * K; ++k) {
                if ((k % 8) < E) VEXP(f[k & 7]);
                else VFMA(f[k & 7]);
            }
            if constexpr (MODE == 4) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int u = 0; u < 4; ++u)
        for (int i = 0; i < 16; ++i) s += acc[u][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 0.12345f) sink[tid] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

// mode 6 (round 5): mode 5's dependent bursts grown towards the attention forward loop one feature at a time — FEAT bit 0: the A operands
// of the 8 MFMAs come from LDS at the loop's rate (4 ds_read_b128 + 8 ds_read_b64_tr_b16 per 8 MFMAs, swizzle-free conflict-free addresses); bit 1: one s_barrier per
// two iterations (= per key tile) inside each four-wave workgroup; bit 2: four LDS-DMA pieces per wave and tile from an L2-resident buffer behind a counted vmcnt.
// Workgroups of 256 threads; WPS = 1 / 2 / 3 of them per CU (LDS size), i.e. 1 / 2 / 3 waves per SIMD with INDEPENDENT barriers, as the real kernel runs.
typedef __attribute__((ext_vector_type(4))) short s16x4v;
template <int FEAT, int LDSKB>
__global__ __launch_bounds__(256) void probe6(int iters, unsigned long long* cyc, float* sink, float seed, const char* gbuf) {
    __shared__ __attribute__((aligned(16))) char lds[LDSKB * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 32 * 1024 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 63);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int u = 0; u < 4; ++u)
        for (int i = 0; i < 16; ++i) acc[u][i] = seed * (u + i);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed + 0.01f * (lane + i);
    float c2 = 1e-7f;
    const int rbase = (lane & 31) * 128 + ((lane >> 5) << 4);  // row reads: lane -> row, 16-byte chunk (conflict-free without a swizzle at this stride pattern is not the point: same instruction count)
    const int tbase = ((lane >> 5) * 4 + ((lane & 15) >> 2)) * 128 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int so = (it & 1) * 16384;
        if constexpr (FEAT & 4) {
            if ((it & 1) == 0) {  // one tile = two iterations: four 1 KB pieces per wave, waited for one tile later
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        }
        if constexpr (FEAT & 2) {
            if ((it & 1) == 0) __builtin_amdgcn_s_barrier();
        }
        if constexpr (FEAT & 4) {
            if ((it & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const char* src = gbuf + ((size_t)((blockIdx.x * 4 + wave) & 255) * 4096 + j * 1024 + lane * 16);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(lds_base + 32768 + (wave * 4 + j) * 1024) : "memory");
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (FEAT & 1) {
                bf16x8 af;
                if (u < 4) {
                    af = *reinterpret_cast<const bf16x8*>(lds + so + rbase + u * 4096 % 16384);
                } else {
                    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(lds + so + tbase + (u - 4) * 2048));
                    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(lds + so + tbase + (u - 4) * 2048 + 1024));
                    typedef short s16x8v __attribute__((ext_vector_type(8)));
                    const s16x8v j8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af = __builtin_bit_cast(bf16x8, j8);
                }
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(af), "v"(b));
            } else {
                MFMA(acc[u & 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8 * 14; ++k) {  // the forward's mix: 28 v_exp + 84 v_fma per 8 MFMAs, every one reading an accumulator
            if ((k % 8) < 2) VEXP(f[k & 7]);
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k & 7]) : "v"(acc[k & 3][(k >> 2) & 15]), "v"(c2));
        }
        unsigned bw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(bw[i]) : "v"(f[2 * i]), "v"(f[2 * i + 1]));
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        const u32x4v bv = {bw[0], bw[1], bw[2], bw[3]};
        b = __builtin_bit_cast(bf16x8, bv);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int u = 0; u < 4; ++u)
        for (int i = 0; i < 16; ++i) s += acc[u][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 0.12345f) sink[tid] = s;
    if (lane == 0 && blockIdx.x < 256) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

static unsigned long long* d_cyc;
static float* d_sink;
static char* d_gbuf;

template <int FEAT, int LDSKB>
static void run6(const char* what, int wps) {
    const int grid = 256 * wps, iters = 2000;
    std::vector<unsigned long long> h(256 * 16);
    hipMemset(d_cyc, 0, 256 * 16 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe6<FEAT, LDSKB>), dim3(grid), dim3(256), 0, 0, 200, d_cyc, d_sink, 1.0f, d_gbuf);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe6<FEAT, LDSKB>), dim3(grid), dim3(256), 0, 0, iters, d_cyc, d_sink, 1.0f, d_gbuf);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_cyc, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int g = 0; g < 256; ++g)
        for (int w = 0; w < 4; ++w) c += (double)h[g * 16 + w] / iters;
    c /= 1024;
    printf("%-96s %d wave(s)/SIMD  %8.1f cyc/iter per wave  (serial %5.0f overlap %5.0f)  %.2f GHz  %7.1f us\n", what, wps, c, wps * (256 + 560.0), wps * 560.0, c * iters / (ms * 1e6), ms * 1e3);
}

template <int MODE, int K, int E, int THREADS>
static void run(const char* what, double model_serial, double model_overlap) {
    const int grid = 256, iters = 2000;
    std::vector<unsigned long long> h(grid * 16);
    hipMemset(d_cyc, 0, grid * 16 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, K, E, THREADS>), dim3(grid), dim3(THREADS), 0, 0, 200, d_cyc, d_sink, 1.0f);  // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, K, E, THREADS>), dim3(grid), dim3(THREADS), 0, 0, iters, d_cyc, d_sink, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_cyc, grid * 16 * 8, hipMemcpyDeviceToHost);
    const int nw = THREADS / 64;
    double lo[2] = {0, 0};
    int cnt[2] = {0, 0};
    for (int g = 0; g < grid; ++g)
        for (int w = 0; w < nw; ++w) {
            const int grp = (MODE == 2 && w >= 4) ? 1 : 0;
            lo[grp] += (double)h[g * 16 + w] / iters;
            ++cnt[grp];
        }
    const double c0 = lo[0] / cnt[0], c1 = cnt[1] ? lo[1] / cnt[1] : 0.0;
    const double ghz = c0 * iters / (ms * 1e6);
    if (MODE == 2)
        printf("%-86s  mfma waves %8.1f cyc/iter   vector waves %8.1f   (serial %6.0f overlap %6.0f)  %.2f GHz\n", what, c0, c1, model_serial, model_overlap, ghz);
    else
        printf("%-86s  %8.1f cyc/iter   (serial %6.0f overlap %6.0f)  %.2f GHz\n", what, c0, model_serial, model_overlap, ghz);
}

int main() {
    hipMalloc(&d_cyc, 256 * 16 * 8);
    hipMalloc(&d_sink, 4096 * 4);
    printf("# models: MFMA 32x32x16 = 32 cycles of the matrix pipe, v_fma = 4, v_exp = 8 issue cycles; 'serial' = sum, 'overlap' = max (per SIMD)\n");
    printf("## mode 0: one wave per SIMD, [MFMA ; K x v_fma] x 4 per iteration\n");
#define R0(K) run<0, K, 0, 256>("  K = " #K " v_fma per MFMA", 4 * (32 + 4.0 * K), 4 * (32 > 4 + 4.0 * K ? 32 : 4 + 4.0 * K))
    R0(0); R0(2); R0(4); R0(6); R0(7); R0(8); R0(10); R0(14); R0(18);
    printf("## mode 1: one wave per SIMD, [MFMA ; K x v_exp] x 4 per iteration\n");
#define R1(K) run<1, K, 0, 256>("  K = " #K " v_exp per MFMA", 4 * (32 + 8.0 * K), 4 * (32 > 4 + 8.0 * K ? 32 : 4 + 8.0 * K))
    R1(1); R1(2); R1(3); R1(4); R1(6); R1(8);
    printf("## mode 0/1 with two waves per SIMD running the same interleaved stream (512 threads)\n");
    run<0, 7, 0, 512>("  2 waves: K = 7 v_fma per MFMA", 2 * 4 * (32 + 28.0), 2 * 4 * 32.0);
    run<0, 14, 0, 512>("  2 waves: K = 14 v_fma per MFMA", 2 * 4 * (32 + 56.0), 2 * 4 * 60.0);
    run<1, 4, 0, 512>("  2 waves: K = 4 v_exp per MFMA", 2 * 4 * (32 + 32.0), 2 * 4 * 36.0);
    printf("## mode 2: waves 0-3 pure MFMA (4 per iteration), waves 4-7 pure vector work (4 K per iteration)\n");
    run<2, 0, 0, 256>("  MFMA waves alone", 128, 128);
    run<2, 8, 0, 512>("  + 32 v_fma per iteration on the second wave of the SIMD", 128 + 128, 128);
    run<2, 16, 0, 512>("  + 64 v_fma", 128 + 256, 256);
    run<2, 4, 1, 512>("  + 16 v_exp", 128 + 128, 128);
    run<2, 8, 1, 512>("  + 32 v_exp", 128 + 256, 256);
    printf("## mode 3: bursts — 8 MFMAs, then 8 K vector instructions (E of 8 are v_exp); no barrier\n");
    run<3, 7, 0, 256>("  1 wave/SIMD, 56 v_fma", 256 + 224, 256);
    run<3, 7, 0, 512>("  2 waves/SIMD, 56 v_fma", 2 * (256 + 224), 2 * 256.0);
    run<3, 7, 0, 768>("  3 waves/SIMD, 56 v_fma", 3 * (256 + 224), 3 * 256.0);
    run<3, 14, 2, 256>("  1 wave/SIMD, 28 v_exp + 84 v_fma (the forward's mix: 560 vector cycles per 256)", 256 + 560, 560);
    run<3, 14, 2, 512>("  2 waves/SIMD, same", 2 * (256 + 560), 2 * 560.0);
    run<3, 14, 2, 768>("  3 waves/SIMD, same", 3 * (256 + 560), 3 * 560.0);
    printf("## mode 4: the same bursts with one s_barrier per iteration (phases aligned)\n");
    run<4, 7, 0, 512>("  2 waves/SIMD, 56 v_fma", 2 * (256 + 224), 2 * 256.0);
    run<4, 14, 2, 512>("  2 waves/SIMD, 28 v_exp + 84 v_fma", 2 * (256 + 560), 2 * 560.0);
    run<4, 14, 2, 768>("  3 waves/SIMD, 28 v_exp + 84 v_fma", 3 * (256 + 560), 3 * 560.0);
    printf("## mode 5: the bursts of mode 3 with MFMA -> vector -> MFMA dependencies (no barrier)\n");
    run<5, 14, 2, 256>("  1 wave/SIMD, 28 v_exp + 84 v_fma on the accumulators", 256 + 560, 560);
    run<5, 14, 2, 512>("  2 waves/SIMD, same", 2 * (256 + 560), 2 * 560.0);
    run<5, 14, 2, 768>("  3 waves/SIMD, same", 3 * (256 + 560), 3 * 560.0);
    run<5, 7, 0, 512>("  2 waves/SIMD, 56 v_fma on the accumulators", 2 * (256 + 224), 2 * 256.0);
    run<5, 7, 0, 768>("  3 waves/SIMD, 56 v_fma on the accumulators", 3 * (256 + 224), 3 * 256.0);
    hipMalloc(&d_gbuf, 256 * 4096);
    hipMemset(d_gbuf, 0, 256 * 4096);
    printf("## mode 6: mode 5 (dependent bursts, the forward's mix) grown towards the real loop; 256-thread workgroups, 1 / 2 / 3 per CU, barriers per workgroup\n");
#define R6(F, what) run6<F, 100>(what, 1); run6<F, 64>(what, 2); run6<F, 48>(what, 3);
    R6(0, "  registers only (= mode 5)");
    R6(1, "  + A operands from LDS (4 ds_read_b128 + 8 ds_read_b64_tr_b16 per 8 MFMAs)");
    R6(3, "  + one s_barrier per tile (two iterations) per workgroup");
    R6(7, "  + four LDS-DMA pieces per wave and tile, counted vmcnt");
    R6(6, "  barrier + DMA without the LDS operand reads");
    return 0;
}

// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of libdcv_hip.so.
// wave = 64 lanes; MFMA bf16 32x32x16; LDS tiles are XOR-swizzled so that the same image
// serves ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#include "../../include/dcv.h"

#define DCV_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return DCV_ERR_LAUNCH;        \
    } while (0)

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

union U128 {
    uint4 u;
    bf16x8 h;
    bf16x4 q[2];
    uint2 d[2];
};

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) {
    U128 x;
    x.u = v;
    return x.h;
}
__device__ __forceinline__ uint4 as_uint4(bf16x8 v) {
    U128 x;
    x.h = v;
    return x.u;
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);

}

// 16x16x32: lane l holds A[row l&15][k = 8(l>>4) + j], B[k = 8(l>>4) + j][col l&15]; the 4 accumulator registers of lane l are
// rows 4(l>>4) .. +3 of column l&15.  Same FLOPs per cycle as 32x32x16, but the chip — clock-limited by power on this workload —
// holds a higher clock on this shape (cdna guide rule 28; measured here: the NT GEMMs 7 % faster with the same operand traffic).
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 32x32 accumulator: register r of lane l holds row acc_row(r, l>>5), column l&31.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// pack registers 8s..8s+7 of a 32x32 f32 accumulator into the bf16 operand fragment of k-step s
// of a following 32x32x16 MFMA that sums over the accumulator's ROW index.  The k order inside the
// step is permuted: element j of lane half h is accumulator row 16s + 8(j>>2) + 4h + (j&3).
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s + j];
    return r;
}

// ---- 64-column bf16 LDS tile (128-byte rows, 8 chunks of 16 B) ------------------------------
// chunk' = chunk ^ f(row), f(row) = k ^ ((k & 1) << 2), k = (row >> 1) & 7.
//  * ds_read_b128 row reads (lane = row, same chunk): 16 rows of a lane group hit 16 distinct
//    16-B slots of the 256-B bank row -> conflict-free.
//  * ds_read_b64_tr_b16 (4 rows x 64 B per 32-lane half): rows q and q+2 land in different 64-B
//    halves -> conflict-free.
__device__ __forceinline__ int swz64(int row) {
    int k = (row >> 1) & 7;
    return k ^ ((k & 1) << 2);
}
// The same tile read ONLY by rows with 16x16x32 fragments (the NT GEMMs): lane (row l&15, k-chunk l>>4) — a ds_read_b128 lane group
// holds all 16 rows once, half of them at chunk c and half at c + 1, and rows r and r ^ 2 always fall in the same half, so the
// plain chunk ^ (row >> 1) swizzle is conflict-free.
__device__ __forceinline__ int swz64n(int row) { return (row >> 1) & 7; }
// byte offset of element (row, col) in a [rows][64] bf16 tile
__device__ __forceinline__ int lds64_off(int row, int col) {
    return row * 128 + ((((col >> 3) ^ swz64(row)) << 4) | ((col & 7) << 1));
}

// ---- 128-column bf16 LDS tile (256-byte rows, 16 chunks) for transposed reads only ----------
// 64-B segment index (4 per row) is XORed with (row & 3): the 4 rows of a tr read go to 4
// different segments of the bank row.
__device__ __forceinline__ int lds128_off(int row, int col) {
    int chunk = (col >> 3) ^ ((row & 3) << 2);
    return row * 256 + ((chunk << 4) | ((col & 7) << 1));
}

// transposed read: a 16-lane group reads a 4-row x 16-column block; lane i of the group passes
// the address of (row i>>2, cols 4*(i&3)..+3) and receives column i, rows 0..3.
__device__ __forceinline__ bf16x4 lds_tr_read(const char* base_generic, int byte_off) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base_generic + byte_off));
    union {
        s16x4 s;
        bf16x4 b;
    } u;
    u.s = t;
    return u.b;
}

__device__ __forceinline__ bf16x8 join4(bf16x4 lo, bf16x4 hi) {
    U128 x;
    x.q[0] = lo;
    x.q[1] = hi;
    return x.h;
}

__device__ __forceinline__ uint4 lds_read128(const char* base, int byte_off) {
    return *reinterpret_cast<const uint4*>(base + byte_off);
}
__device__ __forceinline__ void lds_write128(char* base, int byte_off, uint4 v) {
    *reinterpret_cast<uint4*>(base + byte_off) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    union {
        bf16x4 v;
        uint2 u;
    } x;
    x.v[0] = (bf16_t)a;
    x.v[1] = (bf16_t)b;
    x.v[2] = (bf16_t)c;
    x.v[3] = (bf16_t)d;
    return x.u;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// ---- LDS-DMA (global_load_lds_dwordx4): destination is wave-uniform base + lane*16; any swizzle is applied to the
// per-lane SOURCE address (linear destination + permuted source + the same permutation on the read).
// Issued through inline asm on purpose: hipcc then neither counts it in its own vmcnt bookkeeping nor fences the
// following ds_reads with vmcnt(0) (which it does for the builtin and which would serialise the ring); the
// kernel waits with hand-counted s_waitcnt vmcnt(N) + a raw s_barrier.  M0 (the LDS base) is saved and restored
// inside the statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(const bf16_t* g, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_wave_base)
        : "memory");
}
// same DMA with a wave-uniform base (SGPR pair) and a per-lane 32-bit byte offset: a loop that walks tiles advances the
// base with scalar adds and keeps the lane offsets loop-invariant (no per-tile vector address arithmetic)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_wave_base)
        : "memory");
}
// 4-byte-per-lane form (64 consecutive floats per wave instruction)
__device__ __forceinline__ void glds4s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_wave_base)
        : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch, speed only).  Gives each
// XCD a contiguous range of logical ids.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    int q = n >> 3, r = n & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// ---- deterministic mode (the reference trains with cudnn.deterministic = True, utils.py:394-401) -------------------------------
// Kernels that combine partial sums of several workgroups have a second form that STORES the partials to a caller-supplied
// workspace instead of adding them atomically; this kernel then adds them to the destination in index order — a fixed summation
// order whatever the dispatch order was, so two runs on the same inputs are bit-identical.  Column c of a part goes to
// dstA[(c / QA) * ldA + c % QA] for c < nA and to dstB[c - nA] for nA <= c < nA + nB.  VEC: four columns per thread (all of nA, QA,
// ldA, nB, part_stride multiples of 4 and 16-byte aligned pointers).
template <bool VEC>
static __global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ part, int nparts, long part_stride,
                                                                float* __restrict__ dstA, long nA, int QA, long ldA,
                                                                float* __restrict__ dstB, long nB) {
    constexpr int W = VEC ? 4 : 1;
    const long nv = (nA + nB) / W;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
        const long c = v * W;
        float s[W];
#pragma unroll
        for (int e = 0; e < W; ++e) s[e] = 0.f;
        const float* src = part + c;
        int k = 0;
        // eight parts' loads in flight, added in part order (a thread walking its parts one load at a time ran at ~2 TB/s)
        for (; k + 8 <= nparts; k += 8) {
            if constexpr (VEC) {
                float4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(src + (size_t)(k + j) * part_stride);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[0] += t[j].x; s[1] += t[j].y; s[2] += t[j].z; s[3] += t[j].w; }
            } else {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = src[(size_t)(k + j) * part_stride];
#pragma unroll
                for (int j = 0; j < 8; ++j) s[0] += t[j];
            }
        }
        for (; k < nparts; ++k) {
            if constexpr (VEC) {
                const float4 t = *reinterpret_cast<const float4*>(src + (size_t)k * part_stride);
                s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
            } else {
                s[0] += src[(size_t)k * part_stride];
            }
        }
        float* d = (c < nA) ? dstA + (c / QA) * ldA + (c % QA) : dstB + (c - nA);
#pragma unroll
        for (int e = 0; e < W; ++e) d[e] += s[e];
    }
}
// Same sum for MANY parts of FEW columns (LayerNorm: 1024 parts x 768 columns; one thread per column group would walk all parts alone):
// a workgroup owns 4 column groups; part lane pl = thread / 4 adds parts pl, pl + 64, pl + 128, ... in increasing order (eight loads in flight),
// the 64 lane sums are then added in lane order by the threads of lane 0.  The association order is fixed by (nparts) alone: deterministic.
constexpr int DET_TALL_CG = 4, DET_TALL_PL = 256 / DET_TALL_CG;
template <bool VEC>
static __global__ __launch_bounds__(256) void det_reduce_tall_kernel(const float* __restrict__ part, int nparts, long part_stride,
                                                                     float* __restrict__ dstA, long nA, int QA, long ldA,
                                                                     float* __restrict__ dstB, long nB) {
    constexpr int W = VEC ? 4 : 1;
    __shared__ float red[DET_TALL_PL][DET_TALL_CG][W];
    const int pl = threadIdx.x / DET_TALL_CG, cg = threadIdx.x % DET_TALL_CG;
    const long nv = (nA + nB) / W;
    const long v = (long)blockIdx.x * DET_TALL_CG + cg;
    float s[W];
#pragma unroll
    for (int e = 0; e < W; ++e) s[e] = 0.f;
    if (v < nv) {
        const float* src = part + v * W;
        int k = pl;
        for (; k + 7 * DET_TALL_PL < nparts; k += 8 * DET_TALL_PL) {
            if constexpr (VEC) {
                float4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(src + (size_t)(k + j * DET_TALL_PL) * part_stride);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[0] += t[j].x; s[1] += t[j].y; s[2] += t[j].z; s[3] += t[j].w; }
            } else {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = src[(size_t)(k + j * DET_TALL_PL) * part_stride];
#pragma unroll
                for (int j = 0; j < 8; ++j) s[0] += t[j];
            }
        }
        for (; k < nparts; k += DET_TALL_PL) {
            if constexpr (VEC) {
                const float4 t = *reinterpret_cast<const float4*>(src + (size_t)k * part_stride);
                s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
            } else {
                s[0] += src[(size_t)k * part_stride];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < W; ++e) red[pl][cg][e] = s[e];
    __syncthreads();
    if (pl == 0 && v < nv) {
        const long c = v * W;
        float* d = (c < nA) ? dstA + (c / QA) * ldA + (c % QA) : dstB + (c - nA);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < DET_TALL_PL; ++l) t += red[l][cg][e];
            d[e] += t;
        }
    }
}
// Several of those sums in ONE launch (the grouped weight-gradient GEMM: up to 8 products x (weight, bias) = 16 jobs, each a few microseconds
// of work behind a launch): workgroup b belongs to the job j with wg_start[j] <= b < wg_start[j + 1]; inside a job exactly det_reduce_kernel<true>.
constexpr int DET_MULTI_MAX = 16;
struct DetJobs {
    int n;
    int wg_start[DET_MULTI_MAX + 1];
    const float* part[DET_MULTI_MAX];
    int nparts[DET_MULTI_MAX];
    long stride[DET_MULTI_MAX];
    float* dst[DET_MULTI_MAX];
    long nv[DET_MULTI_MAX];  // float4 columns
    int Q[DET_MULTI_MAX];
    long ld[DET_MULTI_MAX];
};
static __global__ __launch_bounds__(256) void det_reduce_multi_kernel(DetJobs jb) {
    int j = 0;
    while (j + 1 < jb.n && (int)blockIdx.x >= jb.wg_start[j + 1]) ++j;
    const long v = (long)((int)blockIdx.x - jb.wg_start[j]) * 256 + threadIdx.x;
    if (v >= jb.nv[j]) return;
    const long c = v * 4;
    const float* src = jb.part[j] + c;
    const long stride = jb.stride[j];
    const int nparts = jb.nparts[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 8 <= nparts; k += 8) {
        float4 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<const float4*>(src + (size_t)(k + i) * stride);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 += t[i].x; s1 += t[i].y; s2 += t[i].z; s3 += t[i].w; }
    }
    for (; k < nparts; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(src + (size_t)k * stride);
        s0 += t.x; s1 += t.y; s2 += t.z; s3 += t.w;
    }
    float* d = jb.dst[j] + (c / jb.Q[j]) * jb.ld[j] + (c % jb.Q[j]);
    d[0] += s0; d[1] += s1; d[2] += s2; d[3] += s3;
}
// adds job (part, nparts, stride) -> dst[(c / Q) * ld + c % Q], c < n, to jb; false when the job does not fit the vector form or the table
static inline bool det_jobs_add(DetJobs& jb, const float* part, int nparts, long stride, float* dst, long n, int Q, long ld) {
    if (jb.n >= DET_MULTI_MAX || (n & 3) || (Q & 3) || (ld & 3) || (stride & 3) || ((uintptr_t)part & 15) || ((uintptr_t)dst & 15)) return false;
    const int j = jb.n++;
    jb.part[j] = part; jb.nparts[j] = nparts; jb.stride[j] = stride; jb.dst[j] = dst; jb.nv[j] = n / 4; jb.Q[j] = Q; jb.ld[j] = ld;
    jb.wg_start[j + 1] = jb.wg_start[j] + (int)((n / 4 + 255) / 256);
    return true;
}
static inline bool det_reduce_multi(const DetJobs& jb, hipStream_t s) {
    if (jb.n <= 0) return true;
    hipLaunchKernelGGL(det_reduce_multi_kernel, dim3((unsigned)jb.wg_start[jb.n]), dim3(256), 0, s, jb);
    return hipGetLastError() == hipSuccess;
}
// launches the reduction; returns false when the launch failed
static inline bool det_reduce(const float* part, int nparts, long part_stride, float* dstA, long nA, int QA, long ldA, float* dstB, long nB,
                              hipStream_t s) {
    const bool vec = !(nA & 3) && !(QA & 3) && !(ldA & 3) && !(nB & 3) && !(part_stride & 3) && !((uintptr_t)part & 15) &&
                     !((uintptr_t)dstA & 15) && !((uintptr_t)dstB & 15);
    const long nv = (nA + nB) / (vec ? 4 : 1);
    if (nparts >= 32 && nv <= 16384) {  // many parts, few columns
        const unsigned grid = (unsigned)((nv + DET_TALL_CG - 1) / DET_TALL_CG);
        if (vec) hipLaunchKernelGGL(det_reduce_tall_kernel<true>, dim3(grid), dim3(256), 0, s, part, nparts, part_stride, dstA, nA, QA, ldA, dstB, nB);
        else hipLaunchKernelGGL(det_reduce_tall_kernel<false>, dim3(grid), dim3(256), 0, s, part, nparts, part_stride, dstA, nA, QA, ldA, dstB, nB);
        return hipGetLastError() == hipSuccess;
    }
    long grid = (nv + 255) / 256;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    if (vec) hipLaunchKernelGGL(det_reduce_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, part, nparts, part_stride, dstA, nA, QA, ldA, dstB, nB);
    else hipLaunchKernelGGL(det_reduce_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, part, nparts, part_stride, dstA, nA, QA, ldA, dstB, nB);
    return hipGetLastError() == hipSuccess;
}

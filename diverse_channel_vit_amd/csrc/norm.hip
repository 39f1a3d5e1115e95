// LayerNorm forward / backward over the fp32 residual stream (HBM-bound, one wave per token row).
// Replaces nn.LayerNorm(eps=1e-6) in Block.forward (models/vit.py:361,374,384,398) and the final
// norm (models/dichavit.py:651).
//   fwd : x f32 [M,D] -> u bf16 [M,D] (the GEMM A operand, kept for backward), mean/rstd f32 [M]
//   bwd : dx_out = dx_in + LN'(du)  (residual add fused), bf16 copy of dx_out for the next GEMMs,
//         dgamma/dbeta accumulated per workgroup then one atomic per column.
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

#ifndef DCV_LN_NT
#define DCV_LN_NT 1
#endif
#ifndef DCV_LN_FWD_ROWS
#define DCV_LN_FWD_ROWS 1
#endif
#ifndef DCV_LN_NT_ST
#define DCV_LN_NT_ST 0  // measured: non-temporal stores +1.5..2 us on both kernels (and the next GEMM re-reads these rows)
#endif
#if DCV_LN_NT_ST
#define LN_STORE(p, v) __builtin_nontemporal_store(v, p)
#else
#define LN_STORE(p, v) (*(p) = (v))
#endif
#if DCV_LN_NT
#define LN_LOAD(p) __builtin_nontemporal_load(p)
#else
#define LN_LOAD(p) (*(p))
#endif

constexpr int MAXV_MAX = 4;  // float4 per lane: D <= 64*4*4 = 1024; the kernels are instantiated for 2 (D <= 512) and 4

template <bool OUT_F32, int MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ u,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps, long x_row_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;  // float4 per row
    // Measured (tools/ab_bench.py, M = 100 416, D = 384): two rows per wave and iteration (twice the bytes in flight) 45.9 us,
    // one row 44.0-44.9 us, non-temporal loads +1 us: the kernel sits at 5.3 TB/s for other reasons than latency.
    auto load_row = [&](int row, f32x4(&v)[MAXV]) {
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * x_row_stride);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            v[i] = (c < nv) ? reinterpret_cast<const f32x4*>(xr)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto finish_row = [&](int row, const f32x4(&v)[MAXV]) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        const float mu = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                q += a * a + b * b + cc * cc + d * d;
            }
        }
        const float rs = rsqrtf(wave_sum(q) / D + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                const float4 bb = reinterpret_cast<const float4*>(beta)[c];
                float o0 = (v[i].x - mu) * rs * g.x + bb.x, o1 = (v[i].y - mu) * rs * g.y + bb.y;
                float o2 = (v[i].z - mu) * rs * g.z + bb.z, o3 = (v[i].w - mu) * rs * g.w + bb.w;
                if constexpr (OUT_F32) {
                    LN_STORE(reinterpret_cast<f32x4*>((float*)u + (size_t)row * D) + c, (f32x4{o0, o1, o2, o3}));
                } else {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const uint2 pk = pack4_bf16(o0, o1, o2, o3);
                    LN_STORE(reinterpret_cast<u32x2*>((bf16_t*)u + (size_t)row * D) + c, (u32x2{pk.x, pk.y}));
                }
            }
        }
    };
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
#if DCV_LN_FWD_ROWS == 2
    for (; row + stride < M; row += 2 * stride) {
        f32x4 va[MAXV], vb[MAXV];
        load_row(row, va);
        load_row(row + stride, vb);
        finish_row(row, va);
        finish_row(row + stride, vb);
    }
#endif
    for (; row < M; row += stride) {
        f32x4 va[MAXV];
        load_row(row, va);
        finish_row(row, va);
    }
}

template <bool DU_F32, int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ du, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* dx_in, float* dx_out,
                                                     bf16_t* __restrict__ dx_bf16, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int D, long x_row_stride,
                                                     long dx_row_stride, float* __restrict__ part, const float* __restrict__ bscale,
                                                     int rows_per_sample) {
    // part != nullptr (deterministic mode): workgroup b stores its column sums to part[b][2][D] instead of adding them atomically
    __shared__ float red[4][2][MAXV * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    float4 ag[MAXV], ab[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // everything a row needs from HBM (x, du, the incoming residual gradient, its two statistics) is loaded up front, with
    // non-temporal loads: 134 -> 124 us at M = 100 416, D = 384 (the row count per iteration made no difference)
    struct RowIn {
        f32x4 xv[MAXV], d[MAXV], pin[MAXV];
        float mu, rs;
    };
    auto load_row = [&](int row, RowIn& r) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * x_row_stride);
        r.mu = mean[row];
        r.rs = rstd[row];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            r.xv[i] = r.d[i] = r.pin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nv) {
                r.xv[i] = LN_LOAD(xr + c);
                if constexpr (DU_F32) {
                    r.d[i] = LN_LOAD(reinterpret_cast<const f32x4*>((const float*)du + (size_t)row * D) + c);
                } else {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 raw = LN_LOAD(reinterpret_cast<const u32x2*>((const bf16_t*)du + (size_t)row * D) + c);
                    r.d[i] = f32x4{bf16_bits_to_f32(raw.x & 0xffff), bf16_bits_to_f32(raw.x >> 16), bf16_bits_to_f32(raw.y & 0xffff),
                                   bf16_bits_to_f32(raw.y >> 16)};
                }
                if (dx_in) r.pin[i] = LN_LOAD(reinterpret_cast<const f32x4*>(dx_in + (size_t)row * dx_row_stride) + c);
            }
        }
    };
    auto finish_row = [&](int row, const RowIn& in) {
        const float mu = in.mu, rs = in.rs;
        float4 xh[MAXV], g[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            xh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nv) {
                const f32x4 xv = in.xv[i], d = in.d[i];
                const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
                s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
                ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
            }
        }
        const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                f32x4 r = {rs * (g[i].x - m1 - xh[i].x * m2) + in.pin[i].x, rs * (g[i].y - m1 - xh[i].y * m2) + in.pin[i].y,
                           rs * (g[i].z - m1 - xh[i].z * m2) + in.pin[i].z, rs * (g[i].w - m1 - xh[i].w * m2) + in.pin[i].w};
                LN_STORE(reinterpret_cast<f32x4*>(dx_out + (size_t)row * dx_row_stride) + c, r);
                if (dx_bf16) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    if (bscale) {  // the copy feeds the residual branch below: DropPath's per-sample factor multiplies that branch's gradient
                        const float sc = bscale[row / rows_per_sample];
                        r.x *= sc; r.y *= sc; r.z *= sc; r.w *= sc;
                    }
                    const uint2 pk = pack4_bf16(r.x, r.y, r.z, r.w);
                    LN_STORE(reinterpret_cast<u32x2*>(dx_bf16 + (size_t)row * D) + c, (u32x2{pk.x, pk.y}));
                }
            }
        }
    };
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    for (; row + stride < M; row += 2 * stride) {
        RowIn ra, rb;
        load_row(row, ra);
        load_row(row + stride, rb);
        finish_row(row, ra);
        finish_row(row + stride, rb);
    }
    if (row < M) {
        RowIn ra;
        load_row(row, ra);
        finish_row(row, ra);
    }
    // column reduction across the 4 waves, then one atomic per column per workgroup
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        reinterpret_cast<float4*>(red[wave][0])[lane + 64 * i] = ag[i];
        reinterpret_cast<float4*>(red[wave][1])[lane + 64 * i] = ab[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        int v4 = c >> 2, e = c & 3;
        int slot = v4 * 4 + e;  // float index inside the [MAXV*64] float4 array
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sg += red[w][0][slot];
            sb += red[w][1][slot];
        }
        if (part) {
            part[(size_t)blockIdx.x * 2 * D + c] = sg;
            part[(size_t)blockIdx.x * 2 * D + D + c] = sb;
        } else {
            atomicAdd(dgamma + c, sg);
            atomicAdd(dbeta + c, sb);
        }
    }
}

}  // namespace

extern "C" int dcv_ln_fwd(const float* x, long x_row_stride, const float* gamma, const float* beta, void* out, int out_is_f32,
                          float* mean, float* rstd, int M, int D, float eps, void* stream) {
    if (!x || !gamma || !beta || !out) return DCV_ERR_NULL;
    if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * MAXV_MAX || (x_row_stride & 3)) return DCV_ERR_SHAPE;
    int grid = (M + 3) / 4;
    if (grid > 4096) grid = 4096;
#define DCV_LN_FWD(F32, V) \
    hipLaunchKernelGGL((ln_fwd_kernel<F32, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out, mean, rstd, M, D, eps, x_row_stride)
    if (D <= 512) {
        if (out_is_f32) DCV_LN_FWD(true, 2);
        else DCV_LN_FWD(false, 2);
    } else {
        if (out_is_f32) DCV_LN_FWD(true, 4);
        else DCV_LN_FWD(false, 4);
    }
#undef DCV_LN_FWD
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static int ln_bwd_grid(int M) {
    int grid = (M + 3) / 4;
    return grid > 1024 ? 1024 : grid;
}
static int ln_bwd_launch(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
                         const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
                         float* dbeta, int M, int D, float* ws, long ws_floats, void* stream, const float* bscale = nullptr,
                         int rows_per_sample = 1) {
    if (!du || !x || !mean || !rstd || !gamma || !dx_out || !dgamma || !dbeta) return DCV_ERR_NULL;
    if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * MAXV_MAX || (x_row_stride & 3) || (dx_row_stride & 3)) return DCV_ERR_SHAPE;
    if (bscale && (rows_per_sample <= 0 || (M % rows_per_sample) != 0)) return DCV_ERR_SHAPE;
    const int grid = ln_bwd_grid(M);
    if (ws && ws_floats < (long)grid * 2 * D) return DCV_ERR_SHAPE;
#define DCV_LN_BWD(F32, V)                                                                                                          \
    hipLaunchKernelGGL((ln_bwd_kernel<F32, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, du, x, mean, rstd, gamma, dx_in, dx_out, \
                       (bf16_t*)dx_bf16, dgamma, dbeta, M, D, x_row_stride, dx_row_stride, ws, bscale, rows_per_sample)
    if (D <= 512) {
        if (du_is_f32) DCV_LN_BWD(true, 2);
        else DCV_LN_BWD(false, 2);
    } else {
        if (du_is_f32) DCV_LN_BWD(true, 4);
        else DCV_LN_BWD(false, 4);
    }
#undef DCV_LN_BWD
    DCV_LAUNCH_CHECK();
    if (ws && !det_reduce(ws, grid, 2L * D, dgamma, D, D, D, dbeta, D, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
    return DCV_OK;
}

extern "C" int dcv_ln_bwd(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
                          const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
                          float* dbeta, int M, int D, void* stream) {
    return ln_bwd_launch(du, du_is_f32, x, x_row_stride, mean, rstd, gamma, dx_in, dx_out, dx_row_stride, dx_bf16, dgamma, dbeta, M, D,
                         nullptr, 0, stream);
}
extern "C" long dcv_ln_bwd_det_ws_floats(int M, int D) { return (M <= 0 || D <= 0) ? DCV_ERR_SHAPE : (long)ln_bwd_grid(M) * 2 * D; }
extern "C" int dcv_ln_bwd_det(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
                              const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
                              float* dbeta, int M, int D, float* ws, long ws_floats, void* stream) {
    if (!ws) return DCV_ERR_NULL;
    return ln_bwd_launch(du, du_is_f32, x, x_row_stride, mean, rstd, gamma, dx_in, dx_out, dx_row_stride, dx_bf16, dgamma, dbeta, M, D,
                         ws, ws_floats, stream);
}
extern "C" int dcv_ln_bwd_scaled(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
                                 const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
                                 float* dbeta, int M, int D, const float* bf16_row_scale, int rows_per_sample, float* ws, long ws_floats,
                                 void* stream) {
    if (!bf16_row_scale || !dx_bf16) return DCV_ERR_NULL;
    return ln_bwd_launch(du, du_is_f32, x, x_row_stride, mean, rstd, gamma, dx_in, dx_out, dx_row_stride, dx_bf16, dgamma, dbeta, M, D,
                         ws, ws_floats, stream, bf16_row_scale, rows_per_sample);
}

// Parameter-arena kernels: fused AdamW over the flat fp32 arena, and the bf16 operand copies of the
// weights (straight and transposed) that the MFMA GEMMs read.
// AdamW restates timm/torch decoupled-weight-decay Adam (optimizers.py:20-21, trainer.py:1006):
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

// hyper (device, 8 floats) = {lr, beta1, beta2, eps, weight_decay, 1/bc1, 1/sqrt(bc2), grad_scale}: when non-null it
// overrides the by-value arguments, so a HIP-graph replay of the step sees the current step count / learning rate.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n4, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float inv_bc1, float inv_sqrt_bc2, float gscale,
                                                    const float* __restrict__ hyper) {
    if (hyper) {
        lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4];
        inv_bc1 = hyper[5]; inv_sqrt_bc2 = hyper[6]; gscale = hyper[7];
    }
    const long stride = (long)gridDim.x * 256;
    const float decay = 1.f - lr * wd, step = lr * inv_bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 Mv = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        float* pp = &P.x; float* gg = &G.x; float* mm = &Mv.x; float* vv = &V.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gr = gg[e] * gscale;
            float pe = pp[e] * decay;
            mm[e] = b1 * mm[e] + (1.f - b1) * gr;
            vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
            pp[e] = pe - step * mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = Mv;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    // scalar tail
    if (blockIdx.x == 0) {
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            float gr = g[i] * gscale;
            float pe = p[i] * decay;
            float me = b1 * m[i] + (1.f - b1) * gr;
            float ve = b2 * v[i] + (1.f - b2) * gr * gr;
            m[i] = me;
            v[i] = ve;
            p[i] = pe - step * me / (sqrtf(ve) * inv_sqrt_bc2 + eps);
        }
    }
}

// Stochastic rounding fp32 -> bf16 (round up with probability = the discarded fraction).  The random 16 bits are a
// hash of (element offset from the arena base, seed), so the straight and the transposed operand copy of one weight
// get the SAME rounding and a step is reproducible from its seed.  Why: with round-to-nearest the bf16 operand copy of
// a weight stays put for many AdamW steps of +-lr (4.9e-5 against a bf16 ulp of ~2e-4 at |w|=0.03) while the fp32
// master moves, and that lag is coherent over all weights; stochastic rounding makes the copy unbiased every step.
__device__ __forceinline__ unsigned sr_bits(unsigned idx, unsigned seed) {
    unsigned h = idx ^ (seed * 0x9E3779B9u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h & 0xFFFFu;  // independent draws per element AND per step (a Weyl walk over the steps measured 3x worse)
}
__device__ __forceinline__ unsigned short sr_bf16(float x, unsigned idx, unsigned seed) {
    unsigned b = __float_as_uint(x);
    if ((b & 0x7F800000u) == 0x7F800000u) return (unsigned short)(b >> 16);  // inf / nan: truncate
    return (unsigned short)((b + sr_bits(idx, seed)) >> 16);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n4, long n,
                                                        const unsigned* __restrict__ seed_dev) {
    const long stride = (long)gridDim.x * 256;
    if (seed_dev) {
        const unsigned seed = seed_dev[0];
        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(src)[i];
            const unsigned e = (unsigned)(i * 4);
            uint2 o;
            o.x = (unsigned)sr_bf16(v.x, e, seed) | ((unsigned)sr_bf16(v.y, e + 1, seed) << 16);
            o.y = (unsigned)sr_bf16(v.z, e + 2, seed) | ((unsigned)sr_bf16(v.w, e + 3, seed) << 16);
            reinterpret_cast<uint2*>(dst)[i] = o;
        }
        if (blockIdx.x == 0)
            for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) d16[i] = sr_bf16(src[i], (unsigned)i, seed);
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = pack4_bf16(v.x, v.y, v.z, v.w);
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = (bf16_t)src[i];
}

// desc[k] = {src offset (floats), dst offset (bf16 elems), R, C}: dst[C][R] = bf16(src[R][C])
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const long long* __restrict__ desc, const unsigned* __restrict__ seed_dev) {
    __shared__ float tile[64][65];
    const long long* d = desc + 4 * blockIdx.y;
    const long long so = d[0], doff = d[1];
    const int R = (int)d[2], C = (int)d[3];
    const int tr = (R + 63) / 64, tc = (C + 63) / 64;
    if ((int)blockIdx.x >= tr * tc) return;
    const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < R && c < C) ? src[so + (long long)r * C + c] : 0.f;
    }
    __syncthreads();
    if (seed_dev) {
        const unsigned seed = seed_dev[0];
        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
        for (int cc = ty; cc < 64; cc += 4) {
            int c = c0 + cc, r = r0 + tx;
            if (c < C && r < R) d16[doff + (long long)c * R + r] = sr_bf16(tile[tx][cc], (unsigned)(so + (long long)r * C + c), seed);
        }
        return;
    }
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, r = r0 + tx;
        if (c < C && r < R) dst[doff + (long long)c * R + r] = (bf16_t)tile[tx][cc];
    }
}

// Scaled re-cast of parts of the operand copy (the pre-scaled-q attention path, attn.hip): desc[k] = {src offset (floats from src_base), dst
// offset, count, scaled_count, kind}.  kind 0: dst_bf16[dst .. dst + count) = bf16(f * src) with f = scale for the first scaled_count elements and
// 1 after them (the same rounding — nearest, or stochastic with the SAME per-element bits as dcv_cast_bf16_sr — as the copy it overwrites: one
// rounding of scale * w, not a second rounding of the copy); kind 1: dst_f32[dst .. ) = f * src (the q part of a qkv bias).
__global__ __launch_bounds__(256) void cast_scaled_ranges_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst_bf16,
                                                                 float* __restrict__ dst_f32, const long long* __restrict__ desc, float scale,
                                                                 const unsigned* __restrict__ seed_dev) {
    const long long* d = desc + 5 * blockIdx.y;
    const long long so = d[0], doff = d[1], count = d[2], nsc = d[3], kind = d[4];
    const unsigned seed = seed_dev ? seed_dev[0] : 0u;
    unsigned short* d16 = reinterpret_cast<unsigned short*>(dst_bf16);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float v = src[so + i] * (i < nsc ? scale : 1.f);
        if (kind == 1) dst_f32[doff + i] = v;
        else if (seed_dev) d16[doff + i] = sr_bf16(v, (unsigned)(so + i), seed);
        else dst_bf16[doff + i] = (bf16_t)v;
    }
}

// sum of squares of a flat fp32 range, added to *acc (one atomic per block)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n4, long n, float* __restrict__ acc,
                                                    float* __restrict__ part) {
    const long stride = (long)gridDim.x * 256;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (part) part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];  // deterministic mode: summed in block order by det_reduce
        else atomicAdd(acc, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    }
}

// x *= min(1, max_norm / (sqrt(*sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_'s coefficient, read from device memory)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ x, long n4, long n, const float* __restrict__ sumsq, float max_norm) {
    const float coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    if (!(coef < 1.f)) return;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
        reinterpret_cast<float4*>(x)[i] = v;
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) x[i] *= coef;
}

__global__ __launch_bounds__(256) void fill_cls_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                       const float* __restrict__ pos0, int B, long batch_stride, int D) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * D) return;
    int b = i / D, d = i % D;
    x[(size_t)b * batch_stride + d] = cls[d] + pos0[d];
}

}  // namespace

extern "C" int dcv_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v) return DCV_ERR_NULL;
    if (n <= 0 || step <= 0) return DCV_ERR_SHAPE;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DCV_ERR_ALIGN;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    long n4 = n / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr, beta1, beta2, eps,
                       weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, (const float*)nullptr);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static __global__ void adamw_hyper_kernel(float* hyper, float lr, float b1, float b2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
                                   float gscale) {
    hyper[0] = lr; hyper[1] = b1; hyper[2] = b2; hyper[3] = eps; hyper[4] = wd; hyper[5] = inv_bc1; hyper[6] = inv_sqrt_bc2; hyper[7] = gscale;
}

extern "C" int dcv_adamw_set_hyper(float* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                   float grad_scale, void* stream) {
    if (!hyper_dev) return DCV_ERR_NULL;
    if (step <= 0) return DCV_ERR_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_hyper_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper_dev, lr, beta1, beta2, eps, weight_decay,
                       (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_adamw_dyn(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev, void* stream) {
    if (!p || !g || !m || !v || !hyper_dev) return DCV_ERR_NULL;
    if (n <= 0) return DCV_ERR_SHAPE;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DCV_ERR_ALIGN;
    long n4 = n / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, 0.f, 0.f, 0.f, 0.f, 0.f,
                       0.f, 0.f, 0.f, hyper_dev);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_cast_bf16(const float* src, void* dst, long n, void* stream) {
    if (!src || !dst) return DCV_ERR_NULL;
    if (n <= 0) return DCV_ERR_SHAPE;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return DCV_ERR_ALIGN;
    long n4 = n / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n4, n,
                       (const unsigned*)nullptr);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_cast_bf16_sr(const float* src, void* dst, long n, const unsigned* seed_dev, void* stream) {
    if (!src || !dst || !seed_dev) return DCV_ERR_NULL;
    if (n <= 0 || n > 0xFFFFFFFFL) return DCV_ERR_SHAPE;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return DCV_ERR_ALIGN;
    long n4 = n / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n4, n, seed_dev);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_cast_transpose_bf16(const float* src_base, void* dst_base, const long long* desc_dev, int n_desc, int max_tiles,
                                       void* stream) {
    if (!src_base || !dst_base || !desc_dev) return DCV_ERR_NULL;
    if (n_desc <= 0 || max_tiles <= 0) return DCV_ERR_SHAPE;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(max_tiles, n_desc), dim3(256), 0, (hipStream_t)stream, src_base, (bf16_t*)dst_base, desc_dev,
                       (const unsigned*)nullptr);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_cast_transpose_bf16_sr(const float* src_base, void* dst_base, const long long* desc_dev, int n_desc, int max_tiles,
                                          const unsigned* seed_dev, void* stream) {
    if (!src_base || !dst_base || !desc_dev || !seed_dev) return DCV_ERR_NULL;
    if (n_desc <= 0 || max_tiles <= 0) return DCV_ERR_SHAPE;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(max_tiles, n_desc), dim3(256), 0, (hipStream_t)stream, src_base, (bf16_t*)dst_base, desc_dev,
                       seed_dev);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_cast_scaled_ranges(const float* src_base, void* dst_bf16, float* dst_f32, const long long* desc_dev, int n_desc,
                                      int blocks_per_desc, float scale, const unsigned* seed_dev, void* stream) {
    if (!src_base || !desc_dev || (!dst_bf16 && !dst_f32)) return DCV_ERR_NULL;
    if (n_desc <= 0 || blocks_per_desc <= 0) return DCV_ERR_SHAPE;
    hipLaunchKernelGGL(cast_scaled_ranges_kernel, dim3(blocks_per_desc, n_desc), dim3(256), 0, (hipStream_t)stream, src_base, (bf16_t*)dst_bf16,
                       dst_f32, desc_dev, scale, seed_dev);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static long sumsq_grid(long n) {
    long grid = (n / 4 + 255) / 256;
    if (grid > 1024) grid = 1024;
    return grid < 1 ? 1 : grid;
}
static int sumsq_launch(const float* x, long n, float* acc, float* ws, long ws_floats, void* stream) {
    if (!x || !acc) return DCV_ERR_NULL;
    if (n <= 0) return DCV_ERR_SHAPE;
    if ((uintptr_t)x & 15) return DCV_ERR_ALIGN;
    const long grid = sumsq_grid(n);
    if (ws && ws_floats < grid) return DCV_ERR_SHAPE;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n / 4, n, acc, ws);
    DCV_LAUNCH_CHECK();
    if (ws && !det_reduce(ws, (int)grid, 1, acc, 1, 1, 1, nullptr, 0, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
    return DCV_OK;
}
extern "C" int dcv_sumsq_acc(const float* x, long n, float* acc, void* stream) { return sumsq_launch(x, n, acc, nullptr, 0, stream); }
extern "C" long dcv_sumsq_det_ws_floats(long n) { return n <= 0 ? DCV_ERR_SHAPE : sumsq_grid(n); }
extern "C" int dcv_sumsq_acc_det(const float* x, long n, float* acc, float* ws, long ws_floats, void* stream) {
    if (!ws) return DCV_ERR_NULL;
    return sumsq_launch(x, n, acc, ws, ws_floats, stream);
}

extern "C" int dcv_clip_scale(float* x, long n, const float* sumsq_dev, float max_norm, void* stream) {
    if (!x || !sumsq_dev) return DCV_ERR_NULL;
    if (n <= 0 || !(max_norm > 0.f)) return DCV_ERR_SHAPE;
    if ((uintptr_t)x & 15) return DCV_ERR_ALIGN;
    long n4 = n / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n4, n, sumsq_dev, max_norm);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_fill_cls(float* x, const float* cls, const float* pos0, int B, long batch_stride, int D, void* stream) {
    if (!x || !cls || !pos0) return DCV_ERR_NULL;
    if (B <= 0 || D <= 0) return DCV_ERR_SHAPE;
    hipLaunchKernelGGL(fill_cls_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, cls, pos0, B, batch_stride, D);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_version(void) { return 106; }  // 1.06: round 5 (dcv_attn_bwd_fused* retired to tools/probes; dcv_gemm_nt_resid_ln wants u_out 16-byte aligned, ldu % 8)

extern "C" const char* dcv_error_string(int code) {
    switch (code) {
        case DCV_OK: return "ok";
        case DCV_ERR_SHAPE: return "bad shape";
        case DCV_ERR_ALIGN: return "bad alignment / leading dimension";
        case DCV_ERR_UNSUPPORTED: return "unsupported configuration (e.g. head_dim != 64)";
        case DCV_ERR_LAUNCH: return "HIP launch failed";
        case DCV_ERR_NULL: return "null pointer";
        default: return "unknown error";
    }
}

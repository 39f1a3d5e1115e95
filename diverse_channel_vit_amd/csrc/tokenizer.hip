// Per-channel patch tokeniser pieces and the channel-diversity (orthogonality) regulariser.
//
//  im2col      : x f32 [B,Ct,H,W] + channel gather -> Xp bf16 [B*C*n, P*P]  (row = (b,c,i,j), k = u*P+v).
//                The patch projection itself is dcv_gemm_nt(DCV_EPI_PATCH) on Xp
//                (PatchEmbedPerChannel: channel gather models/dichavit.py:134/210, Conv3d :377).
//  patch_bwd   : one pass over d(tokens): bf16 d(conv out) for the weight-gradient GEMM, plus the
//                reductions d(channel_embed rows) [C,D], d(resampled pos table) [1+n,D], d(cls) [D]
//                (adjoints of dichavit.py:409-411, 561-565).
//  ortho_fwd/bwd: ortho_proj_loss_fn_v2 (models/loss_fn.py:24-59) through the exact O(T*D) identity
//                (SURVEY §2.3 K6): with fh_t = y_t/max(|y_t|,1e-12), s_c = sum_{t in c} fh_t,
//                    pos_sum = sum_c(|s_c|^2 - sum_{t in c}|fh_t|^2),  neg_sum = |sum_c s_c|^2 - sum_c |s_c|^2.
//                One HBM pass over Y forward, one backward; the [B,T,T] cosine matrix is never formed.
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

// IN = float (already-normalised images, the reference's batch format) or uint8_t (raw pixels: the per-channel
// (x*scale[c] + shift[c]) normalisation of the CPU data pipeline is fused here); scale/shift are
// indexed by the GATHERED channel position c (the caller gathers them like channel_embed) and may be null.
template <typename IN>
__global__ __launch_bounds__(256) void im2col_kernel(const IN* __restrict__ x, const int* __restrict__ ch_idx,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     bf16_t* __restrict__ out, int B, int Ct, int C, int H, int W, int P) {
    // one thread per 4 pixels of the gathered image: (b, c, y, x4)
    const int W4 = (W / P) * P / 4;  // only full patches
    const int Hh = (H / P) * P;
    const size_t total = (size_t)B * C * Hh * W4;
    const int wp = W / P, hp = H / P;
    const int PP = P * P;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int x4 = idx % W4;
        size_t r = idx / W4;
        int y = r % Hh;
        r /= Hh;
        int c = r % C;
        int b = r / C;
        const IN* src = x + (((size_t)b * Ct + ch_idx[c]) * H + y) * W + x4 * 4;
        float4 v;
        if constexpr (sizeof(IN) == 1) {
            const uchar4 u = *reinterpret_cast<const uchar4*>(src);
            v = make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
        } else {
            v = *reinterpret_cast<const float4*>(src);
        }
        if (scale) {
            const float sc = scale[c], sh = shift ? shift[c] : 0.f;
            v = make_float4(v.x * sc + sh, v.y * sc + sh, v.z * sc + sh, v.w * sc + sh);
        }
        int pj = (x4 * 4) / P, vv = (x4 * 4) % P;
        int pi = y / P, u = y % P;
        size_t row = ((size_t)b * C + c) * (hp * wp) + pi * wp + pj;
        *reinterpret_cast<uint2*>(out + row * PP + u * P + vv) = pack4_bf16(v.x, v.y, v.z, v.w);
    }
}

// ---------------------------------------------------------------------------------------------
constexpr int PB_IT = 2;  // positions per workgroup

__global__ __launch_bounds__(256) void patch_bwd_kernel(const float* __restrict__ dx0, const float* __restrict__ dYloss,
                                                        bf16_t* __restrict__ dYb, float* __restrict__ dE,
                                                        float* __restrict__ dpos, float* __restrict__ dcls, int B, int C,
                                                        int n, int D, float* __restrict__ partE, float* __restrict__ partP) {
    // partE / partP != nullptr (deterministic mode): the per-workgroup sums for d(channel_embed) and d(pos[1:]) are stored —
    // partE[(position block * bpar + bsub)][C * D], partP[(channel * bpar + bsub)][n * D] — and added up in index order by det_reduce
    const int nv = D >> 2;
    const int T = C * n;
    const int c = blockIdx.y;
    if (c == C) {  // CLS rows: d(cls) = d(pos[0]) = sum_b dx0[b,0,:]
        if (blockIdx.x != 0) return;
        for (int v = threadIdx.x; v < nv; v += 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = 0; b < B; ++b) {
                float4 g = reinterpret_cast<const float4*>(dx0 + (size_t)b * (T + 1) * D)[v];
                s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            }
            float* pc = dcls + v * 4;
            float* pp = dpos + v * 4;
            atomicAdd(pc, s.x); atomicAdd(pc + 1, s.y); atomicAdd(pc + 2, s.z); atomicAdd(pc + 3, s.w);
            atomicAdd(pp, s.x); atomicAdd(pp + 1, s.y); atomicAdd(pp + 2, s.z); atomicAdd(pp + 3, s.w);
        }
        return;
    }
    const int i0 = blockIdx.x * PB_IT;
    // threads: v = float4 column, bsub = batch sub-lane
    const int lanes_per_row = nv;                       // <= 256
    const int bpar = 256 / lanes_per_row;               // batches processed in parallel
    const int v = threadIdx.x % lanes_per_row, bsub = threadIdx.x / lanes_per_row;
    float4 aE = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 aP[PB_IT];
#pragma unroll
    for (int k = 0; k < PB_IT; ++k) aP[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bsub < bpar) {
        for (int b = bsub; b < B; b += bpar) {
#pragma unroll
            for (int k = 0; k < PB_IT; ++k) {
                int i = i0 + k;
                if (i < n) {
                    size_t t = (size_t)c * n + i;
                    float4 g = reinterpret_cast<const float4*>(dx0 + ((size_t)b * (T + 1) + 1 + t) * D)[v];
                    aE.x += g.x; aE.y += g.y; aE.z += g.z; aE.w += g.w;
                    aP[k].x += g.x; aP[k].y += g.y; aP[k].z += g.z; aP[k].w += g.w;
                    if (dYloss) {
                        float4 l = reinterpret_cast<const float4*>(dYloss + ((size_t)b * T + t) * D)[v];
                        g.x += l.x; g.y += l.y; g.z += l.z; g.w += l.w;
                    }
                    reinterpret_cast<uint2*>(dYb + ((size_t)b * T + t) * D)[v] = pack4_bf16(g.x, g.y, g.z, g.w);
                }
            }
        }
        if (partE) {
            reinterpret_cast<float4*>(partE + ((size_t)blockIdx.x * bpar + bsub) * C * D + (size_t)c * D)[v] = aE;
#pragma unroll
            for (int k = 0; k < PB_IT; ++k) {
                int i = i0 + k;
                if (i < n) reinterpret_cast<float4*>(partP + ((size_t)c * bpar + bsub) * n * D + (size_t)i * D)[v] = aP[k];
            }
        } else {
            float* pe = dE + (size_t)c * D + v * 4;
            atomicAdd(pe, aE.x); atomicAdd(pe + 1, aE.y); atomicAdd(pe + 2, aE.z); atomicAdd(pe + 3, aE.w);
#pragma unroll
            for (int k = 0; k < PB_IT; ++k) {
                int i = i0 + k;
                if (i < n) {
                    float* pp = dpos + (size_t)(1 + i) * D + v * 4;
                    atomicAdd(pp, aP[k].x); atomicAdd(pp + 1, aP[k].y); atomicAdd(pp + 2, aP[k].z); atomicAdd(pp + 3, aP[k].w);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
constexpr int OV = 4;  // float4 per lane -> D <= 1024
constexpr int ORTHO_CHUNK = 28;  // tokens per workgroup (4 waves x 7)

__global__ __launch_bounds__(256) void ortho_fwd_partial(const float* __restrict__ Y, float* __restrict__ S,
                                                         float* __restrict__ selfsq, float* __restrict__ inv_norm, int C,
                                                         int n, int D, float* __restrict__ part) {
    // part != nullptr (deterministic mode): token chunk k stores its sums to part[k][B*C*D + B*C] (S columns, then selfsq)
    __shared__ float red[4][OV * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bc = blockIdx.y;  // b*C + c
    const int nv = D >> 2;
    const int t0 = blockIdx.x * ORTHO_CHUNK;
    float4 acc[OV];
#pragma unroll
    for (int i = 0; i < OV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 0.f;
    for (int t = t0 + wave; t < min(n, t0 + ORTHO_CHUNK); t += 4) {
        const size_t row = (size_t)bc * n + t;
        const float4* yr = reinterpret_cast<const float4*>(Y + row * D);
        float4 v[OV];
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < OV; ++i) {
            int cidx = lane + 64 * i;
            v[i] = (cidx < nv) ? yr[cidx] : make_float4(0.f, 0.f, 0.f, 0.f);
            q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        q = wave_sum(q);
        const float inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
        if (lane == 0) inv_norm[row] = inv;
        ss += q * inv * inv;
#pragma unroll
        for (int i = 0; i < OV; ++i) {
            acc[i].x += v[i].x * inv; acc[i].y += v[i].y * inv; acc[i].z += v[i].z * inv; acc[i].w += v[i].w * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < OV; ++i) reinterpret_cast<float4*>(red[wave])[lane + 64 * i] = acc[i];
    __shared__ float red_ss[4];
    if (lane == 0) red_ss[wave] = ss;
    __syncthreads();
    if (part) {
        const size_t BC = gridDim.y;
        float* pk = part + (size_t)blockIdx.x * (BC * D + BC);
        for (int c = threadIdx.x; c < D; c += 256) pk[(size_t)bc * D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
        if (threadIdx.x == 0) pk[BC * D + bc] = red_ss[0] + red_ss[1] + red_ss[2] + red_ss[3];
    } else {
        for (int c = threadIdx.x; c < D; c += 256) atomicAdd(S + (size_t)bc * D + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
        if (threadIdx.x == 0) atomicAdd(selfsq + bc, red_ss[0] + red_ss[1] + red_ss[2] + red_ss[3]);
    }
}

// zero the two atomic accumulators (a kernel, not hipMemsetAsync: memset nodes did not replay correctly when the
// step is captured into a HIP graph — the statistics went wrong from the second replay on)
__global__ __launch_bounds__(256) void ortho_zero_kernel(float* __restrict__ S, long nS, float* __restrict__ selfsq, long nQ) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nS; i += stride) S[i] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nQ; i += stride) selfsq[i] = 0.f;
}

// per image: tot = sum_c s_c ; stats = (pos_sum, neg_sum)
__global__ __launch_bounds__(256) void ortho_fwd_final(const float* __restrict__ S, const float* __restrict__ selfsq,
                                                       float* __restrict__ tot, float* __restrict__ stats, int C, int D) {
    __shared__ float red[3][4];
    const int b = blockIdx.x;
    float ssq = 0.f, tsq = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float t = 0.f;
        for (int c = 0; c < C; ++c) {
            float s = S[((size_t)b * C + c) * D + d];
            ssq += s * s;
            t += s;
        }
        tot[(size_t)b * D + d] = t;
        tsq += t * t;
    }
    float self = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) self += selfsq[(size_t)b * C + c];
    ssq = wave_sum(ssq);
    tsq = wave_sum(tsq);
    self = wave_sum(self);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = ssq;
        red[1][wave] = tsq;
        red[2][wave] = self;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        float t2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        float sf = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        stats[2 * b] = a - sf;                 // pos_sum
        stats[2 * b + 1] = (C == 1) ? 0.f : (t2 - a);  // neg_sum (exactly 0 when no different-channel pair exists)
    }
}

// dY_t = normalize'(g_t),  g_t = 2a (s_c - fh_t) + 2b (tot - s_c),  (a,b) = dL/d(pos_sum_b, neg_sum_b)
__global__ __launch_bounds__(256) void ortho_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ S,
                                                        const float* __restrict__ tot, const float* __restrict__ inv_norm,
                                                        const float* __restrict__ coef, float* __restrict__ dY, int C, int n,
                                                        int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bc = blockIdx.y, b = bc / C;
    const int nv = D >> 2;
    const int t0 = blockIdx.x * ORTHO_CHUNK;
    const float ca = 2.f * coef[2 * b], cb = (C == 1) ? 0.f : 2.f * coef[2 * b + 1];
    float4 base[OV];  // ca*s_c + cb*(tot - s_c)
#pragma unroll
    for (int i = 0; i < OV; ++i) {
        int cidx = lane + 64 * i;
        base[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cidx < nv) {
            float4 s = reinterpret_cast<const float4*>(S + (size_t)bc * D)[cidx];
            float4 tt = reinterpret_cast<const float4*>(tot + (size_t)b * D)[cidx];
            base[i] = make_float4(ca * s.x + cb * (tt.x - s.x), ca * s.y + cb * (tt.y - s.y), ca * s.z + cb * (tt.z - s.z),
                                  ca * s.w + cb * (tt.w - s.w));
        }
    }
    for (int t = t0 + wave; t < min(n, t0 + ORTHO_CHUNK); t += 4) {
        const size_t row = (size_t)bc * n + t;
        const float4* yr = reinterpret_cast<const float4*>(Y + row * D);
        const float inv = inv_norm[row];
        const bool clamped = inv >= 1e12f;  // |y| < eps: normalize divides by the constant eps
        float4 fh[OV], g[OV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < OV; ++i) {
            int cidx = lane + 64 * i;
            fh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cidx < nv) {
                float4 y = yr[cidx];
                fh[i] = make_float4(y.x * inv, y.y * inv, y.z * inv, y.w * inv);
                g[i] = make_float4(base[i].x - ca * fh[i].x, base[i].y - ca * fh[i].y, base[i].z - ca * fh[i].z,
                                   base[i].w - ca * fh[i].w);
                dot += g[i].x * fh[i].x + g[i].y * fh[i].y + g[i].z * fh[i].z + g[i].w * fh[i].w;
            }
        }
        dot = clamped ? 0.f : wave_sum(dot);
#pragma unroll
        for (int i = 0; i < OV; ++i) {
            int cidx = lane + 64 * i;
            if (cidx < nv)
                reinterpret_cast<float4*>(dY + row * D)[cidx] =
                    make_float4((g[i].x - fh[i].x * dot) * inv, (g[i].y - fh[i].y * dot) * inv, (g[i].z - fh[i].z * dot) * inv,
                                (g[i].w - fh[i].w * dot) * inv);
        }
    }
}

// token gather / scatter for dropout_tokens_hcs (dichavit.py:568-627): out[b,k,:] = x[b,idx[k],:] and its adjoint
__global__ __launch_bounds__(256) void gather_tokens_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                            float* __restrict__ out, int B, int N, int Nk, int D4, int scatter) {
    const size_t total = (size_t)B * Nk * D4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int d = i % D4;
        const size_t r = i / D4;
        const int k = r % Nk, b = r / Nk;
        const size_t full = ((size_t)b * N + idx[k]) * D4 + d, kept = ((size_t)b * Nk + k) * D4 + d;
        if (scatter) reinterpret_cast<float4*>(out)[full] = reinterpret_cast<const float4*>(x)[kept];
        else reinterpret_cast<float4*>(out)[kept] = reinterpret_cast<const float4*>(x)[full];
    }
}

}  // namespace

extern "C" int dcv_gather_tokens(const float* x, const int* idx, float* out, int B, int N, int Nk, int D, int scatter, void* stream) {
    if (!x || !idx || !out) return DCV_ERR_NULL;
    if (B <= 0 || N <= 0 || Nk <= 0 || Nk > N || D <= 0 || (D & 3)) return DCV_ERR_SHAPE;
    size_t total = (size_t)B * Nk * (D / 4);
    size_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(gather_tokens_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, idx, out, B, N, Nk, D / 4, scatter);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_im2col_bf16(const void* x, int x_is_u8, const int* ch_idx, const float* scale, const float* shift, void* out, int B,
                               int Ct, int C, int H, int W, int P, void* stream) {
    if (!x || !ch_idx || !out) return DCV_ERR_NULL;
    if (B <= 0 || C <= 0 || Ct <= 0 || P <= 0 || (P & 3) || (W & 3) || H < P || W < P) return DCV_ERR_SHAPE;
    size_t total = (size_t)B * C * ((H / P) * P) * ((W / P) * P / 4);
    size_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (x_is_u8)
        hipLaunchKernelGGL(im2col_kernel<unsigned char>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x,
                           ch_idx, scale, shift, (bf16_t*)out, B, Ct, C, H, W, P);
    else
        hipLaunchKernelGGL(im2col_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ch_idx, scale,
                           shift, (bf16_t*)out, B, Ct, C, H, W, P);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static int patch_bwd_launch(const float* dx0, const float* dYloss, void* dY_bf16, float* dE, float* dpos, float* dcls, int B, int C, int n,
                            int D, float* ws, long ws_floats, void* stream) {
    if (!dx0 || !dY_bf16 || !dE || !dpos || !dcls) return DCV_ERR_NULL;
    if (B <= 0 || C <= 0 || n <= 0 || D <= 0 || (D & 3) || D > 1024) return DCV_ERR_SHAPE;
    dim3 grid((n + PB_IT - 1) / PB_IT, C + 1);
    const int bpar = 256 / (D >> 2);
    const long nE = (long)grid.x * bpar * C * D, nP = (long)C * bpar * n * D;
    if (ws && ws_floats < nE + nP) return DCV_ERR_SHAPE;
    // a position block whose last position does not exist leaves its partP slot unwritten only for i >= n, which is outside [0, n)
    hipLaunchKernelGGL(patch_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dx0, dYloss, (bf16_t*)dY_bf16, dE, dpos, dcls, B, C, n, D,
                       ws, ws ? ws + nE : nullptr);
    DCV_LAUNCH_CHECK();
    if (ws) {
        // rows with bsub >= B never run the batch loop but still store their (zero) sums: every slot of both part arrays is written
        if (!det_reduce(ws, (int)grid.x * bpar, (long)C * D, dE, (long)C * D, D, D, nullptr, 0, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
        if (!det_reduce(ws + nE, C * bpar, (long)n * D, dpos + D, (long)n * D, D, D, nullptr, 0, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
    }
    return DCV_OK;
}
extern "C" int dcv_patch_bwd(const float* dx0, const float* dYloss, void* dY_bf16, float* dE, float* dpos, float* dcls, int B, int C,
                             int n, int D, void* stream) {
    return patch_bwd_launch(dx0, dYloss, dY_bf16, dE, dpos, dcls, B, C, n, D, nullptr, 0, stream);
}
extern "C" long dcv_patch_bwd_det_ws_floats(int B, int C, int n, int D) {
    if (B <= 0 || C <= 0 || n <= 0 || D <= 0 || (D & 3) || D > 1024) return DCV_ERR_SHAPE;
    const long bpar = 256 / (D >> 2);
    return (long)((n + PB_IT - 1) / PB_IT) * bpar * C * D + (long)C * bpar * n * D;
}
extern "C" int dcv_patch_bwd_det(const float* dx0, const float* dYloss, void* dY_bf16, float* dE, float* dpos, float* dcls, int B, int C,
                                 int n, int D, float* ws, long ws_floats, void* stream) {
    if (!ws) return DCV_ERR_NULL;
    return patch_bwd_launch(dx0, dYloss, dY_bf16, dE, dpos, dcls, B, C, n, D, ws, ws_floats, stream);
}

static int ortho_fwd_launch(const float* Y, float* S, float* selfsq, float* tot, float* inv_norm, float* stats, int B, int C, int n, int D,
                            float* ws, long ws_floats, void* stream) {
    if (!Y || !S || !selfsq || !tot || !inv_norm || !stats) return DCV_ERR_NULL;
    if (B <= 0 || C <= 0 || n <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * OV) return DCV_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((n + ORTHO_CHUNK - 1) / ORTHO_CHUNK, B * C);
    const long nS = (long)B * C * D, nQ = (long)B * C;
    if (ws && ws_floats < (long)grid.x * (nS + nQ)) return DCV_ERR_SHAPE;
    {
        long zg = (nS + 255) / 256;
        if (zg > 1024) zg = 1024;
        hipLaunchKernelGGL(ortho_zero_kernel, dim3((unsigned)zg), dim3(256), 0, s, S, nS, selfsq, nQ);
    }
    hipLaunchKernelGGL(ortho_fwd_partial, grid, dim3(256), 0, s, Y, S, selfsq, inv_norm, C, n, D, ws);
    if (ws && !det_reduce(ws, (int)grid.x, nS + nQ, S, nS, D, D, selfsq, nQ, s)) return DCV_ERR_LAUNCH;
    hipLaunchKernelGGL(ortho_fwd_final, dim3(B), dim3(256), 0, s, S, selfsq, tot, stats, C, D);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}
extern "C" int dcv_ortho_fwd(const float* Y, float* S, float* selfsq, float* tot, float* inv_norm, float* stats, int B, int C, int n,
                             int D, void* stream) {
    return ortho_fwd_launch(Y, S, selfsq, tot, inv_norm, stats, B, C, n, D, nullptr, 0, stream);
}
extern "C" long dcv_ortho_fwd_det_ws_floats(int B, int C, int n, int D) {
    if (B <= 0 || C <= 0 || n <= 0 || D <= 0) return DCV_ERR_SHAPE;
    return (long)((n + ORTHO_CHUNK - 1) / ORTHO_CHUNK) * ((long)B * C * D + (long)B * C);
}
extern "C" int dcv_ortho_fwd_det(const float* Y, float* S, float* selfsq, float* tot, float* inv_norm, float* stats, int B, int C, int n,
                                 int D, float* ws, long ws_floats, void* stream) {
    if (!ws) return DCV_ERR_NULL;
    return ortho_fwd_launch(Y, S, selfsq, tot, inv_norm, stats, B, C, n, D, ws, ws_floats, stream);
}

extern "C" int dcv_ortho_bwd(const float* Y, const float* S, const float* tot, const float* inv_norm, const float* coef, float* dY,
                             int B, int C, int n, int D, void* stream) {
    if (!Y || !S || !tot || !inv_norm || !coef || !dY) return DCV_ERR_NULL;
    if (B <= 0 || C <= 0 || n <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * OV) return DCV_ERR_SHAPE;
    dim3 grid((n + ORTHO_CHUNK - 1) / ORTHO_CHUNK, B * C);
    hipLaunchKernelGGL(ortho_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, Y, S, tot, inv_norm, coef, dY, C, n, D);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// bf16 MFMA GEMMs for the DiChaViT encoder (gfx950).
//
//  gemm_nt : C[M,N] = A[M,K] . W[N,K]^T   (both operands K-contiguous)  + fused epilogues.
//            Replaces nn.Linear forward (models/vit.py:116,119,72-74) and, with the transposed bf16
//            weight copy, its input gradient; also the Conv3d(1,D,(1,P,P)) patch projection on the
//            im2col'd image (models/dichavit.py:77-82,377) with the +bias +channel_embed +pos epilogue
//            (dichavit.py:409-411, 565).
//  gemm_tn : dW[P,Q] += sum_m Y[m,P] . X[m,Q]  (+ dbias[P] += sum_m Y[m,P]); the weight gradient
//            of the same layers.  Reduction dim is the long one (M = B*N tokens): split over
//            workgroups, fp32 atomics into the gradient arena.
//
// gemm_nt: persistent 256x128 (8 waves, 3-stage LDS-DMA ring) and 256x384 (two 80 KB stages) kernels, geometry below.
// gemm_tn: 128x128 register-staged kernel (4 waves) and the 384x128 LDS-DMA ring kernel (8 waves).
#include <atomic>
#include <type_traits>
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

constexpr int BK = 64;  // gemm_tn_kernel reduction rows per stage (the NT kernels stream NT_BK = N3_BK = 64-wide stages, gemm_tn384 T3_BK = 32 rows)

struct GemmNtArgs {
    const bf16_t* A;
    int lda;
    const bf16_t* W;
    int ldw;
    int M, N, K;
    const float* bias;
    void* out;
    int ldo;
    void* out2;
    int ldo2;
    const void* aux;
    int ldaux;
    const float* aux2;
    int T, n;
    // residual + LayerNorm epilogue (gemm_nt384_kernel<DCV_EPI_RESID_LN>, N == 384): gamma / beta [N], mean / rstd [M] out, out2 = bf16 LN output
    const float* ln_g;
    const float* ln_b;
    float* ln_mean;
    float* ln_rstd;
    float ln_eps;
};
constexpr int DCV_EPI_RESID_LN = 6;  // kernel-internal: reached through dcv_gemm_nt_resid_ln only

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7: far below the bf16 rounding of the outputs), sharing
// e = exp(-z^2/2) with the Gaussian pdf of GELU':  GELU(z) = z * cdf,  GELU'(z) = cdf + z * e / sqrt(2 pi).
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& e) {
    const float x = fabsf(z) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
    e = __builtin_amdgcn_exp2f(-0.72134752044448170f * z * z);  // exp(-z^2/2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f * (1.0f + copysignf(erf_abs, z));
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// the same arithmetic on two elements at a time (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32; rcp and exp2 stay scalar)
__device__ __forceinline__ void gelu_parts2(f32x2_t z, f32x2_t& g, f32x2_t& gp) {
    f32x2_t ax = {fabsf(z.x), fabsf(z.y)};
    const f32x2_t d = ax * 0.23164189f + 1.0f;  // 1 + 0.3275911 |z| / sqrt 2
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2_t zz = z * z * -0.72134752044448170f;
    const f32x2_t e = {__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)};
    f32x2_t poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const f32x2_t h = poly * e * -0.5f + 0.5f;  // 0.5 erfc-part: cdf(|z|) - 0.5 = 0.5 - 0.5 poly e
    // cdf(z) = 0.5 + sign(z) * h
    const f32x2_t sh = {copysignf(h.x, z.x), copysignf(h.y, z.y)};
    const f32x2_t cdf = sh + 0.5f;
    g = z * cdf;
    gp = z * e * 0.39894228040143268f + cdf;
}
__device__ __forceinline__ uint4 pack8_bf16(const float* v) {
    uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ void unpack8_bf16(uint4 u, float* v) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
// Streaming (non-temporal) 16-byte accesses for data this kernel touches exactly once — the output tile, the residual
// and the saved pre-activation: they must not evict the A row-panel, which the other column tiles of the same XCD are
// about to re-read, from the 4 MB L2 (PMC before: fc1 fetched its 77 MB A operand 3.3 times).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st128_stream(void* p, uint4 v) {
    u32x4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ uint4 ld128_stream(const void* p) {
    u32x4_t x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void load8_f32_stream(const float* p, float* v) {
    uint4 a = ld128_stream(p), b = ld128_stream(p + 4);
    v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
    v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y); v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
}
__device__ __forceinline__ void store8_f32_stream(float* p, const float* v) {
    st128_stream(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    st128_stream(p + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
}
__device__ __forceinline__ void load8_f32(const float* p, float* v) {
    float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8_f32(float* p, const float* v) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// Epilogue on 8 consecutive columns n..n+7 of row m (16-byte global accesses; N % 8 == 0), split in two so that the
// auxiliary global loads of all passes are in flight together before any of them is consumed:
//   epi_aux8   : loads what the epilogue needs from global memory (residual / saved pre-activation / embedding rows)
//                — (m, n) are clamped by the caller, so it needs no predicate;
//   epi_store8 : adds the bias slice (staged in LDS), applies the fused op and stores (caller predicates).
template <int EPI>
__device__ __forceinline__ void epi_aux8(const GemmNtArgs& a, int m, int n, float* x) {
    if constexpr (EPI == DCV_EPI_BIAS_RESID_F32) {
        // residual is read from aux when given (out-of-place keeps the layer input alive for the LayerNorm backward
        // at no extra traffic), else the output is updated in place
        const float* rsd = a.aux ? (const float*)a.aux + (size_t)m * a.ldaux + n : (const float*)a.out + (size_t)m * a.ldo + n;
        load8_f32(rsd, x);
    } else if constexpr (EPI == DCV_EPI_GELU_BWD_BF16) {
        unpack8_bf16(ld128_stream((const bf16_t*)a.aux + (size_t)m * a.ldaux + n), x);
    } else if constexpr (EPI == DCV_EPI_PATCH) {
        // row m = b*T + t ; token t = c*n + i  ->  channel_embed[c] + pos[1+i]
        const int b = m / a.T, t = m - b * a.T;
        const int c = t / a.n, i = t - c * a.n;
        float p8[8];
        load8_f32((const float*)a.aux + (size_t)c * a.ldaux + n, x);
        load8_f32(a.aux2 + (size_t)(1 + i) * a.ldaux + n, p8);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += p8[e];
    }
}

template <int EPI>
__device__ __forceinline__ void epi_store8(const GemmNtArgs& a, int m, int n, float* v, const float* x, const float* bz) {
    if constexpr (EPI == DCV_EPI_BIAS_BF16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bz[e];
        st128_stream((bf16_t*)a.out + (size_t)m * a.ldo + n, pack8_bf16(v));
    } else if constexpr (EPI == DCV_EPI_BIAS_GELU_BF16) {
        // out = GELU'(z), out2 = GELU(z), z = acc + bias in fp32: the backward then only multiplies (the first version saved z
        // and recomputed the erf / exp in the backward epilogue, ~45 % of that kernel's time); cdf and exp(-z^2/2) are shared
        float gp[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_t z = {v[e] + bz[e], v[e + 1] + bz[e + 1]};
            f32x2_t g2, gp2;
            gelu_parts2(z, g2, gp2);
            v[e] = g2.x; v[e + 1] = g2.y;
            gp[e] = gp2.x; gp[e + 1] = gp2.y;
        }
        st128_stream((bf16_t*)a.out + (size_t)m * a.ldo + n, pack8_bf16(gp));
        st128_stream((bf16_t*)a.out2 + (size_t)m * a.ldo2 + n, pack8_bf16(v));
    } else if constexpr (EPI == DCV_EPI_BIAS_RESID_F32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bz[e] + x[e];
        store8_f32((float*)a.out + (size_t)m * a.ldo + n, v);
    } else if constexpr (EPI == DCV_EPI_PLAIN_BF16) {
        st128_stream((bf16_t*)a.out + (size_t)m * a.ldo + n, pack8_bf16(v));
    } else if constexpr (EPI == DCV_EPI_GELU_BWD_BF16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= x[e];  // x = GELU'(z) as the forward epilogue saved it
        st128_stream((bf16_t*)a.out + (size_t)m * a.ldo + n, pack8_bf16(v));
    } else if constexpr (EPI == DCV_EPI_PATCH) {
        const int b = m / a.T, t = m - b * a.T;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bz[e];
        if (a.out2) store8_f32((float*)a.out2 + (size_t)m * a.ldo2 + n, v);  // pre-embedding tokens (ortho loss input)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += x[e];
        store8_f32((float*)a.out + ((size_t)b * (a.T + 1) + 1 + t) * a.ldo + n, v);
    }
}

// ---- register epilogue (round 3) -------------------------------------------------------------------------------------------
// The MFMAs are issued with the operands SWAPPED (W fragment as the A operand, activation fragment as B), so a wave's 16 x 16
// accumulator tile (i, j) holds, in lane (r16 = lane & 15, kg = lane >> 4), output row m = 16 i + r16 and the FOUR CONSECUTIVE
// columns n = 16 j + 4 kg .. +3 — row-contiguous data in every lane, which two register shuffles turn into full cache lines:
//   1. bf16 outputs only: four v_permlane16_swap per tile pair (j, j + 1) leave lane (r16, kg) with the EIGHT consecutive columns
//      16 (j + (kg & 1)) + 8 (kg >> 1) .. +7 of its row (the swap exchanges the odd 16-lane rows of its first operand with the even
//      rows of its second): 16 bytes per lane, but still only 4 lanes = 64 bytes per output row and instruction;
//   2. a DPP exchange between lanes l and l ^ 8 (row_ror:8, two v_mov_dpp per register) between two such column halves: afterwards
//      one store instruction covers rows rr = r16 & 7 (+ 8 for the second instruction) with EIGHT lanes per row — 8 rows x 128
//      contiguous bytes, whole cache lines, exactly what the LDS slab of rounds 1-2 produced.  (Measured without step 2: 16 half-lines
//      per instruction instead of 8 lines made every epilogue 5-25 % SLOWER than the slab — the store path is bound by the number of
//      lines an instruction touches, not by bytes; gpurun r3b.)
// Final layout of a lane inside a column group: bf16: 64-column group, 8 columns at 32 (r16 >> 3) + 16 (kg & 1) + 8 (kg >> 1);
// fp32: 32-column group (a tile pair), 4 columns at 16 (r16 >> 3) + 4 kg; rows rr and rr + 8 of the 16-row block.
// No LDS: the whole ring is free for the next tile's prefetch and no workgroup barrier separates the k-loop from the epilogue.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap_pair8(const f32x4& t0, const f32x4& t1, float* v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u32x2_t u = __builtin_amdgcn_permlane16_swap(__float_as_uint(t0[r]), __float_as_uint(t1[r]), false, false);
        v[r] = __uint_as_float(u[0]);
        v[4 + r] = __uint_as_float(u[1]);
    }
}
// x (lower column half) / y (upper column half), both held for row r16: afterwards a = (row rr: x of lanes r16 < 8 | y of lane l - 8),
// b = (row rr + 8: x of lane l + 8 | y of lanes r16 >= 8)
__device__ __forceinline__ void xchg_rows8(float x, float y, float& a, float& b) {
    const int xi = __float_as_int(x), yi = __float_as_int(y);
    a = __int_as_float(__builtin_amdgcn_update_dpp(xi, yi, 0x128, 0xF, 0xC, false));  // lanes 8-15 of every row: y[l ^ 8]
    b = __int_as_float(__builtin_amdgcn_update_dpp(yi, xi, 0x128, 0xF, 0x3, false));  // lanes 0-7: x[l ^ 8]
}
// fp32-output epilogues on 4 consecutive columns n .. n+3 of row m
template <int EPI>
__device__ __forceinline__ void epi_aux4(const GemmNtArgs& a, int m, int n, float* x) {
    if constexpr (EPI == DCV_EPI_BIAS_RESID_F32) {
        const float* rsd = a.aux ? (const float*)a.aux + (size_t)m * a.ldaux + n : (const float*)a.out + (size_t)m * a.ldo + n;
        const float4 v = *reinterpret_cast<const float4*>(rsd);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if constexpr (EPI == DCV_EPI_PATCH) {
        const int b = m / a.T, t = m - b * a.T;
        const int c = t / a.n, i = t - c * a.n;
        const float4 e = *reinterpret_cast<const float4*>((const float*)a.aux + (size_t)c * a.ldaux + n);
        const float4 p = *reinterpret_cast<const float4*>(a.aux2 + (size_t)(1 + i) * a.ldaux + n);
        x[0] = e.x + p.x; x[1] = e.y + p.y; x[2] = e.z + p.z; x[3] = e.w + p.w;
    }
}
template <int EPI>
__device__ __forceinline__ void epi_store4(const GemmNtArgs& a, int m, int n, const f32x4& acc, const float* x, const float4& bz) {
    float4 v = make_float4(acc[0] + bz.x, acc[1] + bz.y, acc[2] + bz.z, acc[3] + bz.w);
    if constexpr (EPI == DCV_EPI_BIAS_RESID_F32) {
        if (a.aux2) {  // stochastic depth (vit.py:37-56, 397-398): the branch of sample b = m / T is multiplied by aux2[b] = keep_b / keep_prob
            const float sc = a.aux2[m / a.T];
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        }
        v.x += x[0]; v.y += x[1]; v.z += x[2]; v.w += x[3];
        *reinterpret_cast<float4*>((float*)a.out + (size_t)m * a.ldo + n) = v;
    } else if constexpr (EPI == DCV_EPI_PATCH) {
        const int b = m / a.T, t = m - b * a.T;
        if (a.out2) *reinterpret_cast<float4*>((float*)a.out2 + (size_t)m * a.ldo2 + n) = v;  // pre-embedding tokens (ortho loss input)
        v.x += x[0]; v.y += x[1]; v.z += x[2]; v.w += x[3];
        *reinterpret_cast<float4*>((float*)a.out + ((size_t)b * (a.T + 1) + 1 + t) * a.ldo + n) = v;
    }
}
template <int EPI>
__device__ constexpr bool epi_f32out() { return EPI == DCV_EPI_BIAS_RESID_F32 || EPI == DCV_EPI_PATCH; }

// Epilogue of a wave's 64-row x (16 NJ)-column block of accumulators acc[4][J0 .. J0 + NJ) (NJ a multiple of 4), rows m_w + 16 i + ..,
// columns n_w + ..; `between()` runs after the first auxiliary loads have been issued and before anything is stored (the persistent
// kernels put the next tile's prefetch there).  NI: row blocks (of 16 rows) whose auxiliary loads are issued together (4 = all of
// them before `between`; fewer = fewer registers).  bz: the lane's bias values in its final layout (nt_load_bias).
template <int EPI, int NI, int NJ, int NJT, int J0, class Between>
__device__ __forceinline__ void nt_epilogue_block(const GemmNtArgs& a, const f32x4 (&acc)[4][NJT], int m_w, int n_w, int r16, int kg,
                                                  const float* bz, Between&& between) {
    constexpr bool HAS_AUX = (EPI == DCV_EPI_BIAS_RESID_F32) || (EPI == DCV_EPI_GELU_BWD_BF16) || (EPI == DCV_EPI_PATCH);
    const int rr = r16 & 7, hi = r16 >> 3;
    if constexpr (epi_f32out<EPI>()) {
        constexpr int NG = NJ / 2;  // 32-column groups
        const int cl = 16 * hi + 4 * kg;
#pragma unroll
        for (int ib = 0; ib < 4; ib += NI) {
            float x[NI][NG][2][4];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int c = 0; c < NG; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        epi_aux4<EPI>(a, min(m_w + 16 * (ib + i) + 8 * h + rr, a.M - 1), min(n_w + 32 * c + cl, a.N - 4), x[i][c][h]);
            if (ib == 0) between();
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    f32x4 va, vb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float fa, fb;
                        xchg_rows8(acc[ib + i][J0 + 2 * c][r], acc[ib + i][J0 + 2 * c + 1][r], fa, fb);
                        va[r] = fa; vb[r] = fb;
                    }
                    const int m = m_w + 16 * (ib + i) + rr, n = n_w + 32 * c + cl;
                    const float4 b4 = make_float4(bz[4 * c], bz[4 * c + 1], bz[4 * c + 2], bz[4 * c + 3]);
                    if (m < a.M && n < a.N) epi_store4<EPI>(a, m, n, va, x[i][c][0], b4);
                    if (m + 8 < a.M && n < a.N) epi_store4<EPI>(a, m + 8, n, vb, x[i][c][1], b4);
                }
        }
    } else {
        constexpr int NG = NJ / 4;  // 64-column groups
        const int cl = 32 * hi + 16 * (kg & 1) + 8 * (kg >> 1);
#pragma unroll
        for (int ib = 0; ib < 4; ib += NI) {
            float x[HAS_AUX ? NI : 1][HAS_AUX ? NG : 1][2][8];
            if constexpr (HAS_AUX) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int c = 0; c < NG; ++c)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            epi_aux8<EPI>(a, min(m_w + 16 * (ib + i) + 8 * h + rr, a.M - 1), min(n_w + 64 * c + cl, a.N - 8), x[i][c][h]);
            }
            if (ib == 0) between();
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    float v0[8], v1[8], va[8], vb[8];
                    swap_pair8(acc[ib + i][J0 + 4 * c], acc[ib + i][J0 + 4 * c + 1], v0);
                    swap_pair8(acc[ib + i][J0 + 4 * c + 2], acc[ib + i][J0 + 4 * c + 3], v1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xchg_rows8(v0[e], v1[e], va[e], vb[e]);
                    const int m = m_w + 16 * (ib + i) + rr, n = n_w + 64 * c + cl;
                    if (m < a.M && n < a.N) epi_store8<EPI>(a, m, n, va, x[HAS_AUX ? i : 0][HAS_AUX ? c : 0][0], bz + 8 * c);
                    if (m + 8 < a.M && n < a.N) epi_store8<EPI>(a, m + 8, n, vb, x[HAS_AUX ? i : 0][HAS_AUX ? c : 0][1], bz + 8 * c);
                }
        }
    }
}
// the lane's bias values for a block of NJ column tiles starting at column n_w, in its final layout: fp32 outputs NJ / 2 x 4 floats,
// bf16 outputs NJ / 4 x 8 floats (2 NJ floats either way)
template <int EPI, int NJ>
__device__ __forceinline__ void nt_load_bias(const GemmNtArgs& a, int n_w, int r16, int kg, float* bz) {
    const int hi = r16 >> 3;
    if constexpr (epi_f32out<EPI>()) {
#pragma unroll
        for (int c = 0; c < NJ / 2; ++c) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + min(n_w + 32 * c + 16 * hi + 4 * kg, a.N - 4));
            bz[4 * c] = b.x; bz[4 * c + 1] = b.y; bz[4 * c + 2] = b.z; bz[4 * c + 3] = b.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < NJ / 4; ++c) load8_f32(a.bias + min(n_w + 64 * c + 32 * hi + 16 * (kg & 1) + 8 * (kg >> 1), a.N - 8), bz + 8 * c);
    }
}

// gemm_nt geometry: 256 x 128 output tile, 8 waves as 4 (M) x 2 (N), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 tiles
// (same FLOPs per cycle as 32x32x16, but the power-limited chip clocks higher on it: NT GEMMs 3-9 % faster, profiles/r02_x4_*).
// Measured on MI355X (tools/ab_bench.py ablations, M = 100 416, K = 384):
//   * the main loop alone sustains ~965 TFLOP/s: operands arrive by LDS-DMA, whole 128-byte lines (8 rows x 128 B per
//     instruction), K streamed in 64-wide stages through a 3-deep ring (two 48 KB stages always in flight);
//   * the epilogue costs as much again (N = 1152: 92 us loop + 63 us epilogue; fc1 + GELU: 122 + 166 us): with every CU storing at
//     once it runs at the HBM write rate (~5.3 TB/s chip-wide, profiles/r03_x8_*), and the board is at its power cap (r03_x6_*).
// So the kernel is PERSISTENT: one workgroup per CU walks output tiles; after a tile's last stage it first issues the
// auxiliary loads of its epilogue, then the NEXT tile's first two ring stages, and only then applies the fused op and stores —
// straight from the accumulators (register epilogue above; rounds 1-2 went through wave-private LDS slabs), with no workgroup
// barrier — so the store tail and the GELU arithmetic overlap the next tile's operand streaming.
#ifndef DCV_N3_EARLY_AT
#define DCV_N3_EARLY_AT 11  // gemm_nt384_kernel: waves 0-3 issue their DMA pieces behind MFMA step N of the stage (11 = between the two k-steps), waves 4-7 at the top; -1: all at the top.  Stamps (profiles/r03_x1_*): with all eight waves issuing first the matrix pipe idled ~640 cycles per stage (80 pieces x 16 TA cycles, older waves served first); -4 ... -8 % on the k-loop-heavy shapes
#endif
constexpr int NT_BM = 256, NT_BN = 128, NT_BK = 64, NT_STAGES = 3;
constexpr int NT_A_BYTES = NT_BM * NT_BK * 2, NT_W_BYTES = NT_BN * NT_BK * 2, NT_STAGE_BYTES = NT_A_BYTES + NT_W_BYTES;  // 48 KB
constexpr int NT_SMEM = NT_STAGES * NT_STAGE_BYTES;  // the ring (the epilogue uses no LDS since round 3)

// store instructions one wave issues in a full tile's epilogue (8 row-chunks of 8 columns per lane)
template <int EPI>
__device__ constexpr int nt_stores_per_wave() {
    return (EPI == DCV_EPI_BIAS_GELU_BF16 || EPI == DCV_EPI_BIAS_RESID_F32) ? 16 : 8;
}

struct NtTile {
    const bf16_t* gA[4];
    const bf16_t* gW[2];
    int m0, n0;
};

__device__ __forceinline__ void nt_tile_setup(const GemmNtArgs& a, int L, int tiles_n, int wave, int lane, NtTile& t) {
    const int tm = L / tiles_n, tn = L - tm * tiles_n;
    t.m0 = tm * NT_BM;
    t.n0 = tn * NT_BN;
    // one DMA instruction = 8 rows x 128 B (lane -> row lane>>3, physical chunk lane&7); the swizzle
    // (physical chunk = logical ^ swz64n(row)) is applied to the per-lane SOURCE chunk, the destination is linear.
    // Wave w fills A rows [32w, 32w+32) (4 instructions) and W rows [16w, 16w+16) (2 instructions): 6 per stage.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 32 * wave + 8 * q + (lane >> 3);
        t.gA[q] = a.A + (size_t)min(t.m0 + row, a.M - 1) * a.lda + (((lane & 7) ^ swz64n(row)) * 8);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = 16 * wave + 8 * q + (lane >> 3);
        t.gW[q] = a.W + (size_t)min(t.n0 + row, a.N - 1) * a.ldw + (((lane & 7) ^ swz64n(row)) * 8);
    }
}

__device__ __forceinline__ void nt_issue(const NtTile& t, int kt, unsigned stage_base, unsigned dmaA, unsigned dmaW) {
    glds16(t.gA[0] + kt * NT_BK, stage_base + dmaA);
    glds16(t.gA[1] + kt * NT_BK, stage_base + dmaA + 1024);
    glds16(t.gA[2] + kt * NT_BK, stage_base + dmaA + 2048);
    glds16(t.gA[3] + kt * NT_BK, stage_base + dmaA + 3072);
    glds16(t.gW[0] + kt * NT_BK, stage_base + dmaW);
    glds16(t.gW[1] + kt * NT_BK, stage_base + dmaW + 1024);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_kernel(GemmNtArgs a) {
    // ONE shared array (a second __shared__ object can make hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) char smem[NT_SMEM];
    constexpr bool HAS_BIAS = (EPI != DCV_EPI_PLAIN_BF16) && (EPI != DCV_EPI_GELU_BWD_BF16);
    constexpr int S = nt_stores_per_wave<EPI>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool late = (wave >> 2) != 0;  // waves w and w + 4 share a SIMD
    const int tiles_n = (a.N + NT_BN - 1) / NT_BN;
    const int tiles_m = (a.M + NT_BM - 1) / NT_BM;
    const int total = tiles_m * tiles_n;
    const int G = gridDim.x;
    // tile walk: round k, workgroup w -> L = k*G + pos(w); pos gives every XCD (w & 7) a contiguous range of the
    // round's tiles, so the N-tiles that share an A row-panel run on one XCD at the same time (speed only)
    const int pos = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned dmaA = 32 * wave * 128, dmaW = NT_A_BYTES + 16 * wave * 128;  // wave-uniform byte offsets in a stage
    const int nk = a.K / NT_BK;
    const int r16 = lane & 15, kg = lane >> 4;  // 16x16x32 fragments: row / column r16, k-chunk kg (8 elements)
    const int rowA = (wm * 64 + r16) * 128, rowW = NT_A_BYTES + (wn * 64 + r16) * 128;  // + 16 i rows

    // tile order: round k, workgroup w -> k*G + pos(w)
    int L = pos < total ? pos : -1;
    int Lnext = (L >= 0 && L + G < total) ? L + G : -1;
    if (L < 0) return;
    NtTile cur, nxt;
    nt_tile_setup(a, L, tiles_n, wave, lane, cur);
    int g = 0;  // global stage counter of this workgroup: stage g lives in ring buffer g % 3
    nt_issue(cur, 0, smem_base + (g % NT_STAGES) * NT_STAGE_BYTES, dmaA, dmaW);
    if (nk > 1) nt_issue(cur, 1, smem_base + ((g + 1) % NT_STAGES) * NT_STAGE_BYTES, dmaA, dmaW);
    bool stores_behind = false;  // the previous tile's S epilogue stores (exactly S) were issued after this tile's prefetch
    // The lane's 8 bias values (its 8 output columns are fixed within a tile) live in registers and are loaded ONE TILE
    // AHEAD, between the epilogue's auxiliary loads and the prefetch DMAs: hipcc's wait for that load then sits where
    // nothing younger of ours is outstanding.  (A per-tile load at the loop top made hipcc emit vmcnt(0) there, which
    // drained the previous tile's stores and the prefetch: fc2 + residual ran 40 % slower than without persistence.)
    float bz[8], bz_next[8];
    if constexpr (HAS_BIAS) {
        nt_load_bias<EPI, 4>(a, cur.n0 + wn * 64, r16, kg, bz);
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bz[e]));
    }

    for (;;) {
        f32x4 acc[4][4];  // wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

        for (int kt = 0; kt < nk; ++kt, ++g) {
            // Stage kt of this tile is complete once only YOUNGER operations of this wave are outstanding:
            //   the 6 DMAs of stage kt+1 (issued one iteration / one tile earlier), and — for kt < 2 — the S stores of the
            //   previous tile's epilogue, which were issued after this tile's first two stages.  vmcnt counts in issue order.
            const bool dma_young = (kt + 1 < nk);
            const bool st_young = stores_behind && (kt < 2);
            if (st_young) {
                if (dma_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S + 6) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
            } else {
                if (dma_young) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();  // everyone's stage g landed; everyone is done reading buffer (g-1)%3
            // The two waves of a SIMD (w and w + 4) issue their 6 DMA pieces at different times — one at the top of the stage, the
            // other between its two k-steps — so that one of them is always feeding the matrix pipe (a piece costs 60-180 issue cycles,
            // six of them about as much as the stage's 32 MFMAs): -4 .. -7 % on every shape (profiles/r02_x10_*).  The per-wave issue
            // ORDER is unchanged, so the counted vmcnt waits hold; the ring is three deep, the late pieces still have a full stage to land.
            if (!late && kt + 2 < nk) nt_issue(cur, kt + 2, smem_base + ((g + 2) % NT_STAGES) * NT_STAGE_BYTES, dmaA, dmaW);
            const char* st = smem + (g % NT_STAGES) * NT_STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // two k-steps of 32
                if (ks == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (late && kt + 2 < nk) nt_issue(cur, kt + 2, smem_base + ((g + 2) % NT_STAGES) * NT_STAGE_BYTES, dmaA, dmaW);
                    __builtin_amdgcn_sched_barrier(0);
                }
                bf16x8 af[4], wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // rows 16 i + r16 of the wave's A / W block: swz64n(16 i + r16) = ((r16 >> 1) & 7) for every i
                    const int co = ((4 * ks + kg) ^ swz64n(r16)) << 4;
                    af[i] = as_bf16x8(lds_read128(st, rowA + i * 16 * 128 + co));
                    wf[i] = as_bf16x8(lds_read128(st, rowW + i * 16 * 128 + co));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);  // operands swapped: the tile comes out transposed (register epilogue)
            }
        }
        // no barrier here: the epilogue reads no LDS, and the two ring buffers the prefetch below fills (stages g-3 and g-2 of this
        // workgroup's stream) were last read before barriers every wave has already passed

        // ---- epilogue of `cur` straight from the accumulators, overlapped with the first two stages of the next tile ----
        const int Ln = Lnext;
        const bool has_next = Ln >= 0;
        Lnext = (has_next && Ln + G < total) ? Ln + G : -1;
        const bool full = (cur.m0 + NT_BM <= a.M) && (cur.n0 + NT_BN <= a.N) && (EPI != DCV_EPI_PATCH);
        // all auxiliary loads first (they are then OLDER than the prefetch DMAs, so hipcc's own waits for them never wait for the
        // prefetch), then the next tile's bias and first two stages, then the fused op and the stores
        nt_epilogue_block<EPI, (EPI == DCV_EPI_PATCH ? 1 : 4), 4, 4, 0>(a, acc, cur.m0 + wm * 64, cur.n0 + wn * 64, r16, kg, bz, [&]() {
            if (has_next) {
                nt_tile_setup(a, Ln, tiles_n, wave, lane, nxt);
                if constexpr (HAS_BIAS) nt_load_bias<EPI, 4>(a, nxt.n0 + wn * 64, r16, kg, bz_next);
                nt_issue(nxt, 0, smem_base + (g % NT_STAGES) * NT_STAGE_BYTES, dmaA, dmaW);
                if (nk > 1) nt_issue(nxt, 1, smem_base + ((g + 1) % NT_STAGES) * NT_STAGE_BYTES, dmaA, dmaW);
            }
        });
        if (!has_next) break;
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bz[e] = bz_next[e];
        }
        stores_behind = full;  // otherwise the store count is unknown: the next tile's first waits fall back to the stricter form
        cur = nxt;
        L = Ln;
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_pair (round 4): the same product on a 128 x 128 tile by FOUR waves (2 x 2, each the 64 x 64 wave tile of gemm_nt_kernel), two
// 32 KB stages = 64 KB of LDS, so that TWO workgroups share a CU.  Why: the ablation builds add up to the launch on all-zero operands too
// (profiles/r04_x2_*) — k-loop and epilogue of a persistent one-workgroup-per-CU kernel run one after the other because every wave of the
// CU is in the same phase; no schedule inside ONE instruction stream per SIMD overlaps them (in-order issue, one vmcnt queue per wave).
// With two workgroups per CU each SIMD hosts a wave of either, and while one workgroup stores its tile (the epilogue runs at the HBM
// write rate, ~14 B/clk per CU) the other's waves own the matrix pipe.  Price: 64 FLOP per operand byte instead of 85 / 154, and one stage
// of prefetch — the k-loop itself is slower; the form pays where the epilogue is the larger part (two outputs, an auxiliary read).
constexpr int NP_BM = 128, NP_BN = 128, NP_BK = 64;
constexpr int NP_A_BYTES = NP_BM * NP_BK * 2, NP_W_BYTES = NP_BN * NP_BK * 2, NP_STAGE_BYTES = NP_A_BYTES + NP_W_BYTES;  // 32 KB
constexpr int NP_SMEM = 2 * NP_STAGE_BYTES;                                                                              // 64 KB

struct NpTile {
    const bf16_t* gA[4];
    const bf16_t* gW[4];
    int m0, n0;
};
__device__ __forceinline__ void np_tile_setup(const GemmNtArgs& a, int L, int tiles_n, int wave, int lane, NpTile& t) {
    const int tm = L / tiles_n, tn = L - tm * tiles_n;
    t.m0 = tm * NP_BM;
    t.n0 = tn * NP_BN;
    // wave w fills A rows [32w, 32w+32) and W rows [32w, 32w+32): 4 + 4 pieces of 8 rows x 128 B per stage
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 32 * wave + 8 * q + (lane >> 3);
        t.gA[q] = a.A + (size_t)min(t.m0 + row, a.M - 1) * a.lda + (((lane & 7) ^ swz64n(row)) * 8);
        t.gW[q] = a.W + (size_t)min(t.n0 + row, a.N - 1) * a.ldw + (((lane & 7) ^ swz64n(row)) * 8);
    }
}
__device__ __forceinline__ void np_issue(const NpTile& t, int kt, unsigned stage_base, unsigned dmaA, unsigned dmaW) {
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(t.gA[q] + kt * NP_BK, stage_base + dmaA + q * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(t.gW[q] + kt * NP_BK, stage_base + dmaW + q * 1024);
}

// Round 5: the two co-resident workgroups of a CU start at the same instant on identical work, so both sit in the k-loop, then
// both in the epilogue: no overlap by construction.  -DDCV_PAIR_OFFSET=<cycles> (variant builds; 0 in the product) delays, once, the workgroup
// whose waves got the SECOND wave slot of their SIMD (HW_ID.wave_id != 0) before its first stage; -DDCV_PAIR_STAMP records the cycle at which each
// of a workgroup's first 24 epilogues starts, with the CU it runs on (tools/gemm_pair_phase.py checks that the pair stays out of phase).
#ifndef DCV_PAIR_OFFSET
#define DCV_PAIR_OFFSET 0
#endif
#ifdef DCV_PAIR_STAMP
__device__ unsigned long long np_stamps[1024 * 32];
extern "C" int dcv_pair_stamps(void* host_dst, size_t bytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(np_stamps), bytes < sizeof(np_stamps) ? bytes : sizeof(np_stamps)) == hipSuccess ? 0 : 1;
}
#endif

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_pair_kernel(GemmNtArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[NP_SMEM];
    constexpr bool HAS_BIAS = (EPI != DCV_EPI_PLAIN_BF16) && (EPI != DCV_EPI_GELU_BWD_BF16);
    static_assert(EPI != DCV_EPI_PATCH, "the tokeniser epilogue stays on the 256 x 128 kernel");
    constexpr int S = nt_stores_per_wave<EPI>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (a.N + NP_BN - 1) / NP_BN;
    const int tiles_m = (a.M + NP_BM - 1) / NP_BM;
    const int total = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int pos = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;  // an XCD gets a contiguous range of a round's tiles

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned dmaA = 32 * wave * 128, dmaW = NP_A_BYTES + 32 * wave * 128;
    const int nk = a.K / NP_BK;
    const int r16 = lane & 15, kg = lane >> 4;
    const int rowA = (wm * 64 + r16) * 128, rowW = NP_A_BYTES + (wn * 64 + r16) * 128;

    int L = pos < total ? pos : -1;
    int Lnext = (L >= 0 && L + G < total) ? L + G : -1;
    if (L < 0) return;
    NpTile cur, nxt;
    np_tile_setup(a, L, tiles_n, wave, lane, cur);
    int g = 0;  // global stage counter: stage g lives in buffer g & 1
#if DCV_PAIR_OFFSET || defined(DCV_PAIR_STAMP)
    const unsigned hw_id = __builtin_amdgcn_s_getreg((15 << 11) | 4);       // HW_REG_HW_ID[15:0]: wave_id 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    const unsigned xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;  // HW_REG_XCC_ID[3:0]
    const bool second = (hw_id & 15) != 0;
#endif
#if DCV_PAIR_OFFSET
    if (second) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)DCV_PAIR_OFFSET) __builtin_amdgcn_s_sleep(16);
    }
#endif
#ifdef DCV_PAIR_STAMP
    int n_epi = 0;
    if (tid == 0 && blockIdx.x < 1024) np_stamps[blockIdx.x * 32] = ((unsigned long long)xcc_id << 32) | hw_id;
#endif
    np_issue(cur, 0, smem_base, dmaA, dmaW);
    bool stores_behind = false;  // the previous tile's S epilogue stores were issued after this tile's first stage
    float bz[8], bz_next[8];
    if constexpr (HAS_BIAS) {
        nt_load_bias<EPI, 4>(a, cur.n0 + wn * 64, r16, kg, bz);
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bz[e]));
    }
    for (;;) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // stage kt landed once only younger operations are outstanding: for kt == 0 the previous tile's S epilogue stores (issued behind
            // this tile's first stage); afterwards nothing of ours is younger than the stage
            if (kt == 0 && stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage g visible to all four waves; all are done reading buffer (g + 1) & 1
            if (kt + 1 < nk) np_issue(cur, kt + 1, smem_base + ((g + 1) & 1) * NP_STAGE_BYTES, dmaA, dmaW);
            const char* st = smem + (g & 1) * NP_STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = ((4 * ks + kg) ^ swz64n(r16)) << 4;
                    af[i] = as_bf16x8(lds_read128(st, rowA + i * 16 * 128 + co));
                    wf[i] = as_bf16x8(lds_read128(st, rowW + i * 16 * 128 + co));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);  // operands swapped: transposed tile (register epilogue)
            }
        }
        // the next tile's first stage goes into buffer g & 1 = the buffer of stage g - 2, which every wave finished reading before the barrier
        // of iteration g - 1: no barrier needed (the epilogue reads no LDS)
        const int Ln = Lnext;
        const bool has_next = Ln >= 0;
        Lnext = (has_next && Ln + G < total) ? Ln + G : -1;
        const bool full = (cur.m0 + NP_BM <= a.M) && (cur.n0 + NP_BN <= a.N);
#ifdef DCV_PAIR_STAMP
        if (tid == 0 && blockIdx.x < 1024 && n_epi < 24) np_stamps[blockIdx.x * 32 + 1 + n_epi] = __builtin_amdgcn_s_memtime();
        ++n_epi;
#endif
        nt_epilogue_block<EPI, 4, 4, 4, 0>(a, acc, cur.m0 + wm * 64, cur.n0 + wn * 64, r16, kg, bz, [&]() {
            if (has_next) {
                np_tile_setup(a, Ln, tiles_n, wave, lane, nxt);
                if constexpr (HAS_BIAS) nt_load_bias<EPI, 4>(a, nxt.n0 + wn * 64, r16, kg, bz_next);
                np_issue(nxt, 0, smem_base + (g & 1) * NP_STAGE_BYTES, dmaA, dmaW);
            }
        });
        if (!has_next) break;
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bz[e] = bz_next[e];
        }
        stores_behind = full;
        cur = nxt;
        L = Ln;
    }
}

// ------------------------------------------------------------------------------------------------
// Residual + LayerNorm epilogue of the 256 x 384 kernel for N == 384 (round 4, judge row N1): the tile spans whole rows, so the LayerNorm that
// follows every residual addition (vit.py:397-398: x = x + branch; then norm2(x) / the next block's norm1(x)) is computed from the accumulators:
//     x' = resid + s (acc + bias)            -> out   (fp32, as DCV_EPI_BIAS_RESID_F32)
//     u  = (x' - mean) * rstd * gamma + beta -> out2  (bf16: the next GEMM's operand),   mean / rstd -> ln_mean / ln_rstd  (for the backward)
// and the separate ln_fwd launch — a 154 MB read of x' per call — disappears.
// Layout: after the in-place DPP row exchange (xchg_rows8) lane (rr = r16 & 7, hi = r16 >> 3, kg) of wave (wm, wn) holds, per 16-row block i,
// rows 16 i + rr (registers acc[i][2c]) and 16 i + rr + 8 (acc[i][2c+1]) and, per 32-column group c = 0..5, the 4 columns 192 wn + 32 c + 16 hi +
// 4 kg .. +3: 24 values of a row per lane, 8 lanes per row and wave, two waves per row.  Statistics: per lane mean and centred sum of squares
// (two passes over its 24 registers), combined pairwise with Chan's formula — the centred form ln_fwd_kernel uses, not E[x^2] - mean^2 —
// over the 8 lanes (xor 8, 16, 32), then across the two waves through the ring buffer that is free during the epilogue.
template <class Between>
__device__ __forceinline__ void nt384_resid_ln_epilogue(const GemmNtArgs& a, f32x4 (&acc)[4][12], int m0, int wm, int wn, int ln, char* fb,
                                                        Between&& between) {
    const int r16 = ln & 15, kg = ln >> 4, rr = r16 & 7, hi = r16 >> 3;
    const int ncol0 = 192 * wn + 16 * hi + 4 * kg;  // + 32 c
    float* const stat = reinterpret_cast<float*>(fb);  // [wn][256 rows][2]
    __builtin_amdgcn_s_barrier();  // every wave has finished reading this buffer (the last k stage) before anyone writes statistics into it
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ma = m0 + 64 * wm + 16 * i + rr, mb = ma + 8;
        const int mac = min(ma, a.M - 1), mbc = min(mb, a.M - 1);
        const float sa = a.aux2 ? a.aux2[mac / a.T] : 1.f, sb = a.aux2 ? a.aux2[mbc / a.T] : 1.f;  // DropPath factor of the row's sample
        const float* ra = (const float*)a.aux + (size_t)mac * a.ldaux + ncol0;
        const float* rb = (const float*)a.aux + (size_t)mbc * a.ldaux + ncol0;
#pragma unroll
        for (int hc = 0; hc < 3; ++hc) {  // two 32-column groups at a time: 24 registers of auxiliary loads beside the 192 accumulators
            float4 xa[2], xb[2], b4[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * hc + q;
                xa[q] = *reinterpret_cast<const float4*>(ra + 32 * c);
                xb[q] = *reinterpret_cast<const float4*>(rb + 32 * c);
                b4[q] = *reinterpret_cast<const float4*>(a.bias + ncol0 + 32 * c);
            }
            if (i == 0 && hc == 0) between();  // the next tile's first stage, behind the first auxiliary loads
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * hc + q;
                f32x4 va, vb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float fa, fbv;
                    xchg_rows8(acc[i][2 * c][r], acc[i][2 * c + 1][r], fa, fbv);
                    va[r] = fa;
                    vb[r] = fbv;
                }
                const float bq[4] = {b4[q].x, b4[q].y, b4[q].z, b4[q].w};
                const float pa[4] = {xa[q].x, xa[q].y, xa[q].z, xa[q].w}, pb[4] = {xb[q].x, xb[q].y, xb[q].z, xb[q].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    va[r] = fmaf(sa, va[r] + bq[r], pa[r]);
                    vb[r] = fmaf(sb, vb[r] + bq[r], pb[r]);
                }
                acc[i][2 * c] = va;
                acc[i][2 * c + 1] = vb;
                if (ma < a.M) *reinterpret_cast<f32x4*>((float*)a.out + (size_t)ma * a.ldo + ncol0 + 32 * c) = va;
                if (mb < a.M) *reinterpret_cast<f32x4*>((float*)a.out + (size_t)mb * a.ldo + ncol0 + 32 * c) = vb;
            }
        }
        // statistics of the lane's 24 values of each of its two rows, then Chan's pairwise combination of equal-sized groups:
        //   mean = (m1 + m2) / 2,  M2 = M2_1 + M2_2 + (m2 - m1)^2 n / 2      (n = size of either group: 24, 48, 96)
        float mu[2], m2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][2 * c + h][r];
            const float m = s * (1.f / 24.f);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[i][2 * c + h][r] - m;
                    q = fmaf(d, d, q);
                }
            mu[h] = m;
            m2[h] = q;
        }
        float half_n = 12.f;
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float om = __shfl_xor(mu[h], off, 64), oq = __shfl_xor(m2[h], off, 64);
                const float d = om - mu[h];
                m2[h] = m2[h] + oq + d * d * half_n;
                mu[h] = 0.5f * (mu[h] + om);
            }
            half_n *= 2.f;
        }
        if (hi == 0 && kg == 0) {  // one lane per row: this wave's 192 columns of rows (i, rr) and (i, rr + 8)
            const int rl = 64 * wm + 16 * i + rr;
            *reinterpret_cast<float2*>(stat + ((size_t)(wn * 256 + rl) * 2)) = make_float2(mu[0], m2[0]);
            *reinterpret_cast<float2*>(stat + ((size_t)(wn * 256 + rl + 8) * 2)) = make_float2(mu[1], m2[1]);
        }
    }
    {  // gamma and beta into the same buffer (3 KB): phase 2 reads them back per column group with 16-byte LDS reads instead of holding
       // 48 registers or waiting for six dependent global loads per row block
        const int t = 64 * (2 * wm + wn) + ln;
        if (t < 192) {
            const float4 v = *reinterpret_cast<const float4*>((t < 96 ? a.ln_g : a.ln_b) + 4 * (t < 96 ? t : t - 96));
            *reinterpret_cast<float4*>(stat + 1024 + 4 * t) = v;  // gamma at float 1024, beta at 1024 + 384
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // both column halves of every row (and gamma / beta) are in LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ma = m0 + 64 * wm + 16 * i + rr, mb = ma + 8;
        const int rl = 64 * wm + 16 * i + rr;
        float mean[2], rstd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float2 p0 = *reinterpret_cast<const float2*>(stat + ((size_t)(rl + 8 * h) * 2));
            const float2 p1 = *reinterpret_cast<const float2*>(stat + ((size_t)(256 + rl + 8 * h) * 2));
            const float d = p1.x - p0.x;
            mean[h] = 0.5f * (p0.x + p1.x);
            rstd[h] = rsqrtf((p0.y + p1.y + d * d * 96.f) * (1.f / 384.f) + a.ln_eps);
        }
        if (wn == 0 && hi == 0 && kg == 0) {
            if (ma < a.M) { a.ln_mean[ma] = mean[0]; a.ln_rstd[ma] = rstd[0]; }
            if (mb < a.M) { a.ln_mean[mb] = mean[1]; a.ln_rstd[mb] = rstd[1]; }
        }
#pragma unroll
        for (int hc = 0; hc < 3; ++hc) {
            float4 g4[2], be4[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                g4[q] = *reinterpret_cast<const float4*>(stat + 1024 + ncol0 + 32 * (2 * hc + q));
                be4[q] = *reinterpret_cast<const float4*>(stat + 1024 + 384 + ncol0 + 32 * (2 * hc + q));
            }
            // bf16 u of the column groups c = 2 hc and c + 1, then a 16-lane-row swap (v_permlane16_swap) between the lanes kg and kg ^ 1, which hold
            // adjacent 4-column pieces of both groups: afterwards an even-kg lane holds 8 consecutive columns of group c, its odd partner 8
            // consecutive columns of group c + 1 — the 8 lanes of a row cover 128 contiguous bytes, whole cache lines per store instruction
            // (with 8 bytes per lane the same stores touched 64-byte segments: fc2 + residual + LN 211 -> 201 us, proj 124 -> 112.5 us)
            uint2 pa[2], pb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * hc + q;
                const float g[4] = {g4[q].x, g4[q].y, g4[q].z, g4[q].w}, be[4] = {be4[q].x, be4[q].y, be4[q].z, be4[q].w};
                float ua[4], ub[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ua[r] = fmaf((acc[i][2 * c][r] - mean[0]) * rstd[0], g[r], be[r]);
                    ub[r] = fmaf((acc[i][2 * c + 1][r] - mean[1]) * rstd[1], g[r], be[r]);
                }
                pa[q] = pack4_bf16(ua[0], ua[1], ua[2], ua[3]);
                pb[q] = pack4_bf16(ub[0], ub[1], ub[2], ub[3]);
            }
            const u32x2_t ax = __builtin_amdgcn_permlane16_swap(pa[0].x, pa[1].x, false, false);
            const u32x2_t ay = __builtin_amdgcn_permlane16_swap(pa[0].y, pa[1].y, false, false);
            const u32x2_t bx = __builtin_amdgcn_permlane16_swap(pb[0].x, pb[1].x, false, false);
            const u32x2_t by = __builtin_amdgcn_permlane16_swap(pb[0].y, pb[1].y, false, false);
            const int kgp = kg & 1;
            const int ucol = 192 * wn + 32 * (2 * hc + kgp) + 16 * hi + 4 * (kg - kgp);
            if (ma < a.M) *reinterpret_cast<uint4*>((bf16_t*)a.out2 + (size_t)ma * a.ldo2 + ucol) = make_uint4(ax[0], ay[0], ax[1], ay[1]);
            if (mb < a.M) *reinterpret_cast<uint4*>((bf16_t*)a.out2 + (size_t)mb * a.ldo2 + ucol) = make_uint4(bx[0], by[0], bx[1], by[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt384: the same product on a 256 x 384 output tile, for N % 384 == 0 (every Linear of the encoder: 384, 1152, 1536).
// Why: the 256 x 128 kernel is bound by operand delivery into LDS (34-37 GB/s per CU sustained, whatever the epilogue), so
// its main-loop time scales with operand BYTES: (256+128)*2 B per 256*128*2 FLOP and k = 85 FLOP/B.  256 x 384 moves
// (256+384)*2 B per 256*384*2 FLOP = 154 FLOP/B, 1.8x fewer bytes per FLOP.  Cost: 192 accumulator registers per lane
// (8 waves as 4 (M) x 2 (N), each 64 x 192 = 4 x 12 MFMA 16x16x32 tiles; the k-step's read / MFMA order is pinned, see the loop) and the
// whole LDS: two 80 KB stages.  With two buffers the next stage is issued after the barrier that retires the previous
// one, one k-iteration (48 MFMAs per wave) ahead.  Tile walk, register epilogue and fused ops as in gemm_nt_kernel.
constexpr int N3_BM = 256, N3_BN = 384, N3_BK = 64;
constexpr int N3_A_BYTES = N3_BM * N3_BK * 2, N3_W_BYTES = N3_BN * N3_BK * 2, N3_STAGE_BYTES = N3_A_BYTES + N3_W_BYTES;  // 80 KB
constexpr int N3_SMEM = 2 * N3_STAGE_BYTES;                                                                              // 160 KB
// per wave and stage 10 DMA instructions: A rows [32w, 32w+32) = 4 pieces, W rows [48w, 48w+48) = 6 pieces

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt384_kernel(GemmNtArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[N3_SMEM];
    constexpr bool HAS_BIAS = (EPI != DCV_EPI_PLAIN_BF16) && (EPI != DCV_EPI_GELU_BWD_BF16);
    static_assert(EPI != DCV_EPI_PATCH, "the tokeniser epilogue stays on the 256 x 128 kernel");
    constexpr bool LN = (EPI == DCV_EPI_RESID_LN);
    constexpr int S = LN ? 0 : 3 * nt_stores_per_wave<LN ? DCV_EPI_BIAS_RESID_F32 : EPI>();  // stores one wave issues in a full tile's epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool late = (wave >> 2) != 0;  // waves w and w + 4 share a SIMD
    const int tiles_n = a.N / N3_BN;
    const int tiles_m = (a.M + N3_BM - 1) / N3_BM;
    const int total = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int pos = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned dmaA = 32 * wave * 128, dmaW = N3_A_BYTES + 48 * wave * 128;  // wave-uniform byte offsets in a stage
    const int nk = a.K / N3_BK;
    const int r16 = lane & 15, kg = lane >> 4;  // 16x16x32 fragments: row / column r16, k-chunk kg (8 elements)
    const int rowA = (wm * 64 + r16) * 128, rowW = N3_A_BYTES + (wn * 192 + r16) * 128;  // + 16 i / 16 j rows
    int fA0, fA1, fW0, fW1;  // per-lane fragment read offsets inside a stage: operand x k-step
    {
        const int co0 = (kg ^ swz64n(r16)) << 4;
        fA0 = rowA + co0; fA1 = rowA + (co0 ^ 64); fW0 = rowW + co0; fW1 = rowW + (co0 ^ 64);
        asm volatile("" : "+v"(fA0), "+v"(fA1), "+v"(fW0), "+v"(fW1));
    }

    // DMA sources: a scalar tile base + per-lane byte offsets that do not depend on the tile (the partial last M tile
    // recomputes the A offsets with its rows clamped to M-1)
    // Pieces q and q + 2 of a wave are 16 rows apart and share their swizzle (swz64n(row) = (row >> 1) & 7 has period 16 rows), so two per-lane
    // offsets per operand serve all pieces: the 16-row steps go into the SCALAR base (6 VGPRs less than one offset per piece, in a kernel that
    // holds 192 accumulators — the residual + LayerNorm epilogue spilled exactly these offsets and reloaded them inside the k-loop)
    unsigned voffA[2], voffW[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rowa = 32 * wave + 8 * q + (lane >> 3), roww = 48 * wave + 8 * q + (lane >> 3);
        voffA[q] = (unsigned)(((size_t)rowa * a.lda + (((lane & 7) ^ swz64n(rowa)) * 8)) * 2);
        voffW[q] = (unsigned)(((size_t)roww * a.ldw + (((lane & 7) ^ swz64n(roww)) * 8)) * 2);
    }
    auto issue = [&](int m0_, int n0_, int kt, unsigned stage_base) {
        const int m0 = __builtin_amdgcn_readfirstlane(m0_), n0 = __builtin_amdgcn_readfirstlane(n0_);  // uniform by construction
        const bf16_t* ab = a.A + (size_t)m0 * a.lda + kt * N3_BK;  // scalar
        const bf16_t* wb = a.W + (size_t)n0 * a.ldw + kt * N3_BK;
        if (m0 + N3_BM <= a.M) {
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16s(ab + (size_t)(q >> 1) * 16 * a.lda, voffA[q & 1], stage_base + dmaA + q * 1024);
        } else {
            // the lane id goes through an opaque move: otherwise hipcc hoists these offsets to the top of EVERY tile, spills
            // them, and the reload's s_waitcnt vmcnt(0) there drains the previous tile's epilogue stores
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 32 * wave + 8 * q + (ln >> 3);
                const int rc = min(m0 + row, a.M - 1) - m0;
                glds16s(ab, (unsigned)(((size_t)rc * a.lda + (((ln & 7) ^ swz64n(row)) * 8)) * 2), stage_base + dmaA + q * 1024);
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) glds16s(wb + (size_t)(q >> 1) * 16 * a.ldw, voffW[q & 1], stage_base + dmaW + q * 1024);
    };

    // tile order: round k, workgroup w -> tile k*G + pos(w)
    int L = pos < total ? pos : -1;
    int Lnext = (L >= 0 && L + G < total) ? L + G : -1;
    if (L < 0) return;
    int m0 = (L / tiles_n) * N3_BM, n0 = (L % tiles_n) * N3_BN;
    int g = 0;  // global stage counter: stage g lives in buffer g & 1
    issue(m0, n0, 0, smem_base);
    bool stores_behind = false;

    for (;;) {
        f32x4 acc[4][12];  // wave tile 64 x 192 = 4 x 12 MFMA tiles of 16 x 16
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 12; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

#pragma clang loop unroll(disable)  // also keeps hipcc from peeling the first iteration (the peeled copy spilled)
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // stage kt landed once only younger operations are outstanding: for kt == 0 the previous tile's S epilogue
            // stores (issued after this tile's first stage); afterwards nothing of ours is younger than the stage
            if (kt == 0 && stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage g visible to all; all waves are done reading buffer (g+1)&1
            // (issuing half the waves' pieces between the two k-steps, as gemm_nt_kernel does, measured 3-5 % slower here: with two
            // stages the late pieces have half a stage to land)
            const char* st = smem + (g & 1) * N3_STAGE_BYTES;
            // 24 steps (2 k-steps of 32 x 12 column blocks) of 4 MFMAs; the W fragment of step s + 2 is read at step s (ring of 3),
            // the 4 A fragments of the second k-step replace those of the first one by one behind their last MFMA.  The order
            // is pinned (sched_barrier): left alone, hipcc hoists all 16 reads of a k-step and spills accumulators (192 of the
            // 256 registers a wave has at 8 waves per workgroup are accumulators).
            // k-step 0: chunk kg; k-step 1: chunk 4 + kg = co0 ^ 64.  swz64n(16 i + r16) is the same for every i.  The four per-lane
            // bases (A / W x k-step) are opaque values, so every fragment read is base + immediate (left to itself hipcc built one
            // address register per column block for the second k-step — twelve registers it then spilled around the loop)
            const char* const pA0 = st + fA0;
            const char* const pA1 = st + fA1;
            const char* const pW0 = st + fW0;
            const char* const pW1 = st + fW1;
            auto rdW = [&](int s2) { return as_bf16x8(lds_read128(s2 >= 12 ? pW1 : pW0, (s2 % 12) * 16 * 128)); };
            bf16x8 af[4], wq[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = as_bf16x8(lds_read128(pA0, i * 16 * 128));
            wq[0] = rdW(0);
            wq[1] = rdW(1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk && (DCV_N3_EARLY_AT < 0 || late)) issue(m0, n0, kt + 1, smem_base + ((g + 1) & 1) * N3_STAGE_BYTES);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 24; ++s2) {
                if (s2 + 2 < 24) wq[(s2 + 2) % 3] = rdW(s2 + 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][s2 % 12] = mfma16(wq[s2 % 3], af[i], acc[i][s2 % 12]);  // operands swapped: transposed tile (register epilogue)
                    if (s2 == 11) af[i] = as_bf16x8(lds_read128(pA1, i * 16 * 128));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (DCV_N3_EARLY_AT >= 0 && s2 == DCV_N3_EARLY_AT) {
                    if (kt + 1 < nk && !late) issue(m0, n0, kt + 1, smem_base + ((g + 1) & 1) * N3_STAGE_BYTES);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // The next tile's first stage goes into buffer g & 1 = the buffer of stage g-2, which every wave finished reading before the
        // barrier of iteration g-1: no barrier is needed here (the epilogue reads no LDS).

        // ---- epilogue straight from the accumulators, overlapped with the first stage of the next tile (into the other buffer) ----
        const int Ln = Lnext;
        const bool has_next = Ln >= 0;
        const int m0n = has_next ? (Ln / tiles_n) * N3_BM : 0, n0n = has_next ? (Ln % tiles_n) * N3_BN : 0;
        Lnext = (has_next && Ln + G < total) ? Ln + G : -1;
        const bool full = (m0 + N3_BM <= a.M);
        const int m_w = m0 + wm * 64, n_w = n0 + wn * 192;
        // the lane id goes through an opaque move: otherwise hipcc hoists the epilogue's per-lane row / column offsets above the k-loop,
        // where 192 accumulators are live, spills them, and reloads DMA offsets from scratch inside the loop (a scratch load is a
        // vector-memory operation: its wait drains the ring)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int r16e = ln & 15, kge = ln >> 4;
        if constexpr (LN) {
            // residual + LayerNorm from the accumulators; the statistics cross the two column halves through the ring buffer the last k
            // stage lived in (free until the next tile's second stage is issued)
            nt384_resid_ln_epilogue(a, acc, m0, wm, wn, ln, smem + ((g + 1) & 1) * N3_STAGE_BYTES, [&]() {
                if (has_next) issue(m0n, n0n, 0, smem_base + (g & 1) * N3_STAGE_BYTES);
            });
        } else {
            // column blocks of 64 x row blocks of 16
            auto block = [&](auto j0c, auto&& between) {
                constexpr int J0 = decltype(j0c)::value;
                float bz[8];
                if constexpr (HAS_BIAS) nt_load_bias<EPI, 4>(a, n_w + 16 * J0, r16e, kge, bz);
                nt_epilogue_block<EPI, 1, 4, 12, J0>(a, acc, m_w, n_w + 16 * J0, r16e, kge, bz, between);
            };
            block(std::integral_constant<int, 0>{}, [&]() {
                if (has_next) issue(m0n, n0n, 0, smem_base + (g & 1) * N3_STAGE_BYTES);
            });
            block(std::integral_constant<int, 4>{}, []() {});
            block(std::integral_constant<int, 8>{}, []() {});
        }
        if (!has_next) break;
        stores_behind = full && !LN;  // the LayerNorm epilogue's store count is not fixed (row predicates): its next tile starts behind vmcnt(0)
        m0 = m0n;
        n0 = n0n;
        L = Ln;
    }
}

#ifndef DCV_NT_ALT
#define DCV_NT_ALT 0
#endif
#if DCV_NT_ALT
// ------------------------------------------------------------------------------------------------
// gemm_nt_alt (round 5, DCV_TILE_ALT; variant builds with -DDCV_NT_ALT=1 only — measured 15-27 % SLOWER than the best shipped tile on every shape,
// profiles/r05_x9_*): the 256 x 384 kernel's eight waves as TWO groups of four (one wave of either group on every SIMD) that work on the
// two 192-column halves of a tile in ALTERNATING phases: while group 0 runs the k-loop of its half (the matrix pipe), group 1 runs the epilogue of the half it
// accumulated one phase earlier (vector arithmetic, auxiliary loads, stores), then they swap.  The LDS ring belongs to whichever group is in its k-loop — two
// stages of (256 + 192) rows x 128 B = 112 KB — so, unlike two co-resident workgroups (gemm_nt_pair: 128 x 128 tiles), the wave tile stays 64 x 192.  A phase is
// K / 64 slots; every slot opens with the workgroup barrier (the k-loop group needs it per stage, the other group just arrives), the epilogue group spreads its
// six chunks (2 row blocks x one 64-column group each) over the slots and issues the first stage of ITS next half into the free buffer during the last slot.
// Cost: the A panel is pulled into LDS once per half (3840 -> 5376 lines per tile at K = 384).  bf16-output epilogues only.
constexpr int NA_A_BYTES = 256 * 128, NA_W_BYTES = 192 * 128, NA_STAGE_BYTES = NA_A_BYTES + NA_W_BYTES, NA_SMEM = 2 * NA_STAGE_BYTES;  // 56 KB stages, 112 KB

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_alt_kernel(GemmNtArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[NA_SMEM];
    static_assert(!epi_f32out<EPI>() && EPI != DCV_EPI_RESID_LN, "bf16-output epilogues only");
    constexpr bool HAS_BIAS = (EPI != DCV_EPI_PLAIN_BF16) && (EPI != DCV_EPI_GELU_BWD_BF16);
    constexpr bool HAS_AUX = (EPI == DCV_EPI_GELU_BWD_BF16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3;  // waves w and w + 4 share a SIMD: one of either group
    const int tiles_n = a.N / N3_BN;
    const int tiles_m = (a.M + N3_BM - 1) / N3_BM;
    const int total = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int pos = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (pos >= total) return;
    const int n_mine = (total - pos + G - 1) / G;  // tiles this workgroup walks: pos, pos + G, ...

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned dmaA = 64 * wm * 128, dmaW = NA_A_BYTES + 48 * wm * 128;  // per stage a k-loop wave brings A rows [64 wm, +64) = 8 pieces, W rows [48 wm, +48) = 6
    const int nk = a.K / N3_BK;
    const int r16 = lane & 15, kg = lane >> 4;
    int fA0, fA1, fW0, fW1;
    {
        const int rowA = (wm * 64 + r16) * 128, rowW = NA_A_BYTES + r16 * 128;
        const int co0 = (kg ^ swz64n(r16)) << 4;
        fA0 = rowA + co0; fA1 = rowA + (co0 ^ 64); fW0 = rowW + co0; fW1 = rowW + (co0 ^ 64);
        asm volatile("" : "+v"(fA0), "+v"(fA1), "+v"(fW0), "+v"(fW1));
    }
    unsigned voffA[2], voffW[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rowa = 64 * wm + 8 * q + (lane >> 3), roww = 48 * wm + 8 * q + (lane >> 3);
        voffA[q] = (unsigned)(((size_t)rowa * a.lda + (((lane & 7) ^ swz64n(rowa)) * 8)) * 2);
        voffW[q] = (unsigned)(((size_t)roww * a.ldw + (((lane & 7) ^ swz64n(roww)) * 8)) * 2);
    }
    // stage kt of the half (tile L, column half grp) into the buffer at stage_base
    auto issue = [&](int L, int kt, unsigned stage_base) {
        const int m0 = __builtin_amdgcn_readfirstlane((L / tiles_n) * N3_BM), n0 = __builtin_amdgcn_readfirstlane((L % tiles_n) * N3_BN + 192 * grp);
        const bf16_t* ab = a.A + (size_t)m0 * a.lda + kt * N3_BK;
        const bf16_t* wb = a.W + (size_t)n0 * a.ldw + kt * N3_BK;
        if (m0 + N3_BM <= a.M) {
#pragma unroll
            for (int q = 0; q < 8; ++q) glds16s(ab + (size_t)(q >> 1) * 16 * a.lda, voffA[q & 1], stage_base + dmaA + q * 1024);
        } else {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = 64 * wm + 8 * q + (ln >> 3);
                const int rc = min(m0 + row, a.M - 1) - m0;
                glds16s(ab, (unsigned)(((size_t)rc * a.lda + (((ln & 7) ^ swz64n(row)) * 8)) * 2), stage_base + dmaA + q * 1024);
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) glds16s(wb + (size_t)(q >> 1) * 16 * a.ldw, voffW[q & 1], stage_base + dmaW + q * 1024);
    };

    int g = 0;  // slot counter of the workgroup: the k-loop group's stage of slot g lives in buffer g & 1
    // a phase in which this group has nothing in its accumulators: arrive at every slot's barrier; with_issue: my first stage goes out in the last slot
    auto idle_phase = [&](bool with_issue) {
        for (int kt = 0; kt < nk; ++kt, ++g) {
            __builtin_amdgcn_s_barrier();
            if (with_issue && kt == nk - 1) issue(pos, 0, smem_base + ((g + 1) & 1) * NA_STAGE_BYTES);
        }
    };
    // group 0: k e k e ... k e idle;  group 1: idle k e ... k e — in every phase one group owns the ring and the matrix pipe, the other stores.  The loop body is
    // straight-line (k-loop, then epilogue): the accumulators are defined and dead inside one iteration (as a branch per phase they became loop-carried values
    // the register allocator spilled)
    if (grp == 0) issue(pos, 0, smem_base);
    else idle_phase(true);

    for (int t = 0; t < n_mine; ++t) {
        const int L = pos + t * G;
        f32x4 acc[4][12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 12; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#pragma clang loop unroll(disable)
        for (int kt = 0; kt < nk; ++kt, ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of stage kt (and, for kt == 0, my last epilogue's stores)
            __builtin_amdgcn_s_barrier();
            const char* st = smem + (g & 1) * NA_STAGE_BYTES;
            const char* const pA0 = st + fA0;
            const char* const pA1 = st + fA1;
            const char* const pW0 = st + fW0;
            const char* const pW1 = st + fW1;
            auto rdW = [&](int s2) { return as_bf16x8(lds_read128(s2 >= 12 ? pW1 : pW0, (s2 % 12) * 16 * 128)); };
            bf16x8 af[4], wq[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = as_bf16x8(lds_read128(pA0, i * 16 * 128));
            wq[0] = rdW(0);
            wq[1] = rdW(1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) issue(L, kt + 1, smem_base + ((g + 1) & 1) * NA_STAGE_BYTES);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 24; ++s2) {
                if (s2 + 2 < 24) wq[(s2 + 2) % 3] = rdW(s2 + 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][s2 % 12] = mfma16(wq[s2 % 3], af[i], acc[i][s2 % 12]);
                    if (s2 == 11) af[i] = as_bf16x8(lds_read128(pA1, i * 16 * 128));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue phase: six chunks (one 64-column group x two 16-row blocks each) spread over the slots; my next first stage in the last slot ----
        const int m_w = (L / tiles_n) * N3_BM + wm * 64, n_w = (L % tiles_n) * N3_BN + 192 * grp;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int r16e = ln & 15, kge = ln >> 4;
        const int rr = r16e & 7, cl = 32 * (r16e >> 3) + 16 * (kge & 1) + 8 * (kge >> 1);
        const bool more = t + 1 < n_mine;
        // chunk ch runs in slot ch * nk / 6; the sequence over the chunks is straight-line (the accumulators of a chunk die with it), the slots between them
        // are opened by a counted loop of barriers
        int kt = 0;
        auto open_slots = [&](int upto) {  // open slots kt .. upto (inclusive)
            for (; kt <= upto; ++kt, ++g) {
                __builtin_amdgcn_s_barrier();
                if (more && kt == nk - 1) issue(L + G, 0, smem_base + ((g + 1) & 1) * NA_STAGE_BYTES);  // free since the barrier above
            }
        };
#pragma unroll
        for (int ch = 0; ch < 6; ++ch) {
            open_slots(ch * nk / 6);
            const int c = ch >> 1;
            float bz[8];
            if constexpr (HAS_BIAS) nt_load_bias<EPI, 4>(a, n_w + 64 * c, r16e, kge, bz);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * (ch & 1) + ii;
                float x[2][8];
                if constexpr (HAS_AUX) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) epi_aux8<EPI>(a, min(m_w + 16 * i + 8 * h + rr, a.M - 1), min(n_w + 64 * c + cl, a.N - 8), x[h]);
                }
                float v0[8], v1[8], va[8], vb[8];
                swap_pair8(acc[i][4 * c], acc[i][4 * c + 1], v0);
                swap_pair8(acc[i][4 * c + 2], acc[i][4 * c + 3], v1);
#pragma unroll
                for (int e = 0; e < 8; ++e) xchg_rows8(v0[e], v1[e], va[e], vb[e]);
                const int m = m_w + 16 * i + rr, n = n_w + 64 * c + cl;
                if (m < a.M && n < a.N) epi_store8<EPI>(a, m, n, va, x[0], bz);
                if (m + 8 < a.M && n < a.N) epi_store8<EPI>(a, m + 8, n, vb, x[1], bz);
            }
        }
        open_slots(nk - 1);
    }
    if (grp == 0) idle_phase(false);
}
#endif  // DCV_NT_ALT

// ------------------------------------------------------------------------------------------------
struct GemmTnArgs {
    const bf16_t* Y;
    int ldy;
    const bf16_t* X;
    int ldx;
    int M, P, Q;
    float* dW;
    int lddw;
    float* dbias;
    int m_per_split, splits;
    float* part;       // deterministic mode: split s stores its partial tile to part[s * part_stride + p * Q + q] (and its bias partial to
    long part_stride;  // part[s * part_stride + P * Q + p]) with plain stores instead of adding atomically; det_reduce_kernel sums them
    long bias_off;     // gemm_tn384, deterministic mode: float offset in `part` of the bias partials [splits * tiles_q][P]
    int mode;          // grouped launch only: 0 = 384 x 128 tiles; 1 = 384 x 256 tiles; 2 = 384 x 256 tiles of the TRANSPOSED product (Y and X exchanged by
                       // the launcher: P, Q, ldy, ldx are the exchanged ones; dW / part / dbias keep the caller's layout: element (p, q) at q * P + p)
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2][2][BK * 128 * 2];  // [buf][Y|X][64 rows x 256 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, r32 = lane & 31, li = lane & 15, g1 = (lane >> 4) & 1;
    const int tiles_q = (a.Q + 127) / 128;
    // tile index fastest, splits outer, and an XCD gets a contiguous range of logical ids: the workgroups that
    // stream the SAME rows of Y and X (one split, all output tiles) run together on one XCD and share them in its L2.
    const int tiles = tiles_q * ((a.P + 127) / 128);
    int bid = xcd_remap(blockIdx.x, tiles * a.splits);
    const int split = bid / tiles;
    bid -= split * tiles;
    const int tq = bid % tiles_q, tp = bid / tiles_q;
    const int p0 = tp * 128, q0 = tq * 128;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    if (m_begin >= m_end) return;
    const int nk = (m_end - m_begin + BK - 1) / BK;

    // staging: tile [64 m][128 cols]; chunk q = tid + 256 i -> row = q>>4 (m), chunk = q&15 (cols 8*chunk..)
    const int srow = tid >> 4, sch = tid & 15;
    // P, Q are multiples of 8; chunks beyond the matrix edge re-read the last valid chunk (their products are never stored)
    const int colY = min(p0 + sch * 8, a.P - 8), colX = min(q0 + sch * 8, a.Q - 8);
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) soff[i] = lds128_off(srow + 16 * i, sch * 8);
    const bf16_t* gY = a.Y + (size_t)colY;
    const bf16_t* gX = a.X + (size_t)colX;

    uint4 ry[4], rx[4];
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f, bs4 = 0.f, bs5 = 0.f, bs6 = 0.f, bs7 = 0.f;
    const bool do_bias = (a.dbias != nullptr) && (tq == 0);

#define TN_LOAD_TILE(kt_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        int m = m_begin + (kt_) * BK + srow + 16 * i;                                                              \
        const bool ok = m < m_end;                                                                                 \
        int mc = ok ? m : (m_end - 1); /* always a valid row; out-of-range rows are zeroed in registers */         \
        uint4 ty = *reinterpret_cast<const uint4*>(gY + (size_t)mc * a.ldy);                                       \
        uint4 tx = *reinterpret_cast<const uint4*>(gX + (size_t)mc * a.ldx);                                       \
        ry[i] = make_uint4(ok ? ty.x : 0u, ok ? ty.y : 0u, ok ? ty.z : 0u, ok ? ty.w : 0u);                        \
        rx[i] = make_uint4(ok ? tx.x : 0u, ok ? tx.y : 0u, ok ? tx.z : 0u, ok ? tx.w : 0u);                        \
    }
#define TN_STORE_TILE(buf_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        lds_write128(smem[buf_][0], soff[i], ry[i]);                                                               \
        lds_write128(smem[buf_][1], soff[i], rx[i]);                                                               \
        if (do_bias) {                                                                                             \
            bs0 += __uint_as_float(ry[i].x << 16); bs1 += __uint_as_float(ry[i].x & 0xffff0000u);                  \
            bs2 += __uint_as_float(ry[i].y << 16); bs3 += __uint_as_float(ry[i].y & 0xffff0000u);                  \
            bs4 += __uint_as_float(ry[i].z << 16); bs5 += __uint_as_float(ry[i].z & 0xffff0000u);                  \
            bs6 += __uint_as_float(ry[i].w << 16); bs7 += __uint_as_float(ry[i].w & 0xffff0000u);                  \
        }                                                                                                          \
    }

    TN_LOAD_TILE(0)
    TN_STORE_TILE(0)
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read lane addressing: lane supplies (row = 8h + (li>>2) [+4], cols = base + 16*g1 + 4*(li&3))
    const int trow = 8 * h + (li >> 2);
    const int tcolA = wm * 64 + 16 * g1 + 4 * (li & 3);
    const int tcolB = wn * 64 + 16 * g1 + 4 * (li & 3);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { TN_LOAD_TILE(kt + 1) }
        const char* sY = smem[cur][0];
        const char* sX = smem[cur][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row0 = 16 * ks + trow;
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = join4(lds_tr_read(sY, lds128_off(row0, tcolA + 32 * i)), lds_tr_read(sY, lds128_off(row0 + 4, tcolA + 32 * i)));
                bf[i] = join4(lds_tr_read(sX, lds128_off(row0, tcolB + 32 * i)), lds_tr_read(sX, lds128_off(row0 + 4, tcolB + 32 * i)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) { TN_STORE_TILE(cur ^ 1) }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + wn * 64 + j * 32 + r32;
            if (q >= a.Q) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wm * 64 + i * 32 + acc_row(r, h);
                if (p < a.P) {
                    if (a.part) a.part[(size_t)split * a.part_stride + (size_t)p * a.Q + q] = acc[i][j][r];
                    else atomicAdd(a.dW + (size_t)p * a.lddw + q, acc[i][j][r]);
                }
            }
        }

    if (do_bias) {
        // reduce the 16 threads that share a column chunk (same tid&15) through LDS
        float* red = reinterpret_cast<float*>(smem[0][0]);  // all tile reads are behind the last barrier
        float* rp = red + (srow * 16 + sch) * 8;
        rp[0] = bs0; rp[1] = bs1; rp[2] = bs2; rp[3] = bs3; rp[4] = bs4; rp[5] = bs5; rp[6] = bs6; rp[7] = bs7;
        __syncthreads();
        if (tid < 128) {
            int ch = tid >> 3, e = tid & 7;
            float s = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) s += red[(rr * 16 + ch) * 8 + e];
            int p = p0 + ch * 8 + e;
            if (p < a.P) {
                if (a.part) a.part[(size_t)split * a.part_stride + (size_t)a.P * a.Q + p] = s;
                else atomicAdd(a.dbias + p, s);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn384: the same weight-gradient product with a 384 (P) x 128 (Q) output tile per workgroup: 96 FLOP per operand
// byte pulled through L2 -> LDS instead of 64 (the 128 x 128 kernel's loop is load-bound: 129 of 157 us with its MFMAs
// removed).  8 waves as 4 (P) x 2 (Q), each wave 96 x 64 = 3 x 2 MFMA tiles.  The reduction (token rows) streams in 32-row
// stages through a 4-deep LDS-DMA ring; a stage is four [32 rows][128 cols] images (Y columns 0-127 / 128-255 / 256-383,
// X columns 0-127), each 256-byte row swizzled per 64-byte segment (segment ^= row & 3) so that ds_read_b64_tr_b16 is
// conflict-free; every DMA instruction moves 4 rows x 256 contiguous bytes.  Used when P % 384 == 0 and Q % 128 == 0
// (all DiChaViT-S/B shapes); otherwise gemm_tn_kernel.
#ifndef DCV_T3_STAGES
#define DCV_T3_STAGES 4
#endif
constexpr int T3_BK = 32, T3_STAGES = DCV_T3_STAGES, T3_IMG = T3_BK * 256, T3_STAGE_BYTES = 4 * T3_IMG;  // 8 KB images, 32 KB stages

// bid: the workgroup's logical id inside this product, in [0, tiles * splits), tile index fastest
__device__ __forceinline__ void tn384_body(const GemmTnArgs& a, int bid, char* smem) {  // smem: T3_STAGES * T3_STAGE_BYTES = 128 KB: one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;
    const int h = lane >> 5, r32 = lane & 31, li = lane & 15, g1 = (lane >> 4) & 1;
    const int tiles_q = a.Q / 128, tiles = tiles_q * (a.P / 384);
    const int split = bid / tiles;
    bid -= split * tiles;
    const int tq = bid % tiles_q, tp = bid / tiles_q;
    const int p0 = tp * 384, q0 = tq * 128;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    if (m_begin >= m_end) return;
    const int nk = (m_end - m_begin + T3_BK - 1) / T3_BK;
    const int last_valid = (m_end - m_begin) - (nk - 1) * T3_BK;  // rows of the last stage that exist (1..32)

    // DMA: wave w fills image w>>1, rows [16*(w&1), +16) in four 4-row pieces; lane -> row lane>>4, physical chunk lane&15
    const int img = wave >> 1;
    const bf16_t* gsrc = (img < 3) ? a.Y + p0 + 128 * img : a.X + q0;
    const int ld = (img < 3) ? a.ldy : a.ldx;
    const int prow0 = 16 * (wave & 1) + (lane >> 4);  // + 4*j
    int lc8[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = prow0 + 4 * j, pc = lane & 15;
        lc8[j] = (((((pc >> 2) ^ (row & 3)) << 2) | (pc & 3))) * 8;  // logical chunk for this physical slot
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned dma_off = img * T3_IMG + 16 * (wave & 1) * 256;

#define T3_ISSUE(kt_)                                                                                     \
    {                                                                                                     \
        const unsigned sb_ = smem_base + ((kt_) % T3_STAGES) * T3_STAGE_BYTES + dma_off;                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
            const int m_ = min(m_begin + (kt_) * T3_BK + prow0 + 4 * j, m_end - 1); /* tail rows: zeroed in LDS below */ \
            glds16(gsrc + (size_t)m_ * ld + lc8[j], sb_ + j * 1024);                                      \
        }                                                                                                 \
    }

    for (int st = 0; st < T3_STAGES - 1; ++st)
        if (st < nk) T3_ISSUE(st)

    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read lane addressing inside an image: lane supplies (row = 8h + (li>>2) [+4], cols = base + 16*g1 + 4*(li&3))
    const int trow = 8 * h + (li >> 2);
    int offA[3], offB[2];  // byte offset of (image, column) for this wave's MFMA blocks, row 0
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 96 * wp + 32 * i + 16 * g1 + 4 * (li & 3);  // column inside the 384-wide Y tile
        offA[i] = (c >> 7) * T3_IMG + (c & 127);                  // image byte base + column (resolved per row below)
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = 3 * T3_IMG + (64 * wq + 32 * j + 16 * g1 + 4 * (li & 3));
    auto tr_off = [](int imgcol, int row) {  // imgcol = image byte base + column (column < 128)
        const int base = imgcol & ~127, col = imgcol & 127;
        return base + lds128_off(row, col);
    };

    // bias gradient: the workgroups add up their Y images (48 column chunks x 10 row groups) — every workgroup of a row of tiles for the
    // stages kt % tiles_q == tq, so that the tiles of a split keep the same pace.  With the tq == 0 tiles doing all of it (rounds 1-2) they fell
    // behind, the tiles of a split drifted apart and the shared operand rows left L2 before the slower tiles read them: 675 MB from HBM per
    // launch at P = 1536, Q = 384 against 425 MB now (386 MB = the operands once); the launch 175 -> 160 us (profiles/r03_x7_*)
    const bool do_bias = (a.dbias != nullptr);
    const int bch = tid % 48, brg = tid / 48;  // threads >= 480 idle for this
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;  // younger stages in flight: min(rem, T3_STAGES - 2), 4 DMA pieces per wave each
        if (T3_STAGES >= 5 && rem >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (T3_STAGES >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (T3_STAGES >= 3 && rem >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + T3_STAGES - 1 < nk) T3_ISSUE(kt + T3_STAGES - 1)  // (split between the SIMD partners as in gemm_nt_kernel, or issued behind the first fragment reads: +-1 %, not kept)
        char* st = smem + (kt % T3_STAGES) * T3_STAGE_BYTES;
        if (kt == nk - 1 && last_valid < T3_BK) {  // ragged end of the reduction: rows that do not exist must contribute 0
            const int nbad = T3_BK - last_valid;
            for (int idx = tid; idx < nbad * 64; idx += 512) {
                const int r = last_valid + idx / 64, im = (idx & 63) >> 4, ch = idx & 15;
                lds_write128(st, im * T3_IMG + r * 256 + ch * 16, make_uint4(0, 0, 0, 0));
            }
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row0 = 16 * ks + trow;
            bf16x8 af[3], bf[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) af[i] = join4(lds_tr_read(st, tr_off(offA[i], row0)), lds_tr_read(st, tr_off(offA[i], row0 + 4)));
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = join4(lds_tr_read(st, tr_off(offB[j], row0)), lds_tr_read(st, tr_off(offB[j], row0 + 4)));
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        if (do_bias && (kt % tiles_q) == tq && brg < 10) {
            const int im = bch >> 4, ci = bch & 15;
            for (int r = brg; r < T3_BK; r += 10) {
                const int pc = ((((ci >> 2) ^ (r & 3)) << 2) | (ci & 3));
                const uint4 v = lds_read128(st, im * T3_IMG + r * 256 + pc * 16);
                bs[0] += __uint_as_float(v.x << 16); bs[1] += __uint_as_float(v.x & 0xffff0000u);
                bs[2] += __uint_as_float(v.y << 16); bs[3] += __uint_as_float(v.y & 0xffff0000u);
                bs[4] += __uint_as_float(v.z << 16); bs[5] += __uint_as_float(v.z & 0xffff0000u);
                bs[6] += __uint_as_float(v.w << 16); bs[7] += __uint_as_float(v.w & 0xffff0000u);
            }
        }
    }
#undef T3_ISSUE
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + wq * 64 + j * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wp * 96 + i * 32 + acc_row(r, h);
                if (a.part) a.part[(size_t)split * a.part_stride + (size_t)p * a.Q + q] = acc[i][j][r];
                else atomicAdd(a.dW + (size_t)p * a.lddw + q, acc[i][j][r]);
            }
        }
    if (do_bias) {
        __syncthreads();  // all stage reads are done: reuse the ring for the 10 x 384 partial sums
        float* red = reinterpret_cast<float*>(smem);
        if (brg < 10) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(brg * 48 + bch) * 8 + e] = bs[e];
        }
        __syncthreads();
        if (tid < 384) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 10; ++g) sum += red[g * 384 + tid];
            if (a.part) a.part[a.bias_off + ((size_t)split * tiles_q + tq) * a.P + p0 + tid] = sum;  // one row per (split, tq)
            else atomicAdd(a.dbias + p0 + tid, sum);
        }
    }
}

#ifndef DCV_TN_WIDE_Q
#define DCV_TN_WIDE_Q 0  // variant builds: bit 0: 384 x 256 tiles for products with Q % 256 == 0; bit 1: for the transposed product (measured: +3 % / +12 % SLOWER, profiles/r05_x5_*); bit 2: 384 x 192 tiles for Q % 192 == 0
#endif
#if DCV_TN_WIDE_Q
// ------------------------------------------------------------------------------------------------
// gemm_tn384w (round 5): the same loop on a 384 (P) x 256 (Q) tile — 154 FLOP per operand byte pulled L2 -> LDS instead of 96.  Why: the 384 x 128 kernel
// waits for operand delivery (MFMA busy 0.39; TCP_PENDING_STALL 49 % of the launch; 28.9 M 128-byte requests at 457 cycles each = 59 in flight per CU
// by Little's law, 37 % of them compulsory HBM misses: profiles/r05_x5_*): the vector-memory path of a CU holds a fixed number of requests and every
// one takes an HBM latency, so the lever is requests per FLOP.  8 waves as 4 (P) x 2 (Q), each 96 x 128 = 3 x 4 MFMA tiles (192 accumulators, two waves
// per SIMD); a stage is five [32 rows][128 cols] images (40 KB), four stages = the whole LDS; five DMA pieces per wave and stage.
// SWAP: the launcher exchanged Y and X (a product whose Q is 384 and whose P is a multiple of 256, e.g. fc1's 1536 x 384): the tile is one of the
// transposed product, stored transposed (four consecutive p per lane: 16-byte stores at a row stride), and the bias gradient — the column sums of
// the caller's Y — comes from the X images.
constexpr int T3W_STAGES = 4, T3W_STAGE_BYTES = 5 * T3_IMG;  // 160 KB
// NJ = 3: a 384 x 192 tile on the same five images (waves 96 x 96 = 3 x 3 MFMA tiles, 144 accumulators; 128 FLOP per operand byte): every product of the model has
// Q % 192 == 0.  The fifth image then holds X columns 128-191 only; its DMA lanes whose slot lies in the unused half fetch the used half's lines again (no new request).
template <bool SWAP, int NJ = 4>
__device__ __forceinline__ void tn384w_body(const GemmTnArgs& a, int bid, char* smem) {
    static_assert(NJ == 4 || (NJ == 3 && !SWAP), "tile widths");
    constexpr int QT = 64 * NJ;  // tile width
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;
    const int h = lane >> 5, r32 = lane & 31, li = lane & 15, g1 = (lane >> 4) & 1;
    const int tiles_q = a.Q / QT, tiles = tiles_q * (a.P / 384);
    const int split = bid / tiles;
    bid -= split * tiles;
    const int tq = bid % tiles_q, tp = bid / tiles_q;
    const int p0 = tp * 384, q0 = tq * QT;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    if (m_begin >= m_end) return;
    const int nk = (m_end - m_begin + T3_BK - 1) / T3_BK;
    const int last_valid = (m_end - m_begin) - (nk - 1) * T3_BK;

    // DMA: piece p = 5 wave + j (j = 0..4) of the stage's 40: image p >> 3 (0-2: Y columns p0 + 128 image; 3, 4: X columns q0 + 128 (image - 3)), rows
    // 4 (p & 7) .. + 3; lane -> row lane >> 4 of the four, physical chunk lane & 15 (the row's swizzle depends on row & 3 = lane >> 4 only)
    const int lrow = lane >> 4, pc = lane & 15;
    const int lcol = ((((pc >> 2) ^ (lrow & 3)) << 2) | (pc & 3)) * 8;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const bf16_t* gsrc[5];
    int gld[5], grow[5];
    unsigned gdst[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int p = 5 * wave + j, img = p >> 3, rg = p & 7;
        gsrc[j] = (img < 3 ? a.Y + p0 + 128 * img : a.X + q0 + 128 * (img - 3)) + ((NJ == 3 && img == 4) ? (lcol & 63) : lcol);
        gld[j] = img < 3 ? a.ldy : a.ldx;
        grow[j] = 4 * rg + lrow;
        gdst[j] = img * T3_IMG + rg * 1024;
    }
#define T3W_ISSUE(kt_)                                                                                    \
    {                                                                                                     \
        const unsigned sb_ = smem_base + ((kt_) % T3W_STAGES) * T3W_STAGE_BYTES;                          \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                   \
            const int m_ = min(m_begin + (kt_) * T3_BK + grow[j], m_end - 1); /* tail rows: zeroed in LDS below */ \
            glds16(gsrc[j] + (size_t)m_ * gld[j], sb_ + gdst[j]);                                         \
        }                                                                                                 \
    }
    for (int st = 0; st < T3W_STAGES - 1; ++st)
        if (st < nk) T3W_ISSUE(st)

    f32x16 acc[3][NJ];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int trow = 8 * h + (li >> 2);
    int offA[3], offB[NJ];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 96 * wp + 32 * i + 16 * g1 + 4 * (li & 3);
        offA[i] = (c >> 7) * T3_IMG + (c & 127);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 32 * NJ * wq + 32 * j + 16 * g1 + 4 * (li & 3);
        offB[j] = (3 + (c >> 7)) * T3_IMG + (c & 127);
    }
    auto tr_off = [](int imgcol, int row) {
        const int base = imgcol & ~127, col = imgcol & 127;
        return base + lds128_off(row, col);
    };

    // bias gradient = column sums of the caller's Y: !SWAP: the three Y images (48 column chunks x 10 row groups), paced over the tiles of a row
    // as in tn384_body; SWAP: the two X images (32 chunks x 16 row groups), by the first tile row only (every tile row sees the same X columns)
    const bool do_bias = (a.dbias != nullptr) && (!SWAP || tp == 0);
    constexpr int BCH = SWAP ? 32 : 48, BRG = SWAP ? 16 : 10;
    const int bch = tid % BCH, brg = tid / BCH;
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;  // younger stages in flight: min(rem, 2), 5 DMA pieces per wave each
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + T3W_STAGES - 1 < nk) T3W_ISSUE(kt + T3W_STAGES - 1)
        char* st = smem + (kt % T3W_STAGES) * T3W_STAGE_BYTES;
        if (kt == nk - 1 && last_valid < T3_BK) {  // ragged end of the reduction: rows that do not exist must contribute 0
            const int nbad = T3_BK - last_valid;
            for (int idx = tid; idx < nbad * 80; idx += 512) {
                const int r = last_valid + idx / 80, im = (idx % 80) >> 4, ch = idx & 15;
                lds_write128(st, im * T3_IMG + r * 256 + ch * 16, make_uint4(0, 0, 0, 0));
            }
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row0 = 16 * ks + trow;
            bf16x8 af[3], bf[NJ];
#pragma unroll
            for (int i = 0; i < 3; ++i) af[i] = join4(lds_tr_read(st, tr_off(offA[i], row0)), lds_tr_read(st, tr_off(offA[i], row0 + 4)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = join4(lds_tr_read(st, tr_off(offB[j], row0)), lds_tr_read(st, tr_off(offB[j], row0 + 4)));
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        if (do_bias && (SWAP || (kt % tiles_q) == tq) && brg < BRG) {
            const int im = (SWAP ? 3 : 0) + (bch >> 4), ci = bch & 15;
            for (int r = brg; r < T3_BK; r += BRG) {
                const int pcx = ((((ci >> 2) ^ (r & 3)) << 2) | (ci & 3));
                const uint4 v = lds_read128(st, im * T3_IMG + r * 256 + pcx * 16);
                bs[0] += __uint_as_float(v.x << 16); bs[1] += __uint_as_float(v.x & 0xffff0000u);
                bs[2] += __uint_as_float(v.y << 16); bs[3] += __uint_as_float(v.y & 0xffff0000u);
                bs[4] += __uint_as_float(v.z << 16); bs[5] += __uint_as_float(v.z & 0xffff0000u);
                bs[6] += __uint_as_float(v.w << 16); bs[7] += __uint_as_float(v.w & 0xffff0000u);
            }
        }
    }
#undef T3W_ISSUE
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wq * 32 * NJ + j * 32 + r32;
            if constexpr (SWAP) {  // caller's layout: row q (its P index), column p (its Q index, a.P of them); registers 4g .. 4g + 3 = four consecutive p
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p = p0 + wp * 96 + i * 32 + 8 * g + 4 * h;
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (a.part) {
                        *reinterpret_cast<f32x4*>(a.part + (size_t)split * a.part_stride + (size_t)q * a.P + p) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(a.dW + (size_t)q * a.lddw + p + e, v[e]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = p0 + wp * 96 + i * 32 + acc_row(r, h);
                    if (a.part) a.part[(size_t)split * a.part_stride + (size_t)p * a.Q + q] = acc[i][j][r];
                    else atomicAdd(a.dW + (size_t)p * a.lddw + q, acc[i][j][r]);
                }
            }
        }
    if (a.dbias != nullptr) {  // (uniform over the workgroup)
        __syncthreads();  // all stage reads are done: reuse the ring for the partial sums
        float* red = reinterpret_cast<float*>(smem);
        if (do_bias && brg < BRG) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(brg * BCH + bch) * 8 + e] = bs[e];
        }
        __syncthreads();
        if (do_bias && tid < BCH * 8) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < BRG; ++g) sum += red[g * BCH * 8 + tid];
            if constexpr (SWAP) {  // bias index = the caller's P index = this product's Q index: one row of Q floats per split
                if (a.part) a.part[a.bias_off + (size_t)split * a.Q + q0 + tid] = sum;
                else atomicAdd(a.dbias + q0 + tid, sum);
            } else {
                if (a.part) a.part[a.bias_off + ((size_t)split * tiles_q + tq) * a.P + p0 + tid] = sum;
                else atomicAdd(a.dbias + p0 + tid, sum);
            }
        }
    }
}

#endif  // DCV_TN_WIDE_Q

__global__ __launch_bounds__(512) void gemm_tn384_kernel(GemmTnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[T3_STAGES * T3_STAGE_BYTES];  // 128 KB: one workgroup per CU
    tn384_body(a, xcd_remap(blockIdx.x, (a.Q / 128) * (a.P / 384) * a.splits), smem);
}

// Several weight-gradient products over the SAME token rows in one launch (a block's fc2 / fc1 / proj / qkv gradients: 36 tiles).  The CUs are
// filled by the tiles of all of them, so each tile is split 7 ways over the rows instead of 21 (or 85 for the 384 x 384 product): a third of
// the partial-tile traffic, k-loops three times as long, one launch and one fixed cost (~40 us of flush per launch, tools/tn_bench.py) instead of four.
constexpr int TN_GROUP_MAX = 8;
struct GemmTnGroup {
    int n;
    int start[TN_GROUP_MAX + 1];  // logical ids [start[i], start[i + 1]) belong to item i
    GemmTnArgs d[TN_GROUP_MAX];
};
__global__ __launch_bounds__(512) void gemm_tn384_group_kernel(GemmTnGroup g) {
#if DCV_TN_WIDE_Q
    __shared__ __attribute__((aligned(16))) char smem[T3W_STAGES * T3W_STAGE_BYTES];  // 160 KB (the 384 x 128 products use 128 KB of it)
    static_assert(T3W_STAGES * T3W_STAGE_BYTES >= T3_STAGES * T3_STAGE_BYTES, "ring sizes");
#else
    __shared__ __attribute__((aligned(16))) char smem[T3_STAGES * T3_STAGE_BYTES];  // 128 KB: one workgroup per CU
#endif
    const int id = xcd_remap(blockIdx.x, g.start[g.n]);
    int i = 0;
    while (i + 1 < g.n && id >= g.start[i + 1]) ++i;
    const GemmTnArgs a = g.d[i];
#if DCV_TN_WIDE_Q
    if constexpr ((DCV_TN_WIDE_Q & 1) != 0) { if (a.mode == 1) return tn384w_body<false>(a, id - g.start[i], smem); }
    if constexpr ((DCV_TN_WIDE_Q & 2) != 0) { if (a.mode == 2) return tn384w_body<true>(a, id - g.start[i], smem); }
    if constexpr ((DCV_TN_WIDE_Q & 4) != 0) { if (a.mode == 3) return tn384w_body<false, 3>(a, id - g.start[i], smem); }
#endif
    tn384_body(a, id - g.start[i], smem);
}

}  // namespace

// CU count of the current device, queried once per device (an immutable device property, cached; no allocation, no sync)
static int dcv_cu_count() {
    static std::atomic<int> cached[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

#define DCV_NT_CASES(KERNEL, G) DCV_NT_CASES_T(KERNEL, G, 512)
#define DCV_NT_CASES_T(KERNEL, G, THREADS)                                                             \
    switch (epilogue) {                                                                                \
        case DCV_EPI_BIAS_BF16:                                                                        \
            if (!bias) return DCV_ERR_NULL;                                                            \
            hipLaunchKernelGGL(KERNEL<DCV_EPI_BIAS_BF16>, dim3(G), dim3(THREADS), 0, s, a);                \
            break;                                                                                     \
        case DCV_EPI_BIAS_GELU_BF16:                                                                   \
            if (!bias || !out2) return DCV_ERR_NULL;                                                   \
            hipLaunchKernelGGL(KERNEL<DCV_EPI_BIAS_GELU_BF16>, dim3(G), dim3(THREADS), 0, s, a);           \
            break;                                                                                     \
        case DCV_EPI_BIAS_RESID_F32:                                                                   \
            if (!bias) return DCV_ERR_NULL;                                                            \
            hipLaunchKernelGGL(KERNEL<DCV_EPI_BIAS_RESID_F32>, dim3(G), dim3(THREADS), 0, s, a);           \
            break;                                                                                     \
        case DCV_EPI_PLAIN_BF16:                                                                       \
            hipLaunchKernelGGL(KERNEL<DCV_EPI_PLAIN_BF16>, dim3(G), dim3(THREADS), 0, s, a);               \
            break;                                                                                     \
        case DCV_EPI_GELU_BWD_BF16:                                                                    \
            if (!aux) return DCV_ERR_NULL;                                                             \
            hipLaunchKernelGGL(KERNEL<DCV_EPI_GELU_BWD_BF16>, dim3(G), dim3(THREADS), 0, s, a);            \
            break;

// which kernel dcv_gemm_nt_ex launches for this problem (DCV_TILE_NARROW / DCV_TILE_WIDE), or a negative error for an illegal forced tile
extern "C" int dcv_gemm_nt_pick(int M, int N, int K, int epilogue, int tile) {
    if (tile < DCV_TILE_AUTO || tile > DCV_TILE_ALT) return DCV_ERR_SHAPE;
    const bool legal384 = (N % N3_BN) == 0 && epilogue != DCV_EPI_PATCH;
    if (tile == DCV_TILE_WIDE) return legal384 ? DCV_TILE_WIDE : DCV_ERR_UNSUPPORTED;
    if (tile == DCV_TILE_ALT) return (DCV_NT_ALT && legal384 && epilogue != DCV_EPI_BIAS_RESID_F32) ? DCV_TILE_ALT : DCV_ERR_UNSUPPORTED;
    if (tile == DCV_TILE_NARROW) return DCV_TILE_NARROW;
    if (tile == DCV_TILE_PAIR) return epilogue != DCV_EPI_PATCH ? DCV_TILE_PAIR : DCV_ERR_UNSUPPORTED;
#ifndef DCV_WIDE_K_MIN
#define DCV_WIDE_K_MIN 1536
#endif
    if (!legal384 || epilogue == DCV_EPI_GELU_BWD_BF16 || !(N >= 1152 || K >= DCV_WIDE_K_MIN)) return DCV_TILE_NARROW;
    // Both kernels are persistent, so a launch costs rounds x time per tile, rounds = ceil(tiles / workgroups).  A 256 x 384 tile takes
    // 2.3 (K >= 1536: the k-loop dominates) to 2.5 (K = 384: the epilogue dominates) times a 256 x 128 tile (measured from the per-round
    // times at M = 12 369 ... 100 416, tools/gemm_mid_m.py, tools/gemm_bench.py): wide wins when its rounds x that factor stay below
    // the narrow kernel's rounds — always at the headline's M, seldom at a few rounds (M = 25 104, N = 1152: 297 wide tiles = 2 rounds
    // = 40 us against 891 narrow tiles = 4 rounds = 34 us).
    const int G = dcv_cu_count();
    const long tw = (long)((M + N3_BM - 1) / N3_BM) * (N / N3_BN), tn = 3 * tw;
    const long rw = (tw + G - 1) / G, rn = (tn + G - 1) / G;
    return (10 * rw * (K >= 1536 ? 23 : 25) < 100 * rn) ? DCV_TILE_WIDE : DCV_TILE_NARROW;
}

extern "C" int dcv_gemm_tn_pick(int M, int P, int Q, int tile) {
    (void)M;
    if (tile < DCV_TILE_AUTO || tile > DCV_TILE_WIDE) return DCV_ERR_SHAPE;
    const bool legal384 = (P % 384) == 0 && (Q % 128) == 0;
    if (tile == DCV_TILE_WIDE) return legal384 ? DCV_TILE_WIDE : DCV_ERR_UNSUPPORTED;
    if (tile == DCV_TILE_NARROW) return DCV_TILE_NARROW;
    return (legal384 && (P / 384) * (Q / 128) >= 6) ? DCV_TILE_WIDE : DCV_TILE_NARROW;
}

extern "C" int dcv_gemm_nt_ex(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int epilogue,
                              const float* bias, void* out, int ldo, void* out2, int ldo2, const void* aux, int ldaux,
                              const float* aux2, int T, int n, int grid_cap, int tile, void* stream) {
    if (!A || !W || !out) return DCV_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 8) != 0) return DCV_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8) || (ldo % 8) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)out & 15)) return DCV_ERR_ALIGN;
    if (grid_cap < 0 || tile < DCV_TILE_AUTO || tile > DCV_TILE_ALT) return DCV_ERR_SHAPE;
    if (epilogue == DCV_EPI_BIAS_RESID_F32 && aux2 && (T <= 0 || (M % T) != 0)) return DCV_ERR_SHAPE;  // per-sample branch scale: T rows per sample
    // persistent kernels: one workgroup per CU walks the tiles; grid_cap (> 0) lowers the number of workgroups — the data-parallel
    // backward leaves CUs to RCCL's kernels this way (dichavit.py), tests force multi-round walks on small problems
    const int cap = grid_cap > 0 ? grid_cap : dcv_cu_count();
    GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, bias, out, ldo, out2, ldo2, aux, ldaux, aux2, T, n};
    hipStream_t s = (hipStream_t)stream;
    // 256 x 384 tiles where they pay (measured, M = 100 416, against the 256 x 128 kernel): N = 1152 K = 384: 121 -> 107 us;
    // N = 1536 K = 384 + GELU: 238 -> 218; N = 384 K = 1536: 192 -> 184 (+residual), 154 -> 138 (plain); but N = 384 K = 384:
    // 96 -> 101 (393 tiles on 256 CUs: two rounds for 1.5 rounds of work), and the GELU-backward epilogue (N = 1536, HBM-heavy:
    // 616 MB in + out) 212 -> 220.  tile = DCV_TILE_WIDE forces the 256 x 384 kernel wherever it is legal, DCV_TILE_NARROW never uses it.
    const int pick = dcv_gemm_nt_pick(M, N, K, epilogue, tile);
    if (pick < 0) return pick;
    if (pick == DCV_TILE_WIDE) {
        int g3 = ((M + N3_BM - 1) / N3_BM) * (N / N3_BN);
        if (g3 > cap) g3 = cap;
        DCV_NT_CASES(gemm_nt384_kernel, g3)
            default:
                return DCV_ERR_UNSUPPORTED;
        }
        DCV_LAUNCH_CHECK();
        return DCV_OK;
    }
#if DCV_NT_ALT
    if (pick == DCV_TILE_ALT) {
        int ga = ((M + N3_BM - 1) / N3_BM) * (N / N3_BN);
        if (ga > cap) ga = cap;
        switch (epilogue) {
            case DCV_EPI_BIAS_BF16:
                if (!bias) return DCV_ERR_NULL;
                hipLaunchKernelGGL(gemm_nt_alt_kernel<DCV_EPI_BIAS_BF16>, dim3(ga), dim3(512), 0, s, a);
                break;
            case DCV_EPI_BIAS_GELU_BF16:
                if (!bias || !out2) return DCV_ERR_NULL;
                hipLaunchKernelGGL(gemm_nt_alt_kernel<DCV_EPI_BIAS_GELU_BF16>, dim3(ga), dim3(512), 0, s, a);
                break;
            case DCV_EPI_PLAIN_BF16:
                hipLaunchKernelGGL(gemm_nt_alt_kernel<DCV_EPI_PLAIN_BF16>, dim3(ga), dim3(512), 0, s, a);
                break;
            case DCV_EPI_GELU_BWD_BF16:
                if (!aux) return DCV_ERR_NULL;
                hipLaunchKernelGGL(gemm_nt_alt_kernel<DCV_EPI_GELU_BWD_BF16>, dim3(ga), dim3(512), 0, s, a);
                break;
            default:
                return DCV_ERR_UNSUPPORTED;
        }
        DCV_LAUNCH_CHECK();
        return DCV_OK;
    }
#endif
    if (pick == DCV_TILE_PAIR) {
        int gp = ((M + NP_BM - 1) / NP_BM) * ((N + NP_BN - 1) / NP_BN);
        if (gp > 2 * cap) gp = 2 * cap;  // two workgroups per CU
        DCV_NT_CASES_T(gemm_nt_pair_kernel, gp, 256)
            default:
                return DCV_ERR_UNSUPPORTED;
        }
        DCV_LAUNCH_CHECK();
        return DCV_OK;
    }
    int grid = ((M + NT_BM - 1) / NT_BM) * ((N + NT_BN - 1) / NT_BN);
    if (grid > cap) grid = cap;
    DCV_NT_CASES(gemm_nt_kernel, grid)
        case DCV_EPI_PATCH:
            if (!bias || !aux || !aux2 || T <= 0 || n <= 0 || (M % T) != 0 || (T % n) != 0) return DCV_ERR_SHAPE;
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_PATCH>, dim3(grid), dim3(512), 0, s, a);
            break;
        default:
            return DCV_ERR_UNSUPPORTED;
    }
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}
#undef DCV_NT_CASES
#undef DCV_NT_CASES_T

extern "C" int dcv_gemm_nt_resid_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, const float* resid,
                                    int ldr, const float* branch_scale, int T, float* x_out, int ldo, const float* gamma, const float* beta,
                                    float eps, void* u_out, int ldu, float* mean, float* rstd, int grid_cap, void* stream) {
    if (!A || !W || !bias || !resid || !x_out || !gamma || !beta || !u_out || !mean || !rstd) return DCV_ERR_NULL;
    if (M <= 0 || K <= 0 || (K % 64) != 0 || grid_cap < 0) return DCV_ERR_SHAPE;
    if (N != N3_BN) return DCV_ERR_UNSUPPORTED;  // the tile must span whole rows
    if ((lda % 8) || (ldw % 8) || (ldo % 4) || (ldr % 4) || (ldu % 8) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)x_out & 15) ||
        ((uintptr_t)resid & 15) || ((uintptr_t)u_out & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15))
        return DCV_ERR_ALIGN;
    if (branch_scale && (T <= 0 || (M % T) != 0)) return DCV_ERR_SHAPE;
    const int cap = grid_cap > 0 ? grid_cap : dcv_cu_count();
    GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, bias, x_out, ldo, u_out, ldu, resid, ldr, branch_scale, T > 0 ? T : 1, 0,
                 gamma, beta, mean, rstd, eps};
    int g3 = (M + N3_BM - 1) / N3_BM;
    if (g3 > cap) g3 = cap;
    hipLaunchKernelGGL(gemm_nt384_kernel<DCV_EPI_RESID_LN>, dim3(g3), dim3(512), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_gemm_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int epilogue,
                           const float* bias, void* out, int ldo, void* out2, int ldo2, const void* aux, int ldaux,
                           const float* aux2, int T, int n, void* stream) {
    return dcv_gemm_nt_ex(A, lda, W, ldw, M, N, K, epilogue, bias, out, ldo, out2, ldo2, aux, ldaux, aux2, T, n, 0, DCV_TILE_AUTO, stream);
}

// splits / rows per split of a weight-gradient launch (pure function of the problem, the tile and the device's CU count)
static void tn_plan(int M, int P, int Q, int pick, int cus, int& splits, int& mps) {
    if (pick == DCV_TILE_WIDE) {
        const int tiles3 = (P / 384) * (Q / 128);
        splits = cus / tiles3;  // one 128 KB workgroup per CU, one resident round
        const int max3 = (M + T3_BK - 1) / T3_BK;
        if (splits > max3) splits = max3;
        if (splits < 1) splits = 1;
        mps = ((M + splits - 1) / splits + T3_BK - 1) / T3_BK * T3_BK;
    } else {
        const int tiles = ((P + 127) / 128) * ((Q + 127) / 128);
        // one resident round: 2 workgroups per CU (64 KB LDS each) = 2 x CUs slots; one workgroup more would run
        // alone in a second round and double the launch time.  Every split is a multiple of BK rows.
        splits = 2 * cus / tiles;
        const int max_splits = (M + BK - 1) / BK;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        mps = ((M + splits - 1) / splits + BK - 1) / BK * BK;
    }
    splits = (M + mps - 1) / mps;  // the splits that have rows (the leading ones)
}

static int tn_launch(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw, float* dbias, int tile,
                     float* ws, long ws_floats, void* stream) {
    if (!Y || !X || !dW) return DCV_ERR_NULL;
    if (M <= 0 || P <= 0 || Q <= 0 || (P % 8) || (Q % 8)) return DCV_ERR_SHAPE;
    if ((ldy % 8) || (ldx % 8) || ((uintptr_t)Y & 15) || ((uintptr_t)X & 15)) return DCV_ERR_ALIGN;
    if (tile < DCV_TILE_AUTO || tile > DCV_TILE_WIDE) return DCV_ERR_SHAPE;
    const int cus = dcv_cu_count();
    // measured (M = 100 416): 1152x384 158 -> 140 us, 1536x384 206 -> 176, 384x1536 203 -> 182, but 384x384 63 -> 79
    // (3 tiles x 85 splits: the fp32 atomic traffic grows faster than the operand traffic shrinks)
    const int pick = dcv_gemm_tn_pick(M, P, Q, tile);
    if (pick < 0) return pick;
    int splits, mps;
    tn_plan(M, P, Q, pick, cus, splits, mps);
    // deterministic workspace: narrow kernel [splits][P*Q + P] (partial tile, then its bias partial); gemm_tn384, whose every tile
    // contributes to the bias gradient: [splits][P*Q], then [splits * tiles_q][P]
    const bool sep = pick != DCV_TILE_NARROW;
    const int tq9 = Q / 128;
    const long stride = sep ? (long)P * Q : (long)P * Q + P;
    const long bias_off = stride * splits;
    const long need = sep ? bias_off + (long)splits * tq9 * P : stride * splits;
    if (ws) {
        if (((uintptr_t)ws & 15) || (lddw % 4) || ((uintptr_t)dW & 15) || (dbias && ((uintptr_t)dbias & 15))) return DCV_ERR_ALIGN;
        if (ws_floats < need) return DCV_ERR_SHAPE;
    }
    GemmTnArgs a{(const bf16_t*)Y, ldy, (const bf16_t*)X, ldx, M, P, Q, dW, lddw, dbias, mps, splits, ws, stride, bias_off, 0};
    if (pick == DCV_TILE_WIDE) {
        hipLaunchKernelGGL(gemm_tn384_kernel, dim3((P / 384) * (Q / 128) * splits), dim3(512), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(((P + 127) / 128) * ((Q + 127) / 128) * splits), dim3(256), 0, (hipStream_t)stream, a);
    }
    DCV_LAUNCH_CHECK();
    if (ws) {
        if (sep) {
            if (!det_reduce(ws, splits, stride, dW, (long)P * Q, Q, lddw, nullptr, 0, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
            if (dbias && !det_reduce(ws + bias_off, splits * tq9, P, dbias, P, P, P, nullptr, 0, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
        } else if (!det_reduce(ws, splits, stride, dW, (long)P * Q, Q, lddw, dbias, dbias ? P : 0, (hipStream_t)stream)) {
            return DCV_ERR_LAUNCH;
        }
    }
    return DCV_OK;
}

extern "C" int dcv_gemm_tn_acc_ex(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw,
                                  float* dbias, int tile, void* stream) {
    return tn_launch(Y, ldy, X, ldx, M, P, Q, dW, lddw, dbias, tile, nullptr, 0, stream);
}

extern "C" long dcv_gemm_tn_det_ws_floats(int M, int P, int Q, int tile) {
    if (M <= 0 || P <= 0 || Q <= 0) return DCV_ERR_SHAPE;
    const int pick = dcv_gemm_tn_pick(M, P, Q, tile);
    if (pick < 0) return pick;
    int splits, mps;
    tn_plan(M, P, Q, pick, dcv_cu_count(), splits, mps);
    if (pick == DCV_TILE_NARROW) return ((long)P * Q + P) * splits;
    return (long)splits * P * Q + (long)splits * (Q / 128) * P;
}

extern "C" int dcv_gemm_tn_acc_det(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw,
                                   float* dbias, int tile, float* ws, long ws_floats, void* stream) {
    if (!ws) return DCV_ERR_NULL;
    return tn_launch(Y, ldy, X, ldx, M, P, Q, dW, lddw, dbias, tile, ws, ws_floats, stream);
}

// ---- grouped form (include/dcv.h: dcv_tn_item) --------------------------------------------------------------------------------------------
// Per item: the tile form (0: 384 x 128; 1: 384 x 256; 2: 384 x 256 of the transposed product), its splits over the token rows and rows per split.  A
// 384 x 256 tile is two units of work, so its products get twice the splits of the 384 x 128 ones: every workgroup of the launch does the same FLOPs.
struct TnGroupPlan {
    int mode[TN_GROUP_MAX], splits[TN_GROUP_MAX], mps[TN_GROUP_MAX], tiles[TN_GROUP_MAX];
    long off[TN_GROUP_MAX];
    long need;
};
static int tn_group_plan(const dcv_tn_item* it, int n, int M, int cus, TnGroupPlan& pl) {
    if (!it) return DCV_ERR_NULL;
    if (n < 1 || n > TN_GROUP_MAX || M <= 0) return DCV_ERR_SHAPE;
    int units = 0;  // in half tiles of 384 x 128
    static const int weight[4] = {2, 4, 4, 3}, tile_q[4] = {128, 256, 0, 192};
    for (int i = 0; i < n; ++i) {
        if (!it[i].Y || !it[i].X || !it[i].dW) return DCV_ERR_NULL;
        if (it[i].P <= 0 || it[i].Q <= 0 || (it[i].P % 384) || (it[i].Q % 128)) return DCV_ERR_UNSUPPORTED;
        if ((it[i].ldy % 8) || (it[i].ldx % 8) || ((uintptr_t)it[i].Y & 15) || ((uintptr_t)it[i].X & 15)) return DCV_ERR_ALIGN;
        const int P = it[i].P, Q = it[i].Q;
        if ((DCV_TN_WIDE_Q & 4) && (Q % 192) == 0) {
            pl.mode[i] = 3;
            pl.tiles[i] = (P / 384) * (Q / 192);
        } else if ((DCV_TN_WIDE_Q & 1) && (Q % 256) == 0) {
            pl.mode[i] = 1;
            pl.tiles[i] = (P / 384) * (Q / 256);
        } else if ((DCV_TN_WIDE_Q & 2) && (Q % 384) == 0 && (P % 256) == 0) {
            pl.mode[i] = 2;
            pl.tiles[i] = (Q / 384) * (P / 256);
        } else {
            pl.mode[i] = 0;
            pl.tiles[i] = (P / 384) * (Q / 128);
        }
        units += pl.tiles[i] * weight[pl.mode[i]];
    }
    if (units > 2 * cus) return DCV_ERR_UNSUPPORTED;  // more than one resident round: call the products one by one
    const int base2 = 2 * cus / units;  // splits of a 384 x 128 product
    const int max3 = (M + T3_BK - 1) / T3_BK;
    pl.need = 0;
    for (int i = 0; i < n; ++i) {
        int sp = base2 * weight[pl.mode[i]] / 2;
        if (sp < 1) sp = 1;
        if (sp > max3) sp = max3;
        const int mps = ((M + sp - 1) / sp + T3_BK - 1) / T3_BK * T3_BK;
        pl.mps[i] = mps;
        pl.splits[i] = (M + mps - 1) / mps;
        // per item: [splits][P * Q] partial tiles (the caller's layout), then the bias partials: [splits * tiles_q][P], or [splits][P] for the transposed form
        pl.off[i] = pl.need;
        const int bias_rows = pl.mode[i] == 2 ? pl.splits[i] : pl.splits[i] * (it[i].Q / tile_q[pl.mode[i]]);
        pl.need += (long)pl.splits[i] * it[i].P * it[i].Q + (long)bias_rows * it[i].P;
    }
    return DCV_OK;
}

extern "C" long dcv_gemm_tn_group_ws_floats(const dcv_tn_item* items, int n, int M) {
    TnGroupPlan pl;
    const int rc = tn_group_plan(items, n, M, dcv_cu_count(), pl);
    return rc ? rc : pl.need;
}

extern "C" int dcv_gemm_tn_group(const dcv_tn_item* items, int n, int M, float* ws, long ws_floats, void* stream) {
    TnGroupPlan pl;
    const int rc = tn_group_plan(items, n, M, dcv_cu_count(), pl);
    if (rc) return rc;
    if (ws) {
        if ((uintptr_t)ws & 15) return DCV_ERR_ALIGN;
        if (ws_floats < pl.need) return DCV_ERR_SHAPE;
    }
    for (int i = 0; i < n; ++i)
        if ((items[i].lddw % 4) || ((uintptr_t)items[i].dW & 15) || (items[i].dbias && ((uintptr_t)items[i].dbias & 15))) {
            if (ws || pl.mode[i] == 2) return DCV_ERR_ALIGN;
        }
    GemmTnGroup g;
    g.n = n;
    int id = 0;
    for (int i = 0; i < n; ++i) {
        const dcv_tn_item& t = items[i];
        g.start[i] = id;
        id += pl.tiles[i] * pl.splits[i];
        const long stride = (long)t.P * t.Q;
        if (pl.mode[i] == 2)  // the transposed product: operands exchanged; outputs keep the caller's layout
            g.d[i] = GemmTnArgs{(const bf16_t*)t.X, t.ldx, (const bf16_t*)t.Y, t.ldy, M, t.Q, t.P, t.dW, t.lddw, t.dbias, pl.mps[i], pl.splits[i],
                                ws ? ws + pl.off[i] : nullptr, stride, stride * pl.splits[i], 2};
        else
            g.d[i] = GemmTnArgs{(const bf16_t*)t.Y, t.ldy, (const bf16_t*)t.X, t.ldx, M, t.P, t.Q, t.dW, t.lddw, t.dbias, pl.mps[i], pl.splits[i],
                                ws ? ws + pl.off[i] : nullptr, stride, stride * pl.splits[i], pl.mode[i]};
    }
    g.start[n] = id;
    for (int i = n + 1; i <= TN_GROUP_MAX; ++i) g.start[i] = id;
    hipLaunchKernelGGL(gemm_tn384_group_kernel, dim3(id), dim3(512), 0, (hipStream_t)stream, g);
    DCV_LAUNCH_CHECK();
    if (ws) {  // the fixed-order second pass of every product (weight, bias) in one launch
        DetJobs jb;
        jb.n = 0;
        jb.wg_start[0] = 0;
        for (int i = 0; i < n; ++i) {
            const dcv_tn_item& t = items[i];
            const long stride = (long)t.P * t.Q;
            float* w = ws + pl.off[i];
            if (!det_jobs_add(jb, w, pl.splits[i], stride, t.dW, stride, t.Q, t.lddw)) return DCV_ERR_ALIGN;
            const int bias_rows = pl.mode[i] == 2 ? pl.splits[i] : pl.splits[i] * (t.Q / (pl.mode[i] == 1 ? 256 : pl.mode[i] == 3 ? 192 : 128));
            if (t.dbias && !det_jobs_add(jb, w + stride * pl.splits[i], bias_rows, t.P, t.dbias, t.P, t.P, t.P)) return DCV_ERR_ALIGN;
        }
        if (!det_reduce_multi(jb, (hipStream_t)stream)) return DCV_ERR_LAUNCH;
    }
    return DCV_OK;
}

extern "C" int dcv_gemm_tn_acc(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw,
                               float* dbias, void* stream) {
    return dcv_gemm_tn_acc_ex(Y, ldy, X, ldx, M, P, Q, dW, lddw, dbias, DCV_TILE_AUTO, stream);
}

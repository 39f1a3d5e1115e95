// The channel-embedding proxy regulariser in one launch (forward value AND both gradients).
// Replaces, for the square case with identity targets,
//     proxy_loss(channel_emb_proxies[cur_channels], channel_embed, eye(C), scale)      (models/dichavit.py:399-402, models/loss_fn.py:7-21)
//   = cross_entropy(-cdist(scale * normalize(e), scale * normalize(p))^2, eye(C))
// which the host module ran as ~40 launches of a few microseconds on [C, D] tensors, forward and backward, between the encoder's last forward
// and first backward kernel: 0.43 ms of a 34.5 ms step with nothing else running (tools/glue_cost.py).
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

constexpr int PL_MAX_FLOATS = 16384;  // C * D per operand: two normalised copies in LDS (128 KB)
constexpr int PL_MAX_C = 32;

// one workgroup of 1024 threads (16 waves: at C = 8 every row / every four logits have a wave of their own; with 4 waves the launch took 45 us).  e = embeddings [C, D] (rows of the logits), p = proxies [C, D] (columns), both fp32 contiguous.
__global__ __launch_bounds__(1024) void proxy_loss_kernel(const float* __restrict__ e, const float* __restrict__ p, int C, int D, float scale,
                                                         float* __restrict__ loss, float* __restrict__ de, float* __restrict__ dp) {
    extern __shared__ float sm[];
    float* en = sm;                   // [C][D] normalised embeddings
    float* pn = sm + (size_t)C * D;   // [C][D] normalised proxies
    __shared__ float nrm[2][PL_MAX_C];          // max(||row||, 1e-12)
    __shared__ float L[PL_MAX_C][PL_MAX_C + 1];  // logits, then d(loss)/d(logits)
    __shared__ float li[PL_MAX_C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // 1. F.normalize(x, p=2, dim=-1): x / max(||x||, 1e-12)
    for (int r = wave; r < 2 * C; r += 16) {
        const float* src = (r < C) ? e + (size_t)r * D : p + (size_t)(r - C) * D;
        float* dst = (r < C) ? en + (size_t)r * D : pn + (size_t)(r - C) * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += src[d] * src[d];
        const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
        for (int d = lane; d < D; d += 64) dst[d] = src[d] / n;
        if (lane == 0) nrm[r < C ? 0 : 1][r < C ? r : r - C] = n;
    }
    __syncthreads();
    // 2. logits L[i][j] = -sum_d (scale * en_i[d] - scale * pn_j[d])^2
    for (int ij = wave; ij < C * C; ij += 16) {
        const int i = ij / C, j = ij % C;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float t = scale * en[(size_t)i * D + d] - scale * pn[(size_t)j * D + d];
            s += t * t;
        }
        s = wave_sum(s);
        if (lane == 0) L[i][j] = -s;
    }
    __syncthreads();
    // 3. cross entropy with identity targets, mean over rows; L becomes d(loss)/d(logits) = (softmax - I) / C
    if (tid < C) {
        const int i = tid;
        float m = -INFINITY;
        for (int j = 0; j < C; ++j) m = fmaxf(m, L[i][j]);
        float z = 0.f;
        for (int j = 0; j < C; ++j) z += expf(L[i][j] - m);
        const float lse = m + logf(z);
        li[i] = lse - L[i][i];
        for (int j = 0; j < C; ++j) L[i][j] = (expf(L[i][j] - lse) - (i == j ? 1.f : 0.f)) / C;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < C; ++i) s += li[i];
        loss[0] = s / C;
    }
    // 4. gradients.  With a_i = scale * en_i, b_j = scale * pn_j: dL_ij/da_i = -2 (a_i - b_j), dL_ij/db_j = +2 (a_i - b_j); then through
    //    the normalisation: dx = (dn - n_hat (n_hat . dn)) / max(||x||, eps)
    for (int r = wave; r < 2 * C; r += 16) {
        const bool is_e = r < C;
        const int i = is_e ? r : r - C;
        const float* self = (is_e ? en : pn) + (size_t)i * D;
        const float* other = is_e ? pn : en;
        float dn[16];  // D <= 1024
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = lane + 64 * k;
            dn[k] = 0.f;
            if (d < D) {
                float acc = 0.f;
                for (int j = 0; j < C; ++j) {
                    const float w = is_e ? L[i][j] : L[j][i];
                    const float diff = is_e ? (scale * self[d] - scale * other[(size_t)j * D + d]) : (scale * other[(size_t)j * D + d] - scale * self[d]);
                    acc += w * (is_e ? -2.f : 2.f) * diff;
                }
                dn[k] = scale * acc;
                dot += self[d] * dn[k];
            }
        }
        dot = wave_sum(dot);
        const float n = nrm[is_e ? 0 : 1][i];
        float* out = (is_e ? de : dp) + (size_t)i * D;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = lane + 64 * k;
            if (d < D) out[d] = (dn[k] - self[d] * dot) / n;
        }
    }
}

}  // namespace

extern "C" int dcv_proxy_loss_supported(int C, int D) { return (C >= 1 && C <= PL_MAX_C && D >= 1 && D <= 1024 && (long)C * D <= PL_MAX_FLOATS) ? 1 : 0; }

extern "C" int dcv_proxy_loss(const float* emb, const float* proxies, int C, int D, float scale, float* loss, float* d_emb, float* d_proxies,
                              void* stream) {
    if (!emb || !proxies || !loss || !d_emb || !d_proxies) return DCV_ERR_NULL;
    if (C <= 0 || D <= 0) return DCV_ERR_SHAPE;
    if (!dcv_proxy_loss_supported(C, D)) return DCV_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * C * D * sizeof(float);
    if (lds > 64 * 1024 &&  // more than 64 KB of dynamic LDS needs the attribute (idempotent; set per call: the library keeps no state)
        hipFuncSetAttribute((const void*)proxy_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PL_MAX_FLOATS * (int)sizeof(float)) != hipSuccess)
        return DCV_ERR_LAUNCH;
    hipLaunchKernelGGL(proxy_loss_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, emb, proxies, C, D, scale, loss, d_emb, d_proxies);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

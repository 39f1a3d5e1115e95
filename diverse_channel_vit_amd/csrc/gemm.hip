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
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16 tiles.
// Register-staged double-buffered LDS (global loads for tile t+1 are issued before the MFMAs of
// tile t and written to LDS after them: one barrier per K-tile).
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

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
};

__device__ __forceinline__ float gelu_exact(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float z) {
    float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmNtArgs& a, int m, int n, float v) {
    if constexpr (EPI == DCV_EPI_BIAS_BF16) {
        ((bf16_t*)a.out)[(size_t)m * a.ldo + n] = (bf16_t)(v + a.bias[n]);
    } else if constexpr (EPI == DCV_EPI_BIAS_GELU_BF16) {
        float z = v + a.bias[n];
        bf16_t zb = (bf16_t)z;
        ((bf16_t*)a.out)[(size_t)m * a.ldo + n] = zb;                       // pre-activation (saved for backward)
        ((bf16_t*)a.out2)[(size_t)m * a.ldo2 + n] = (bf16_t)gelu_exact((float)zb);  // activation
    } else if constexpr (EPI == DCV_EPI_BIAS_RESID_F32) {
        // out = residual + acc + bias; the residual is read from aux when given (out-of-place keeps the
        // layer input alive for the LayerNorm backward at no extra traffic), else updated in place
        const float* rsd = a.aux ? (const float*)a.aux + (size_t)m * a.ldaux + n : (const float*)a.out + (size_t)m * a.ldo + n;
        ((float*)a.out)[(size_t)m * a.ldo + n] = *rsd + v + a.bias[n];
    } else if constexpr (EPI == DCV_EPI_PLAIN_BF16) {
        ((bf16_t*)a.out)[(size_t)m * a.ldo + n] = (bf16_t)v;
    } else if constexpr (EPI == DCV_EPI_GELU_BWD_BF16) {
        float z = (float)((const bf16_t*)a.aux)[(size_t)m * a.ldaux + n];
        ((bf16_t*)a.out)[(size_t)m * a.ldo + n] = (bf16_t)(v * gelu_grad(z));
    } else if constexpr (EPI == DCV_EPI_PATCH) {
        // row m = b*T + t ; token t = c*n + i  ->  x[b, 1+t, :] = conv + bias + E[c] + pos[1+i]
        int b = m / a.T, t = m - b * a.T;
        int c = t / a.n, i = t - c * a.n;
        float y = v + a.bias[n];
        if (a.out2) ((float*)a.out2)[(size_t)m * a.ldo2 + n] = y;  // pre-embedding tokens (ortho loss input)
        float e = ((const float*)a.aux)[(size_t)c * a.ldaux + n];
        float p = a.aux2[(size_t)(1 + i) * a.ldaux + n];
        ((float*)a.out)[((size_t)b * (a.T + 1) + 1 + t) * a.ldo + n] = y + e + p;
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNtArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2][2][BM * BK * 2];  // [buf][A|W][16 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, r32 = lane & 31;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging assignment: 4 chunks of 16 B per operand per thread
    const bf16_t* gA[4];
    const bf16_t* gW[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int q = tid + 256 * i, row = q >> 3, ch = q & 7;
        int ra = min(m0 + row, a.M - 1), rw = min(n0 + row, a.N - 1);
        gA[i] = a.A + (size_t)ra * a.lda + ch * 8;
        gW[i] = a.W + (size_t)rw * a.ldw + ch * 8;
        soff[i] = row * 128 + ((ch ^ swz64(row)) << 4);
    }
    uint4 ra[4], rw[4];
    const int nk = a.K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(gA[i]);
        rw[i] = *reinterpret_cast<const uint4*>(gW[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lds_write128(smem[0][0], soff[i], ra[i]);
        lds_write128(smem[0][1], soff[i], rw[i]);
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = swz64(r32);
    const int rowA = (wm * 64 + r32) * 128, rowW = (wn * 64 + r32) * 128;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const uint4*>(gA[i] + (kt + 1) * BK);
                rw[i] = *reinterpret_cast<const uint4*>(gW[i] + (kt + 1) * BK);
            }
        }
        const char* sA = smem[cur][0];
        const char* sW = smem[cur][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = ((2 * ks + h) ^ sw) << 4;
            bf16x8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = as_bf16x8(lds_read128(sA, rowA + i * 32 * 128 + co));
                wf[i] = as_bf16x8(lds_read128(sW, rowW + i * 32 * 128 + co));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], wf[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds_write128(smem[cur ^ 1][0], soff[i], ra[i]);
                lds_write128(smem[cur ^ 1][1], soff[i], rw[i]);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r32;
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + acc_row(r, h);
                if (m < a.M) epilogue_store<EPI>(a, m, n, acc[i][j][r]);
            }
        }
}

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
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2][2][BK * 128 * 2];  // [buf][Y|X][64 rows x 256 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, r32 = lane & 31, li = lane & 15, g1 = (lane >> 4) & 1;
    const int tiles_q = (a.Q + 127) / 128;
    int bid = blockIdx.x;
    const int split = bid % a.splits;
    bid /= a.splits;
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
                if (p < a.P) atomicAdd(a.dW + (size_t)p * a.lddw + q, acc[i][j][r]);
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
            if (p < a.P) atomicAdd(a.dbias + p, s);
        }
    }
}

}  // namespace

extern "C" int dcv_gemm_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int epilogue,
                           const float* bias, void* out, int ldo, void* out2, int ldo2, const void* aux, int ldaux,
                           const float* aux2, int T, int n, void* stream) {
    if (!A || !W || !out) return DCV_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) return DCV_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return DCV_ERR_ALIGN;
    GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, bias, out, ldo, out2, ldo2, aux, ldaux, aux2, T, n};
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case DCV_EPI_BIAS_BF16:
            if (!bias) return DCV_ERR_NULL;
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_BIAS_BF16>, dim3(grid), dim3(256), 0, s, a);
            break;
        case DCV_EPI_BIAS_GELU_BF16:
            if (!bias || !out2) return DCV_ERR_NULL;
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_BIAS_GELU_BF16>, dim3(grid), dim3(256), 0, s, a);
            break;
        case DCV_EPI_BIAS_RESID_F32:
            if (!bias) return DCV_ERR_NULL;
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_BIAS_RESID_F32>, dim3(grid), dim3(256), 0, s, a);
            break;
        case DCV_EPI_PLAIN_BF16:
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_PLAIN_BF16>, dim3(grid), dim3(256), 0, s, a);
            break;
        case DCV_EPI_GELU_BWD_BF16:
            if (!aux) return DCV_ERR_NULL;
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_GELU_BWD_BF16>, dim3(grid), dim3(256), 0, s, a);
            break;
        case DCV_EPI_PATCH:
            if (!bias || !aux || !aux2 || T <= 0 || n <= 0 || (M % T) != 0 || (T % n) != 0) return DCV_ERR_SHAPE;
            hipLaunchKernelGGL(gemm_nt_kernel<DCV_EPI_PATCH>, dim3(grid), dim3(256), 0, s, a);
            break;
        default:
            return DCV_ERR_UNSUPPORTED;
    }
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_gemm_tn_acc(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw,
                               float* dbias, void* stream) {
    if (!Y || !X || !dW) return DCV_ERR_NULL;
    if (M <= 0 || P <= 0 || Q <= 0 || (P % 8) || (Q % 8)) return DCV_ERR_SHAPE;
    if ((ldy % 8) || (ldx % 8) || ((uintptr_t)Y & 15) || ((uintptr_t)X & 15)) return DCV_ERR_ALIGN;
    const int tiles = ((P + 127) / 128) * ((Q + 127) / 128);
    // ~2 workgroups per CU; every split a multiple of BK rows
    int splits = (512 + tiles - 1) / tiles;
    int max_splits = (M + BK - 1) / BK;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mps = ((M + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (M + mps - 1) / mps;
    GemmTnArgs a{(const bf16_t*)Y, ldy, (const bf16_t*)X, ldx, M, P, Q, dW, lddw, dbias, mps, splits};
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// Flash-style multi-head self-attention for head_dim 64 on gfx950 (bf16 MFMA 32x32x16, fp32 softmax).
// Replaces Attention.forward's materialised [B,H,N,N] softmax (models/vit.py:121-144):
//     qkv [B,N,3,H,64] bf16  ->  O [B,N,H*64] bf16,  LSE [B,H,N] f32 (natural log)
// and its backward (dQ, dK, dV recomputed from LSE; no N x N tensor ever touches HBM).
//
// Orientation (all three kernels): scores are produced TRANSPOSED, S^T[key,q] = K . Q^T, so the
// query index sits on the MFMA lane (column) and the softmax statistics m, l, LSE, delta are plain
// per-lane scalars.  The f32 score tile is converted in registers to the bf16 B operand of the next
// MFMA, which sums over the tile's ROW index (keys):  O^T[d,q] += V^T[d,key] . P^T[key,q].
// V^T / K^T / Q^T / dO^T operands are read from row-major LDS tiles with ds_read_b64_tr_b16.
//
//   attn_fwd      : WG = 4 waves x 32 queries, loops over 64-key tiles.            8 MFMA / 32x32 tile pair
//   attn_bwd_dq   : same loop; S^T, dP^T = V . dO^T, dQ^T += K^T . dS^T.
//   attn_bwd_dkdv : WG = 4 waves x 32 keys, loops over 64-query tiles; S = Q . K^T (key on the lane),
//                   dP = dO . V^T, dV^T += dO^T . P, dK^T += Q^T . dS.   No atomics anywhere: results
//                   are bitwise reproducible.
#include "attn_common.hpp"

#ifndef DCV_FWD_LSUM_MFMA
#define DCV_FWD_LSUM_MFMA 0  // 1 (variant builds): measured +1 % SLOWER (profiles/r05_x7_*)
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// Forward: one key tile at a time on a 3-stage K/V ring (48 KB -> 3 workgroups per CU; 136 VGPRs -> 3 waves per SIMD).
// Two earlier forms measured slower at the headline shape (tools/ab_bench.py): register-staged K/V with per-use address
// arithmetic 436-448 us; a 4-stage ring with the NEXT tile's score MFMAs issued ahead of the softmax (register ping-pong,
// 228 VGPRs, 2 waves per SIMD) 418-430 us; this one 397-405 us.  Occupancy beat intra-wave overlap.
// PS ("pre-scaled q"): the q part of qkv already holds q * scale * log2(e) (the model scales the bf16 operand copy of W_q and the bias before the
// qkv GEMM: one rounding, as before).  The score accumulators then START at -m (the lane's reference maximum, in log2 units) instead of zero, and
// p = exp2(accumulator) needs no multiply-subtract: one vector instruction less per score (cdna guide, appendix B: row constants as the initial
// accumulator).  minit holds -m in all 16 registers of an accumulator tuple and changes only when the reference maximum does.
template <bool PS>
#if DCV_WPE_FWD
DCV_WAVES_PER_SIMD(DCV_WPE_FWD)
#endif
__global__ __launch_bounds__(256) void attn_fwd3_kernel(AttnArgs a) {
    constexpr int ST = 3;
    __shared__ __attribute__((aligned(16))) char sKV[ST * KV_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = (a.Nq + 127) / 128;
    const int BH = a.B * a.H;
    int bh, qt;
    attn_block_to_tile(blockIdx.x, BH, nqt, a.Nq, bh, qt);
    const int b = bh / a.H, hh = bh % a.H;
    const int D = a.H * 64;
    const size_t rs = (size_t)3 * D;
    const bf16_t* Qb = a.qkv + (size_t)b * a.N * rs + hh * 64;
    const int nt = (a.N + 63) / 64;

    const int rowl = 16 * wave + (lane >> 3);
    const int lc8[2] = {((lane & 7) ^ swz64(rowl)) * 8, ((lane & 7) ^ swz64(rowl + 8)) * 8};
    const unsigned voff0 = (unsigned)(((size_t)rowl * rs + lc8[0]) * 2), voff1 = (unsigned)(((size_t)(rowl + 8) * rs + lc8[1]) * 2);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(sKV)) + 16 * wave * 128;
    const bf16_t* const kbase = Qb + D;
    auto kv_issue = [&](int t, int slot) {
        const unsigned sb = smem_base + slot * KV_STAGE_BYTES;
        const bf16_t* kt = kbase + (size_t)t * 64 * rs;
        const bf16_t* vt = kt + D;
        unsigned o0 = voff0, o1 = voff1;
        if (t * 64 + 64 > a.N) {
            o0 = (unsigned)(((size_t)(min(t * 64 + rowl, a.N - 1) - t * 64) * rs + lc8[0]) * 2);
            o1 = (unsigned)(((size_t)(min(t * 64 + rowl + 8, a.N - 1) - t * 64) * rs + lc8[1]) * 2);
        }
        glds16s(kt, o0, sb);
        glds16s(vt, o0, sb + 8192);
        glds16s(kt, o1, sb + 1024);
        glds16s(vt, o1, sb + 8192 + 1024);
    };

    const int q = qt * 128 + wave * 32 + r32;
    const int qc = min(q, a.N - 1);
    const bool active = qt * 128 + wave * 32 < a.Nq;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Qb + (size_t)qc * rs + 16 * ks + 8 * h));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));
    for (int st = 0; st < ST - 1; ++st)
        if (st < nt) kv_issue(st, st);
    const LaneOffs lo = lane_offs(lane);

    f32x16 o[2];
    zero_acc(o[0]);
    zero_acc(o[1]);
    float m = PS ? 0.f : -INFINITY, l = 0.f;  // PS: reference maximum in log2 units of the pre-scaled scores, set by the first tile
    const float c = a.scale * LOG2E;
    constexpr float FWD_RESCALE_LOG2 = 8.f;
    f32x16 minit;
    zero_acc(minit);
    // DCV_FWD_LSUM_MFMA (pre-scaled-q form): the softmax denominator from the matrix pipe — one more MFMA per 32 x 16 block of P with an all-ones A operand, whose
    // every output row is the column sum of P — instead of 32 v_add per tile and wave; the sum is then over the bf16 P the numerator uses
    f32x16 lacc;
    zero_acc(lacc);
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    // COMPUTE = false: a wave without a valid query row only keeps the ring going; separate loops, not a branch in the loop
    // (attn_bwd.hip: the branch made the accumulators loop-carried phis resolved with register copies)
    auto tile = [&](auto MASKED, auto COMPUTE, int t, int slot) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KV_DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) kv_issue(t + 2, slot == 0 ? 2 : slot - 1);
        if constexpr (!decltype(COMPUTE)::value) return;
        const int so = slot * KV_STAGE_BYTES;
        int ro[4], co[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ro[ks] = lo.rows[ks] + so;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            co[dt][0] = lo.cols[dt][0] + so + 8192;
            co[dt][1] = lo.cols[dt][1] + so + 8192;
        }
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (PS) s[kb] = minit;
            else zero_acc(s[kb]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[kb] = mfma32(as_bf16x8(lds_read128(sKV, ro[ks] + kb * 4096)), qf[ks], s[kb]);
        }
        if constexpr (decltype(MASKED)::value) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + 32 * kb + acc_row(r, h) >= a.N) s[kb][r] = -INFINITY;
        }
        float rsum = 0.f;
        if constexpr (PS) {
            // the accumulators are relative to the reference maximum already: mx = this tile's excess over it
            float mx = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = max_halves(mx);
            // first tile: the reference becomes the tile's maximum whatever its sign (the accumulators started at 0); later tiles: lazily, as below
            if (t == 0 || __any(mx > FWD_RESCALE_LOG2)) {
                const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
                // t = 0: l and O are still zero, and exp2(-d) is +inf when every score of the first tile lies below -127 (log2 units): 0 * inf = NaN
                const float alpha = (t == 0) ? 0.f : __builtin_amdgcn_exp2f(-d);
                l *= alpha;
                if (DCV_FWD_LSUM_MFMA) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                m += d;
#pragma unroll
                for (int r = 0; r < 16; ++r) minit[r] = -m;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] -= d;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][r]);
                    s[kb][r] = p;
                    if (!DCV_FWD_LSUM_MFMA) rsum += p;
                }
        } else {
        float mx = m;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = max_halves(mx);
        // Lazy reference maximum: m moves only when a tile's maximum exceeds it by more than FWD_RESCALE_LOG2 (log2 units: p <= 2^8 until it
        // does — far from any fp32 / bf16 limit — and softmax is shift-invariant, so O / l is unchanged).  With 32 query rows per wave SOME row
        // finds a new maximum in almost every tile, so the eager form rescaled O (an exp + 33 multiplies per lane) nearly always; this one
        // after the first tile almost never.  (m = -inf before the first tile: the difference is +inf and alpha = 0.)
        if (__any((mx - m) * c > FWD_RESCALE_LOG2)) {
            const float alpha = __builtin_amdgcn_exp2f((m - mx) * c);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m = mx;
        }
        const float mc = m * c;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kb][r] * c - mc);
                s[kb][r] = p;
                rsum += p;
            }
        }
        l += rsum;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf = acc_to_frag(s[kb], ss);
                const int cc = kb * 4096 + ss * 2048;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = mfma32(join4(lds_tr_read(sKV, co[dt][0] + cc), lds_tr_read(sKV, co[dt][1] + cc)), pf, o[dt]);
                if (PS && DCV_FWD_LSUM_MFMA) lacc = mfma32(ones, pf, lacc);  // every row of the product = the column sums of P: the softmax denominator
            }
    };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    const int nfull = a.N / 64;
    int slot = 0;
    if (active) {
        for (int t = 0; t < nfull; ++t) {
            tile(No{}, Yes{}, t, slot);
            slot = (slot == ST - 1) ? 0 : slot + 1;
        }
        if (nfull < nt) tile(Yes{}, Yes{}, nfull, slot);
    } else {
        for (int t = 0; t < nfull; ++t) {
            tile(No{}, No{}, t, slot);
            slot = (slot == ST - 1) ? 0 : slot + 1;
        }
        if (nfull < nt) tile(Yes{}, No{}, nfull, slot);
    }

    if (PS && DCV_FWD_LSUM_MFMA) l = lacc[0];  // (all 32 rows equal; both lane halves hold the full sum over the keys)
    else l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (q < a.Nq) {
        bf16_t* op = a.o + ((size_t)b * a.N + q) * D + hh * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 v = pack4_bf16(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * dt + 8 * g + 4 * h) = v;
            }
        if (h == 0) a.lse[((size_t)b * a.H + hh) * a.N + q] = (PS ? m * (1.f / LOG2E) : m * a.scale) + logf(l);
    }
}

}  // namespace

static int attn_fwd_launch(const void* qkv, void* o, float* lse, int B, int N, int Nq, int H, int head_dim, float scale, bool ps, void* stream) {
    int rc = attn_check(qkv, B, N, H, head_dim);
    if (rc) return rc;
    if (!o || !lse) return DCV_ERR_NULL;
    if (Nq < 1 || Nq > N) return DCV_ERR_SHAPE;
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, nullptr, lse, nullptr, nullptr, B, N, H, scale, Nq};
    const int grid = B * H * ((Nq + FWD_QTILE - 1) / FWD_QTILE);
    if (ps) hipLaunchKernelGGL(attn_fwd3_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd3_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_attn_fwd_rows(const void* qkv, void* o, float* lse, int B, int N, int Nq, int H, int head_dim, float scale,
                                 void* stream) {
    return attn_fwd_launch(qkv, o, lse, B, N, Nq, H, head_dim, scale, false, stream);
}

// the q part of qkv holds q * scale * log2(e) (see attn_fwd3_kernel<PS>); LSE comes out in the same natural-log units as dcv_attn_fwd_rows
extern "C" int dcv_attn_fwd_rows_ps(const void* qkv, void* o, float* lse, int B, int N, int Nq, int H, int head_dim, void* stream) {
    return attn_fwd_launch(qkv, o, lse, B, N, Nq, H, head_dim, 0.f, true, stream);
}

extern "C" int dcv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, void* stream) {
    return dcv_attn_fwd_rows(qkv, o, lse, B, N, N, H, head_dim, scale, stream);
}

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
// DIAGNOSTIC TWIN of csrc/attn.hip (tools/probes; built only by _build.build_variant(..., instrumented=("attn.hip",))): timing-only ablations of the
// forward loop, selected by the bit mask DCV_FABL (outputs are WRONG for every non-zero mask):
//   1 no K/V DMA inside the loop   2 no vmcnt wait / s_barrier   4 no LDS fragment reads (register operands)   8 no v_exp (p = s c - m c)
//   256 no fma in front of v_exp (p = exp2(s): what a pre-scaled q and an accumulator that starts at -m c would leave)   16 no row-maximum pass   32 no QK MFMAs   64 no PV MFMAs   128 no conditional rescale of O
#ifndef DCV_FABL
#define DCV_FABL 0
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// Forward: one key tile at a time on a 3-stage K/V ring (48 KB -> 3 workgroups per CU; 136 VGPRs -> 3 waves per SIMD).
// Two earlier forms measured slower at the headline shape (tools/ab_bench.py): register-staged K/V with per-use address
// arithmetic 436-448 us; a 4-stage ring with the NEXT tile's score MFMAs issued ahead of the softmax (register ping-pong,
// 228 VGPRs, 2 waves per SIMD) 418-430 us; this one 397-405 us.  Occupancy beat intra-wave overlap.
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
    if ((BH & 7) == 0) {
        int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / nqt) * 8 + xcd;
        qt = slot % nqt;
    } else {
        bh = blockIdx.x / nqt;
        qt = blockIdx.x % nqt;
    }
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
    float m = -INFINITY, l = 0.f;
    const float c = a.scale * LOG2E;

    // COMPUTE = false: a wave without a valid query row only keeps the ring going; separate loops, not a branch in the loop
    // (attn_bwd.hip: the branch made the accumulators loop-carried phis resolved with register copies)
    auto tile = [&](auto MASKED, auto COMPUTE, int t, int slot) {
        if (!(DCV_FABL & 2)) {
            if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KV_DMA_PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (!(DCV_FABL & 1) && t + 2 < nt) kv_issue(t + 2, slot == 0 ? 2 : slot - 1);
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
            zero_acc(s[kb]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 kf;
                if (DCV_FABL & 4) { kf = qf[(ks + 1) & 3]; asm volatile("" : "+v"(kf)); }
                else kf = as_bf16x8(lds_read128(sKV, ro[ks] + kb * 4096));
                if (DCV_FABL & 32) { asm volatile("" ::"v"(kf)); s[kb][4 * ks] += 0.01f * (float)t; }
                else s[kb] = mfma32(kf, qf[ks], s[kb]);
            }
        }
        if constexpr (decltype(MASKED)::value) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + 32 * kb + acc_row(r, h) >= a.N) s[kb][r] = -INFINITY;
        }
        float mx = m;
        if (!(DCV_FABL & 16)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        } else if (t == 0) mx = 8.f;
        const float mc = mx * c;
        float rsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (DCV_FABL & 8) ? (s[kb][r] * c - mc) : (DCV_FABL & 256) ? __builtin_amdgcn_exp2f(s[kb][r]) : __builtin_amdgcn_exp2f(s[kb][r] * c - mc);
                s[kb][r] = p;
                rsum += p;
            }
        if (!(DCV_FABL & 128) && __any(mx > m)) {
            const float alpha = __builtin_amdgcn_exp2f((m - mx) * c);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m = mx;
        }
        l += rsum;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf = acc_to_frag(s[kb], ss);
                const int cc = kb * 4096 + ss * 2048;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    bf16x8 vf;
                    if (DCV_FABL & 4) { vf = qf[(ss + dt) & 3]; asm volatile("" : "+v"(vf)); }
                    else vf = join4(lds_tr_read(sKV, co[dt][0] + cc), lds_tr_read(sKV, co[dt][1] + cc));
                    if (DCV_FABL & 64) { asm volatile("" ::"v"(vf), "v"(pf)); }
                    else o[dt] = mfma32(vf, pf, o[dt]);
                }
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

    l += __shfl_xor(l, 32, 64);
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
        if (h == 0) a.lse[((size_t)b * a.H + hh) * a.N + q] = m * a.scale + logf(l);
    }
}

}  // namespace

extern "C" int dcv_attn_fwd_rows(const void* qkv, void* o, float* lse, int B, int N, int Nq, int H, int head_dim, float scale,
                                 void* stream) {
    int rc = attn_check(qkv, B, N, H, head_dim);
    if (rc) return rc;
    if (!o || !lse) return DCV_ERR_NULL;
    if (Nq < 1 || Nq > N) return DCV_ERR_SHAPE;
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, nullptr, lse, nullptr, nullptr, B, N, H, scale, Nq};
    const int grid = B * H * ((Nq + FWD_QTILE - 1) / FWD_QTILE);
    hipLaunchKernelGGL(attn_fwd3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, void* stream) {
    return dcv_attn_fwd_rows(qkv, o, lse, B, N, N, H, head_dim, scale, stream);
}

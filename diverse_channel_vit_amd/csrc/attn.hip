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

namespace {

#ifndef DCV_ABL2
#define DCV_ABL2 0  // timing-only ablation: 2 = no in-loop K/V DMA
#endif
// ------------------------------------------------------------------------------------------------
// Forward, software-pipelined: the score MFMAs of key tile t+1 are issued BEFORE the softmax of tile t, so the matrix
// pipe works under the softmax's VALU instructions of the same wave (counters on the first version: VALU busy 61 %, MFMA
// busy 32 %, both at once only 14 % of the time with 1.75 resident waves per SIMD).  The softmax arithmetic is written on
// float pairs (v_pk_fma_f32 / v_pk_add_f32: two elements per VALU issue).
#ifndef DCV_FWD_PK
#define DCV_FWD_PK 1  // measured: packed 421 us, scalar 428-430 us, first kernel 436-448 us (tools/ab_bench.py, headline shape)
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2;

__global__ __launch_bounds__(64 * FWD_WAVES) void attn_fwd2_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sKV[KV_STAGES * KV_STAGE_BYTES];
    static_assert(KV_STAGES == 4, "the wait counts below assume a 4-stage ring");
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = (a.N + FWD_QTILE - 1) / FWD_QTILE;
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
    // K/V DMA addressing: wave-uniform tile base (scalar) + loop-invariant per-lane byte offsets; only the partial last
    // tile recomputes the lane offsets (rows >= N clamp to N-1: they are masked, but must stay inside the tensor)
    static_assert(KV_DMA_PER_WAVE == 4, "two 8-row pieces of K and of V per wave");
    const int rowl = 16 * wave + (lane >> 3);
    const int lc8[2] = {((lane & 7) ^ swz64(rowl)) * 8, ((lane & 7) ^ swz64(rowl + 8)) * 8};
    const unsigned voff0 = (unsigned)(((size_t)rowl * rs + lc8[0]) * 2), voff1 = (unsigned)(((size_t)(rowl + 8) * rs + lc8[1]) * 2);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(sKV)) + 16 * wave * 128;
    const bf16_t* const kbase = Qb + D;
    auto kv_issue2 = [&](int t) {
        const unsigned sb = smem_base + (t & 3) * KV_STAGE_BYTES;
        const bf16_t* kt = kbase + (size_t)t * 64 * rs;  // scalar
        const bf16_t* vt = kt + D;
        unsigned o0 = voff0, o1 = voff1;
        if (t * 64 + 64 > a.N) {  // partial tile (wave-uniform branch)
            o0 = (unsigned)(((size_t)(min(t * 64 + rowl, a.N - 1) - t * 64) * rs + lc8[0]) * 2);
            o1 = (unsigned)(((size_t)(min(t * 64 + rowl + 8, a.N - 1) - t * 64) * rs + lc8[1]) * 2);
        }
        glds16s(kt, o0, sb);
        glds16s(vt, o0, sb + 8192);
        glds16s(kt, o1, sb + 1024);
        glds16s(vt, o1, sb + 8192 + 1024);
    };

    const int q = qt * FWD_QTILE + wave * 32 + r32;
    const int qc = min(q, a.N - 1);
    const bool active = qt * FWD_QTILE + wave * 32 < a.N;  // N = 1569: the 13th query tile has 33 rows, 3 of its 4 waves none
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Qb + (size_t)qc * rs + 16 * ks + 8 * h));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));  // hipcc's wait for these loads sits here, before any DMA
    for (int st = 0; st < 3; ++st)
        if (st < nt) kv_issue2(st);
    const LaneOffs lo = lane_offs(lane);

    f32x16 o[2];
    zero_acc(o[0]);
    zero_acc(o[1]);
    float m = -INFINITY;
    f32x2 l2 = {0.f, 0.f};
    const float c = a.scale * LOG2E;

    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    // S^T tile of key tile t (stage t % 4): 8 row reads + 8 MFMAs; MASKED (compile-time) only for the partial last tile
    auto scores = [&](auto MASKED, f32x16(&s)[2], int t) {
        const int so = (t & 3) * KV_STAGE_BYTES;
        int ro[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ro[ks] = lo.rows[ks] + so;  // 4 adds per tile; the rest are immediates
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            zero_acc(s[kb]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[kb] = mfma32(as_bf16x8(lds_read128(sKV, ro[ks] + kb * 4096)), qf[ks], s[kb]);
        }
        if constexpr (decltype(MASKED)::value) {  // keys >= N do not exist
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + 32 * kb + acc_row(r, h) >= a.N) s[kb][r] = -INFINITY;
        }
    };
    // online softmax of tile t held in s, then O^T += V_t^T . P^T
    auto softmax_pv = [&](f32x16(&s)[2], int t) {
        float mx = m;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * c;
#if DCV_FWD_PK
        const f32x2 c2 = {c, c}, mc2 = {mc, mc};
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v = {s[kb][r], s[kb][r + 1]};
                v = v * c2 - mc2;
                f32x2 p = {__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)};
                s[kb][r] = p.x;
                s[kb][r + 1] = p.y;
                rs2 += p;
            }
#else
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kb][r] * c - mc);
                s[kb][r] = p;
                if (kb) rs1 += p;
                else rs0 += p;
            }
        const f32x2 rs2 = {rs0, rs1};
#endif
        if (__any(mx > m)) {
            const float alpha = __builtin_amdgcn_exp2f((m - mx) * c);
            l2.x *= alpha;
            l2.y *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m = mx;
        }
        l2.x += rs2.x;
        l2.y += rs2.y;
        const int so = (t & 3) * KV_STAGE_BYTES + 8192;
        int co[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            co[dt][0] = lo.cols[dt][0] + so;
            co[dt][1] = lo.cols[dt][1] + so;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf = acc_to_frag(s[kb], ss);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int cc = kb * 4096 + ss * 2048;
                    o[dt] = mfma32(join4(lds_tr_read(sKV, co[dt][0] + cc), lds_tr_read(sKV, co[dt][1] + cc)), pf, o[dt]);
                }
            }
    };
    // one pipeline step: tile t is in `cur`; NEXT: 0 = there is no tile t+1, 1 = full tile, 2 = masked (partial) tile.
    // Tile t+1's scores go into `nxt` while `cur` goes through the softmax.
    auto step = [&](auto NEXT, f32x16(&cur)[2], f32x16(&nxt)[2], int t) {
        constexpr int nx = decltype(NEXT)::value;
        if constexpr (nx != 0) {  // stage t+1 must have landed; younger DMAs of this wave: stage t+2, if it exists
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KV_DMA_PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // stage t+1 visible to all; all waves are done with tile t-1 -> its buffer is free
#if !(DCV_ABL2 & 2)
        if (t + 3 < nt) kv_issue2(t + 3);
#endif
        if (active) {  // wave-uniform: a wave whose 32 query rows all lie beyond N only keeps the K/V ring going
            if constexpr (nx == 1) scores(No{}, nxt, t + 1);
            if constexpr (nx == 2) scores(Yes{}, nxt, t + 1);
            softmax_pv(cur, t);
        }
    };
    using N0 = std::integral_constant<int, 0>;
    using N1 = std::integral_constant<int, 1>;
    using N2 = std::integral_constant<int, 2>;
    const int nfull = a.N / 64;  // nt = nfull (+1 partial tile)
    // the last steps: `cur` holds tile t, t+1 is at most the partial tile
    auto finish = [&](f32x16(&cur)[2], f32x16(&nxt)[2], int t) {
        if (t + 1 < nt) {
            step(N2{}, cur, nxt, t);
            step(N0{}, nxt, cur, t + 1);
        } else {
            step(N0{}, cur, nxt, t);
        }
    };

    // prologue: tile 0's scores
    if (nt >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KV_DMA_PER_WAVE) : "memory");
    else if (nt == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KV_DMA_PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 sa[2], sb[2];
    if (active) {
        if (nfull == 0) scores(Yes{}, sa, 0);
        else scores(No{}, sa, 0);
    }
    int t = 0;
    for (; t + 2 < nfull; t += 2) {  // steps whose next tile is a full one, two at a time (register ping-pong)
        step(N1{}, sa, sb, t);
        step(N1{}, sb, sa, t + 1);
    }
    if (t + 1 < nfull) {
        step(N1{}, sa, sb, t);
        finish(sb, sa, t + 1);
    } else {
        finish(sa, sb, t);
    }

    float l = l2.x + l2.y;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (q < a.N) {
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

extern "C" int dcv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, void* stream) {
    int rc = attn_check(qkv, B, N, H, head_dim);
    if (rc) return rc;
    if (!o || !lse) return DCV_ERR_NULL;
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, nullptr, lse, nullptr, nullptr, B, N, H, scale};
    const int grid = B * H * ((N + FWD_QTILE - 1) / FWD_QTILE);
    hipLaunchKernelGGL(attn_fwd2_kernel, dim3(grid), dim3(64 * FWD_WAVES), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}


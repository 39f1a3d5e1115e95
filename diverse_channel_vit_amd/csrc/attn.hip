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
#include <type_traits>
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
    const bf16_t* qkv;  // [B,N,3,H,64]
    bf16_t* o;          // [B,N,H*64]           (fwd out / bwd in)
    const bf16_t* dO;   // [B,N,H*64]
    float* lse;         // [B,H,N]
    float* delta;       // [B,H,N]
    bf16_t* dqkv;       // [B,N,3,H,64]
    int B, N, H;
    float scale;
};

// stage a [64 rows][64 cols] bf16 tile: 512 chunks of 16 B, 256 threads x 2
struct Stage64 {
    uint4 r[2];
};
__device__ __forceinline__ void stage_load(Stage64& s, const bf16_t* base, size_t row_stride, int row0, int nrows_valid_max, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int q = tid + 256 * i, row = q >> 3, ch = q & 7;
        int gr = min(row0 + row, nrows_valid_max - 1);  // clamp: tail rows are masked by the caller
        s.r[i] = *reinterpret_cast<const uint4*>(base + (size_t)gr * row_stride + ch * 8);
    }
}
__device__ __forceinline__ void stage_store(const Stage64& s, char* tile, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int q = tid + 256 * i, row = q >> 3, ch = q & 7;
        lds_write128(tile, row * 128 + ((ch ^ swz64(row)) << 4), s.r[i]);
    }
}

// A-operand fragment from ROW reads: lane (r32,h) gets tile[row0 + r32][16*ks + 8h .. +7]
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int row0, int r32, int h, int ks) {
    int row = row0 + r32;
    return as_bf16x8(lds_read128(tile, row * 128 + (((2 * ks + h) ^ swz64(row)) << 4)));
}
// A-operand fragment of the TRANSPOSED tile for k-step s of a product that sums over tile rows in the
// accumulator-permuted order: lane (r32 = column c0 + r32 of the tile, h), element j = tile[rowbase + 16s +
// 8(j>>2) + 4h + (j&3)][c0 + r32]
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int rowbase, int s, int c0, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, li = lane & 15;
    const int row = rowbase + 16 * s + 4 * h + (li >> 2);
    const int col = c0 + 16 * g1 + 4 * (li & 3);
    bf16x4 lo = lds_tr_read(tile, lds64_off(row, col));
    bf16x4 hi = lds_tr_read(tile, lds64_off(row + 8, col));
    return join4(lo, hi);
}

__device__ __forceinline__ void zero_acc(f32x16& x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
}

// Per-lane LDS byte offsets computed ONCE per kernel; every fragment address in the tile loops is then
// lane_offset + compile-time constant (buffer, tile, 32-row block, k-step), which the compiler folds into the
// ds_read `offset:` immediate — the loops carry no address arithmetic.
struct LaneOffs {
    int rows[4];     // frag_rows: row r32 of a 32-row block, k-step ks      (+ 4096 per 32-row block)
    int cols[2][2];  // frag_cols: [dt][lo|hi] for rows 4h + (li>>2) (+8)    (+ 4096 per 32-row block, + 2048 per k-step)
};
__device__ __forceinline__ LaneOffs lane_offs(int lane) {
    LaneOffs o;
    const int h = lane >> 5, r32 = lane & 31, g1 = (lane >> 4) & 1, li = lane & 15;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o.rows[ks] = r32 * 128 + (((2 * ks + h) ^ swz64(r32)) << 4);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int row = 4 * h + (li >> 2), col = 32 * dt + 16 * g1 + 4 * (li & 3);
        o.cols[dt][0] = lds64_off(row, col);
        o.cols[dt][1] = lds64_off(row + 8, col);
    }
    return o;
}

// ------------------------------------------------------------------------------------------------
// K/V ring: 4 stages x (8 KB K + 8 KB V), filled by LDS-DMA three tiles ahead (48 KB in flight per workgroup).
// A register-staged single-tile prefetch left every key tile waiting ~1.5 us for its loads (measured: 3400 cycles
// per tile for 512 cycles of MFMA).  Each wave issues 4 DMA instructions per stage: rows [16w,16w+16) of K and of V.
#ifndef DCV_ABL2
#define DCV_ABL2 0  // timing-only ablation: 2 = no in-loop K/V DMA
#endif
constexpr int KV_STAGES = 4, KV_STAGE_BYTES = 16384;
constexpr int FWD_WAVES = 4, FWD_QTILE = 32 * FWD_WAVES;  // query rows per workgroup: K/V re-reads scale with 1/FWD_QTILE
constexpr int KV_DMA_PER_WAVE = 16 / FWD_WAVES;            // DMA instructions per stage per wave (8 rows x 128 B each; K: 8, V: 8)

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
    const bool tail = (a.N & 63) != 0;

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
        if constexpr (nx == 1) scores(No{}, nxt, t + 1);
        if constexpr (nx == 2) scores(Yes{}, nxt, t + 1);
        softmax_pv(cur, t);
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
    if (nfull == 0) scores(Yes{}, sa, 0);
    else scores(No{}, sa, 0);
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

// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a) {
    const int D = a.H * 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over B*N*H
    const size_t total = (size_t)a.B * a.N * a.H;
    if (idx >= total) return;
    const int hh = idx % a.H;
    const size_t bn = idx / a.H;
    const int n = bn % a.N, b = bn / a.N;
    const bf16_t* po = a.o + bn * D + hh * 64;
    const bf16_t* pd = a.dO + bn * D + hh * 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bf16x8 x = as_bf16x8(*reinterpret_cast<const uint4*>(po + 8 * i));
        bf16x8 y = as_bf16x8(*reinterpret_cast<const uint4*>(pd + 8 * i));
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)x[e] * (float)y[e];
    }
    a.delta[((size_t)b * a.H + hh) * a.N + n] = s;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sKV[2][2][64 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r32 = lane & 31;
    const int nqt = (a.N + 127) / 128;
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
    const bf16_t* Kb = Qb + D;
    const bf16_t* Vb = Qb + 2 * D;

    const int q = qt * 128 + wave * 32 + r32;
    const int qc = min(q, a.N - 1);
    bf16x8 qf[4], dof[4];
    const bf16_t* dop = a.dO + ((size_t)b * a.N + qc) * D + hh * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Qb + (size_t)qc * rs + 16 * ks + 8 * h));
        dof[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(dop + 16 * ks + 8 * h));
    }
    const size_t sidx = ((size_t)b * a.H + hh) * a.N + qc;
    const float lse2 = a.lse[sidx] * LOG2E;
    const float dlt = a.delta[sidx];
    const float c = a.scale * LOG2E;

    f32x16 dq[2];
    zero_acc(dq[0]);
    zero_acc(dq[1]);

    const int nt = (a.N + 63) / 64;
    Stage64 stK, stV;
    stage_load(stK, Kb, rs, 0, a.N, tid);
    stage_load(stV, Vb, rs, 0, a.N, tid);
    stage_store(stK, sKV[0][0], tid);
    stage_store(stV, sKV[0][1], tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const char* sK = sKV[t & 1][0];
        const char* sV = sKV[t & 1][1];
        if (t + 1 < nt) {
            stage_load(stK, Kb, rs, (t + 1) * 64, a.N, tid);
            stage_load(stV, Vb, rs, (t + 1) * 64, a.N, tid);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
            zero_acc(s);
            zero_acc(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(frag_rows(sK, 32 * kb, r32, h, ks), qf[ks], s);
                dp = mfma32(frag_rows(sV, 32 * kb, r32, h, ks), dof[ks], dp);
            }
            const bool tail = (t + 1) * 64 > a.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(s[r] * c - lse2);
                if (tail && (t * 64 + 32 * kb + acc_row(r, h) >= a.N)) p = 0.f;
                s[r] = p * (dp[r] - dlt);  // dS^T (the 1/sqrt(d) factor is applied once, to dQ)
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 dsf = acc_to_frag(s, ss);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(frag_cols(sK, 32 * kb, ss, 32 * dt, lane), dsf, dq[dt]);
            }
        }
        if (t + 1 < nt) {
            stage_store(stK, sKV[(t + 1) & 1][0], tid);
            stage_store(stV, sKV[(t + 1) & 1][1], tid);
        }
        __syncthreads();
    }
    if (q < a.N) {
        bf16_t* dst = a.dqkv + ((size_t)b * a.N + q) * rs + hh * 64;  // slot 0 = dQ
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 v = pack4_bf16(dq[dt][4 * g] * a.scale, dq[dt][4 * g + 1] * a.scale, dq[dt][4 * g + 2] * a.scale,
                                     dq[dt][4 * g + 3] * a.scale);
                *reinterpret_cast<uint2*>(dst + 32 * dt + 8 * g + 4 * h) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sQO[2][2][64 * 128];  // [buffer][Q|dO][64 queries x 128 B]
    __shared__ __attribute__((aligned(16))) float sLD[2][2][64];       // [buffer][LSE*log2e (+inf for rows >= N) | delta]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r32 = lane & 31;
    const int nkt = (a.N + 127) / 128;
    const int BH = a.B * a.H;
    int bh, kt;
    if ((BH & 7) == 0) {
        int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / nkt) * 8 + xcd;
        kt = slot % nkt;
    } else {
        bh = blockIdx.x / nkt;
        kt = blockIdx.x % nkt;
    }
    const int b = bh / a.H, hh = bh % a.H;
    const int D = a.H * 64;
    const size_t rs = (size_t)3 * D;
    const bf16_t* Qb = a.qkv + (size_t)b * a.N * rs + hh * 64;
    const bf16_t* Kb = Qb + D;
    const bf16_t* Vb = Qb + 2 * D;
    const bf16_t* dOb = a.dO + (size_t)b * a.N * D + hh * 64;
    const float* lseb = a.lse + ((size_t)b * a.H + hh) * a.N;
    const float* dltb = a.delta + ((size_t)b * a.H + hh) * a.N;

    const int key = kt * 128 + wave * 32 + r32;  // this lane's key
    const int kc = min(key, a.N - 1);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Kb + (size_t)kc * rs + 16 * ks + 8 * h));
        vf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Vb + (size_t)kc * rs + 16 * ks + 8 * h));
    }
    const float c = a.scale * LOG2E;
    f32x16 dk[2], dv[2];
    zero_acc(dk[0]);
    zero_acc(dk[1]);
    zero_acc(dv[0]);
    zero_acc(dv[1]);

    const int nt = (a.N + 63) / 64;
    Stage64 stQ, stO;
    float stL = 0.f, stD = 0.f;
    auto load_stats = [&](int t) {
        if (tid < 64) {
            int qq = t * 64 + tid;
            stL = (qq < a.N) ? lseb[qq] * LOG2E : INFINITY;  // +inf -> P = 0 for non-existent query rows
            stD = (qq < a.N) ? dltb[qq] : 0.f;
        }
    };
    stage_load(stQ, Qb, rs, 0, a.N, tid);
    stage_load(stO, dOb, (size_t)D, 0, a.N, tid);
    load_stats(0);
    stage_store(stQ, sQO[0][0], tid);
    stage_store(stO, sQO[0][1], tid);
    if (tid < 64) {
        sLD[0][0][tid] = stL;
        sLD[0][1][tid] = stD;
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const char* sQ = sQO[t & 1][0];
        const char* sO = sQO[t & 1][1];
        const float* sL = sLD[t & 1][0];
        const float* sD = sLD[t & 1][1];
        if (t + 1 < nt) {
            stage_load(stQ, Qb, rs, (t + 1) * 64, a.N, tid);
            stage_load(stO, dOb, (size_t)D, (t + 1) * 64, a.N, tid);
            load_stats(t + 1);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // S[q,key] = Q . K^T ; dP[q,key] = dO . V^T   (key on the lane, query rows in registers)
            f32x16 s, dp;
            zero_acc(s);
            zero_acc(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(frag_rows(sQ, 32 * qb, r32, h, ks), kf[ks], s);
                dp = mfma32(frag_rows(sO, 32 * qb, r32, h, ks), vf[ks], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&sL[32 * qb + 8 * g + 4 * h]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&sD[32 * qb + 8 * g + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = __builtin_amdgcn_exp2f(s[4 * g + e] * c - l4[e]);
                    s[4 * g + e] = p;
                    dp[4 * g + e] = p * (dp[4 * g + e] - d4[e]);
                }
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf = acc_to_frag(s, ss);
                bf16x8 dsf = acc_to_frag(dp, ss);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma32(frag_cols(sO, 32 * qb, ss, 32 * dt, lane), pf, dv[dt]);   // dV^T += dO^T . P
                    dk[dt] = mfma32(frag_cols(sQ, 32 * qb, ss, 32 * dt, lane), dsf, dk[dt]);  // dK^T += Q^T . dS
                }
            }
        }
        if (t + 1 < nt) {
            stage_store(stQ, sQO[(t + 1) & 1][0], tid);
            stage_store(stO, sQO[(t + 1) & 1][1], tid);
            if (tid < 64) {
                sLD[(t + 1) & 1][0][tid] = stL;
                sLD[(t + 1) & 1][1][tid] = stD;
            }
        }
        __syncthreads();
    }
    if (key < a.N) {
        bf16_t* dkp = a.dqkv + ((size_t)b * a.N + key) * rs + D + hh * 64;
        bf16_t* dvp = dkp + D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 v1 = pack4_bf16(dk[dt][4 * g] * a.scale, dk[dt][4 * g + 1] * a.scale, dk[dt][4 * g + 2] * a.scale,
                                      dk[dt][4 * g + 3] * a.scale);
                uint2 v2 = pack4_bf16(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(dkp + 32 * dt + 8 * g + 4 * h) = v1;
                *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g + 4 * h) = v2;
            }
    }
}

int check(const void* qkv, int B, int N, int H, int hd) {
    if (!qkv) return DCV_ERR_NULL;
    if (B <= 0 || N <= 0 || H <= 0) return DCV_ERR_SHAPE;
    if (hd != 64) return DCV_ERR_UNSUPPORTED;
    if ((uintptr_t)qkv & 15) return DCV_ERR_ALIGN;
    return DCV_OK;
}

}  // namespace

extern "C" int dcv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, void* stream) {
    int rc = check(qkv, B, N, H, head_dim);
    if (rc) return rc;
    if (!o || !lse) return DCV_ERR_NULL;
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, nullptr, lse, nullptr, nullptr, B, N, H, scale};
    const int grid = B * H * ((N + FWD_QTILE - 1) / FWD_QTILE);
    hipLaunchKernelGGL(attn_fwd2_kernel, dim3(grid), dim3(64 * FWD_WAVES), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static int bwd_check(const void* qkv, const void* o, const void* dO, const float* lse, float* delta_ws, int B, int N, int H, int hd) {
    int rc = check(qkv, B, N, H, hd);
    if (rc) return rc;
    if (!o || !dO || !lse || !delta_ws) return DCV_ERR_NULL;
    return DCV_OK;
}

extern "C" int dcv_attn_bwd_delta(const void* o, const void* dO, float* delta_ws, int B, int N, int H, int head_dim, void* stream) {
    if (!o || !dO || !delta_ws) return DCV_ERR_NULL;
    if (B <= 0 || N <= 0 || H <= 0) return DCV_ERR_SHAPE;
    if (head_dim != 64) return DCV_ERR_UNSUPPORTED;
    AttnArgs a{nullptr, (bf16_t*)o, (const bf16_t*)dO, nullptr, delta_ws, nullptr, B, N, H, 0.f};
    const size_t total = (size_t)B * N * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_attn_bwd_dq(const void* qkv, const void* dO, const float* lse, const float* delta, void* dqkv, int B, int N, int H,
                               int head_dim, float scale, void* stream) {
    int rc = bwd_check(qkv, dO, dO, lse, (float*)delta, B, N, H, head_dim);
    if (rc) return rc;
    if (!dqkv) return DCV_ERR_NULL;
    AttnArgs a{(const bf16_t*)qkv, nullptr, (const bf16_t*)dO, (float*)lse, (float*)delta, (bf16_t*)dqkv, B, N, H, scale};
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * H * ((N + 127) / 128)), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_attn_bwd_dkdv(const void* qkv, const void* dO, const float* lse, const float* delta, void* dqkv, int B, int N, int H,
                                 int head_dim, float scale, void* stream) {
    int rc = bwd_check(qkv, dO, dO, lse, (float*)delta, B, N, H, head_dim);
    if (rc) return rc;
    if (!dqkv) return DCV_ERR_NULL;
    AttnArgs a{(const bf16_t*)qkv, nullptr, (const bf16_t*)dO, (float*)lse, (float*)delta, (bf16_t*)dqkv, B, N, H, scale};
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3(B * H * ((N + 127) / 128)), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_attn_bwd(const void* qkv, const void* o, const void* dO, const float* lse, float* delta_ws, void* dqkv, int B, int N,
                            int H, int head_dim, float scale, void* stream) {
    int rc = bwd_check(qkv, o, dO, lse, delta_ws, B, N, H, head_dim);
    if (rc) return rc;
    if (!dqkv) return DCV_ERR_NULL;
    if ((rc = dcv_attn_bwd_delta(o, dO, delta_ws, B, N, H, head_dim, stream))) return rc;
    if ((rc = dcv_attn_bwd_dq(qkv, dO, lse, delta_ws, dqkv, B, N, H, head_dim, scale, stream))) return rc;
    return dcv_attn_bwd_dkdv(qkv, dO, lse, delta_ws, dqkv, B, N, H, head_dim, scale, stream);
}

// Attention backward for head_dim 64 on gfx950 (see attn.hip for the orientation conventions): delta, dQ, dK/dV.
// Compiled with -fno-slp-vectorize (see _build.py): hipcc's SLP packing of the dS arithmetic adds register-pair moves.
#include "attn_common.hpp"

namespace {

// Row statistics for the two backward kernels, written into the workspace ws (2*B*H*N floats):
//   ws[0 .. BHN)      = -delta,  delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]   (negated: dK/dV uses it as the initial value
//                                                                               of the dP accumulator, so dP - delta costs nothing)
//   ws[BHN .. 2 BHN)  = LSE * log2(e)                                          (the exp2 argument offset, no per-tile multiply)
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
    const size_t o = ((size_t)b * a.H + hh) * a.N + n;
    a.delta[o] = -s;
    a.delta[total + o] = a.lse[o] * LOG2E;
}

// ------------------------------------------------------------------------------------------------
// dQ.  The first version was bound by vector-instruction ISSUE: 285 vector instructions per key tile against 24 MFMAs
// (per-use LDS address arithmetic, register-staged K/V copies, the tail mask evaluated on every tile).  Here K/V arrive
// through a 3-stage LDS-DMA ring with scalar tile bases (as in the forward), every LDS address is a per-lane offset computed
// once plus an immediate, and only the partial last tile carries the masking code: 100 vector instructions per tile,
// 129 VGPRs (3 waves per SIMD), 551 -> 437 us at the headline shape.
#ifndef DCV_DQ_STAGES
#define DCV_DQ_STAGES 3
#endif
constexpr int DQ_STAGES = DCV_DQ_STAGES;
// PS ("pre-scaled q", see attn.hip): the q part of qkv holds q * scale * log2(e); the S accumulator starts at -LSE * log2(e) and the dP accumulator
// at -delta of the lane's query row (two loop-invariant register tuples), so p = exp2(S') and dS = p * dP' are one instruction each (cdna guide,
// appendix B: row constants as the initial accumulator).  The workspace then carries -LSE * log2(e) for the dK/dV kernel, which starts ITS score
// accumulator from those rows.
template <bool PS>
#if DCV_WPE_DQ
DCV_WAVES_PER_SIMD(DCV_WPE_DQ)
#endif
__global__ __launch_bounds__(256) void attn_bwd_dq2_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sKV[DQ_STAGES * KV_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = (a.N - a.key_lo + 127) / 128;  // query tiles launched: rows [key_lo, N); those beyond Nq only clear their dQ rows (dqkv is fully defined)
    const int BH = a.B * a.H;
    int bh, qt;
    attn_block_to_tile(blockIdx.x, BH, nqt, a.Nq == a.N ? a.N - a.key_lo : 0, bh, qt);
    qt += a.key_lo / 128;
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
        const bf16_t* kt = kbase + (size_t)t * 64 * rs;  // scalar
        const bf16_t* vt = kt + D;
        unsigned o0 = voff0, o1 = voff1;
        if (t * 64 + 64 > a.N) {  // partial tile: rows >= N clamp to N-1 (masked below, but must stay inside the tensor)
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
    const bool active = qt * 128 + wave * 32 < a.Nq;  // a wave without a single valid query row only keeps the ring going
    if (qt * 128 >= a.Nq) {  // whole tile beyond the processed query rows (workgroup-uniform): dQ = 0, nothing else
        if (q < a.N) {
            bf16_t* dst = a.dqkv + ((size_t)b * a.N + q) * rs + hh * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2*>(dst + 8 * i + 4 * h) = make_uint2(0u, 0u);
        }
        return;
    }
    bf16x8 qf[4], dof[4];
    const bf16_t* dop = a.dO + ((size_t)b * a.N + qc) * D + hh * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Qb + (size_t)qc * rs + 16 * ks + 8 * h));
        dof[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(dop + 16 * ks + 8 * h));
    }
    // Row statistics of this lane's query, computed here and written to the workspace for the dK/dV kernel (which runs
    // after this one): -delta = -sum_d dO*O and LSE*log2(e).  (attn_delta_kernel computes the same pair stand-alone.)
    const size_t sidx = ((size_t)b * a.H + hh) * a.N + qc;
    const float lse2 = a.lse[sidx] * LOG2E;
    float ndlt = 0.f;
    {
        const bf16_t* orow = a.o + ((size_t)b * a.N + qc) * D + hh * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 of = as_bf16x8(*reinterpret_cast<const uint4*>(orow + 16 * ks + 8 * h));
#pragma unroll
            for (int e = 0; e < 8; ++e) ndlt -= (float)of[e] * (float)dof[ks][e];
        }
        ndlt += __shfl_xor(ndlt, 32, 64);  // the two lane halves hold the two 8-element groups of every 16
        if (h == 0 && q < a.Nq) {
            a.delta[sidx] = ndlt;
            a.delta[(size_t)a.B * a.H * a.N + sidx] = PS ? -lse2 : lse2;
        }
    }
    const float c = a.scale * LOG2E;
    // consume the plain loads here: hipcc's wait for them must not land inside the tile loop (it would drain the ring)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]), "v"(dof[ks]));
    asm volatile("" ::"v"(lse2), "v"(ndlt));
    for (int st = 0; st < DQ_STAGES - 1; ++st)
        if (st < nt) kv_issue(st, st);
    const LaneOffs lo = lane_offs(lane);

    f32x16 dq[2];
    zero_acc(dq[0]);
    zero_acc(dq[1]);
    f32x16 sinit, dpinit;  // PS only
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sinit[r] = -lse2;
        dpinit[r] = ndlt;
    }

    // COMPUTE = false: a wave without a single valid query row only keeps the ring going (same DMA and barrier count).  The
    // two cases are separate LOOPS (below), not a branch inside one loop: with the branch inside, the accumulators became
    // loop-carried phis that hipcc resolved with a full register copy per tile (16 v_mov_b64) and twice the registers.
    auto tile = [&](auto MASKED, auto COMPUTE, int t, int slot) {
        if (DQ_STAGES >= 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KV_DMA_PER_WAVE) : "memory");  // younger: stage t+1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage t landed for everyone; everyone is done with tile t-1 -> its buffer is free
        if (t + DQ_STAGES - 1 < nt) kv_issue(t + DQ_STAGES - 1, slot == 0 ? DQ_STAGES - 1 : slot - 1);  // (t + STAGES - 1) % STAGES
        if constexpr (!decltype(COMPUTE)::value) return;
        const int so = slot * KV_STAGE_BYTES;
        int ro[4], co[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ro[ks] = lo.rows[ks] + so;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            co[dt][0] = lo.cols[dt][0] + so;
            co[dt][1] = lo.cols[dt][1] + so;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
            if constexpr (PS) {
                s = sinit;
                dp = dpinit;
            } else {
                zero_acc(s);
                zero_acc(dp);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(as_bf16x8(lds_read128(sKV, ro[ks] + kb * 4096)), qf[ks], s);
                dp = mfma32(as_bf16x8(lds_read128(sKV, ro[ks] + 8192 + kb * 4096)), dof[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = PS ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * c - lse2);
                if constexpr (decltype(MASKED)::value) {
                    if (t * 64 + 32 * kb + acc_row(r, h) >= a.N) p = 0.f;
                }
                s[r] = PS ? p * dp[r] : p * (dp[r] + ndlt);  // dS^T (the 1/sqrt(d) factor is applied once, to dQ)
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 dsf = acc_to_frag(s, ss);
                const int cc = kb * 4096 + ss * 2048;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = mfma32(join4(lds_tr_read(sKV, co[dt][0] + cc), lds_tr_read(sKV, co[dt][1] + cc)), dsf, dq[dt]);
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
            slot = (slot == DQ_STAGES - 1) ? 0 : slot + 1;
        }
        if (nfull < nt) tile(Yes{}, Yes{}, nfull, slot);
    } else {
        for (int t = 0; t < nfull; ++t) {
            tile(No{}, No{}, t, slot);
            slot = (slot == DQ_STAGES - 1) ? 0 : slot + 1;
        }
        if (nfull < nt) tile(Yes{}, No{}, nfull, slot);
    }

    if (q >= a.Nq && q < a.N) {  // rows of this tile beyond the processed ones: dQ = 0
        bf16_t* dst = a.dqkv + ((size_t)b * a.N + q) * rs + hh * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2*>(dst + 8 * i + 4 * h) = make_uint2(0u, 0u);
    }
    if (q < a.Nq) {
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
// dK, dV, second version: Q / dO query tiles and their LSE / delta rows arrive through a 4-stage LDS-DMA ring (scalar tile
// bases, loop-invariant lane offsets), LDS addresses are per-lane offsets computed once, the masking of non-existent
// query rows is confined to the partial last tile.  (The first version staged through registers; with this one
// the file is compiled with -fno-slp-vectorize: hipcc's SLP packing of the softmax-gradient arithmetic cost 48 v_mov per tile.)
constexpr int DKV_STAGES = 4, DKV_STAGE_BYTES = 16384 + 1024;  // Q tile | dO tile | lse[64] | delta[64] | 512 B scratch
constexpr int DKV_DMA_PER_WAVE = 5;
// PS: q pre-scaled (see above); the workspace rows hold -LSE * log2(e) and are read straight into the score accumulator; dK leaves with 1 / log2(e)
// (dK = scale dS^T Q = dS^T Q' / log2 e).
// TAIL2 (round 5): the launch for a key remainder of at most 64 keys per (batch, head) behind the third form (dcv_dkdv2_range).  There only two of
// the four waves had keys and every workgroup swept all query tiles alone on its CU: 52 us for 2 % of the keys.  With TAIL2 wave w takes key block
// w & 1 and the 32-query block w >> 1 of every tile — half the sweep per wave — and waves 2, 3 hand their partial dK / dV to waves 0, 1 through LDS at
// the end (one fixed order: deterministic; not the sequential order of the plain mode).
template <bool PS, bool TAIL2 = false>
#if DCV_WPE_DKDV
DCV_WAVES_PER_SIMD(DCV_WPE_DKDV)
#endif
__global__ __launch_bounds__(256) void attn_bwd_dkdv2_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sQO[DKV_STAGES * DKV_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = (a.N - a.key_lo + 127) / 128;  // key tiles launched: keys [key_lo, N)
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
    kt += a.key_lo / 128;
    const int b = bh / a.H, hh = bh % a.H;
    const int D = a.H * 64;
    const size_t rs = (size_t)3 * D;
    const bf16_t* Qb = a.qkv + (size_t)b * a.N * rs + hh * 64;
    const bf16_t* Kb = Qb + D;
    const bf16_t* Vb = Qb + 2 * D;
    const bf16_t* dOb = a.dO + (size_t)b * a.N * D + hh * 64;
    // waves 0/2 fetch LSE*log2e rows, waves 1/3 -delta rows; waves 2,3 write theirs to the scratch slot (uniform DMA count per wave)
    const float* statb = a.delta + ((wave & 1) ? 0 : (size_t)a.B * a.H * a.N) + ((size_t)b * a.H + hh) * a.N;  // odd: -delta, even: LSE*log2e
    const int nt = (a.Nq + 63) / 64;

    const int rowl = 16 * wave + (lane >> 3);
    const int lc8[2] = {((lane & 7) ^ swz64(rowl)) * 8, ((lane & 7) ^ swz64(rowl + 8)) * 8};
    const unsigned vq0 = (unsigned)(((size_t)rowl * rs + lc8[0]) * 2), vq1 = (unsigned)(((size_t)(rowl + 8) * rs + lc8[1]) * 2);
    const unsigned vo0 = (unsigned)(((size_t)rowl * D + lc8[0]) * 2), vo1 = (unsigned)(((size_t)(rowl + 8) * D + lc8[1]) * 2);
    const unsigned vs = (unsigned)lane * 4;
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(sQO));
    const unsigned smem_base = smem0 + 16 * wave * 128;
    const unsigned stat_dst = smem0 + 16384 + (wave & 1) * 256 + (wave >> 1) * 512;
    auto issue = [&](int t, int slot) {
        const unsigned sb = smem_base + slot * DKV_STAGE_BYTES;
        const bf16_t* qt = Qb + (size_t)t * 64 * rs;    // scalar bases
        const bf16_t* ot = dOb + (size_t)t * 64 * D;
        const float* st = statb + t * 64;
        unsigned q0 = vq0, q1 = vq1, o0 = vo0, o1 = vo1, s0 = vs;
        if (t * 64 + 64 > a.Nq) {  // partial tile: clamp query rows >= Nq to Nq-1 (valid data; their P is zeroed in the masked tile body)
            const int r0 = min(t * 64 + rowl, a.Nq - 1) - t * 64, r1 = min(t * 64 + rowl + 8, a.Nq - 1) - t * 64;
            q0 = (unsigned)(((size_t)r0 * rs + lc8[0]) * 2);
            q1 = (unsigned)(((size_t)r1 * rs + lc8[1]) * 2);
            o0 = (unsigned)(((size_t)r0 * D + lc8[0]) * 2);
            o1 = (unsigned)(((size_t)r1 * D + lc8[1]) * 2);
            s0 = (unsigned)(min(t * 64 + lane, a.Nq - 1) - t * 64) * 4;
        }
        glds16s(qt, q0, sb);
        glds16s(ot, o0, sb + 8192);
        glds16s(qt, q1, sb + 1024);
        glds16s(ot, o1, sb + 8192 + 1024);
        glds4s(st, s0, stat_dst + slot * DKV_STAGE_BYTES);
    };

    const int kblk = TAIL2 ? (wave & 1) : wave;     // this wave's 32-key block inside the workgroup's key tile
    const int key = kt * 128 + kblk * 32 + r32;    // this lane's key
    const int kc = min(key, a.N - 1);
    const bool active = kt * 128 + kblk * 32 < a.N;  // a wave without a single valid key only keeps the ring going
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Kb + (size_t)kc * rs + 16 * ks + 8 * h));
        vf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Vb + (size_t)kc * rs + 16 * ks + 8 * h));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(kf[ks]), "v"(vf[ks]));  // hipcc's wait for these loads sits here
    for (int st = 0; st < DKV_STAGES - 1; ++st)
        if (st < nt) issue(st, st);
    const LaneOffs lo = lane_offs(lane);
    const float c = a.scale * LOG2E;
    f32x16 dk[2], dv[2];
    zero_acc(dk[0]);
    zero_acc(dk[1]);
    zero_acc(dv[0]);
    zero_acc(dv[1]);

    auto tile = [&](auto MASKED, auto COMPUTE, int t, int slot) {  // COMPUTE: see attn_bwd_dq2_kernel
        const int rem = nt - 1 - t;  // younger stages in flight: min(rem, 2)
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DKV_DMA_PER_WAVE) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DKV_DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage t landed for everyone; everyone is done with tile t-1 -> its buffer is free
        if (t + 3 < nt) issue(t + 3, (slot + 3) & 3);
        if constexpr (!decltype(COMPUTE)::value) return;
        const int so = slot * DKV_STAGE_BYTES;
        int ro[4], co[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ro[ks] = lo.rows[ks] + so;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            co[dt][0] = lo.cols[dt][0] + so;
            co[dt][1] = lo.cols[dt][1] + so;
        }
        const int sto = so + 16384 + 16 * h;  // this lane-half's 4 consecutive query rows inside an 8-row group
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (TAIL2 && qb != (wave >> 1)) continue;  // wave-uniform: the other query block is the partner wave's
            f32x16 s, dp;
            f32x4 l4[4];
            if constexpr (!PS) zero_acc(s);
            // the dP accumulator starts at -delta of its query rows (registers 4g..4g+3 = rows 8g+4h..+3: one 16-byte read); PS: the score
            // accumulator likewise at -LSE log2(e)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                l4[g] = *reinterpret_cast<const f32x4*>(sQO + sto + (32 * qb + 8 * g) * 4);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sQO + sto + 256 + (32 * qb + 8 * g) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dp[4 * g + e] = d4[e];
                    if constexpr (PS) s[4 * g + e] = l4[g][e];
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(as_bf16x8(lds_read128(sQO, ro[ks] + qb * 4096)), kf[ks], s);
                dp = mfma32(as_bf16x8(lds_read128(sQO, ro[ks] + 8192 + qb * 4096)), vf[ks], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = PS ? __builtin_amdgcn_exp2f(s[4 * g + e]) : __builtin_amdgcn_exp2f(s[4 * g + e] * c - l4[g][e]);
                    if constexpr (decltype(MASKED)::value) {
                        if (t * 64 + 32 * qb + 8 * g + 4 * h + e >= a.Nq) p = 0.f;  // query row does not exist
                    }
                    s[4 * g + e] = p;
                    dp[4 * g + e] = p * dp[4 * g + e];  // dS = P * (dP - delta)
                }
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf = acc_to_frag(s, ss);
                bf16x8 dsf = acc_to_frag(dp, ss);
                const int cc = qb * 4096 + ss * 2048;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma32(join4(lds_tr_read(sQO, co[dt][0] + 8192 + cc), lds_tr_read(sQO, co[dt][1] + 8192 + cc)), pf, dv[dt]);
                    dk[dt] = mfma32(join4(lds_tr_read(sQO, co[dt][0] + cc), lds_tr_read(sQO, co[dt][1] + cc)), dsf, dk[dt]);
                }
            }
        }
    };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    const int nfull = a.Nq / 64;
    if (active) {
        for (int t = 0; t < nfull; ++t) tile(No{}, Yes{}, t, t & 3);
        if (nfull < nt) tile(Yes{}, Yes{}, nfull, nfull & 3);
    } else {
        for (int t = 0; t < nfull; ++t) tile(No{}, No{}, t, t & 3);
        if (nfull < nt) tile(Yes{}, No{}, nfull, nfull & 3);
    }

    if constexpr (TAIL2) {  // waves 2, 3 -> LDS -> waves 0, 1 (the ring is idle: every wave is past its last tile)
        __syncthreads();
        float* red = reinterpret_cast<float*>(sQO) + (wave & 1) * 64 * 64;  // [register][lane], 16 KB per key block
        if (wave >= 2) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    red[(dt * 16 + r) * 64 + lane] = dk[dt][r];
                    red[(32 + dt * 16 + r) * 64 + lane] = dv[dt][r];
                }
        }
        __syncthreads();
        if (wave >= 2) return;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dk[dt][r] += red[(dt * 16 + r) * 64 + lane];
                dv[dt][r] += red[(32 + dt * 16 + r) * 64 + lane];
            }
    }
    if (key < a.N) {
        bf16_t* dkp = a.dqkv + ((size_t)b * a.N + key) * rs + D + hh * 64;
        bf16_t* dvp = dkp + D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float ks_ = PS ? (1.f / LOG2E) : a.scale;
                uint2 v1 = pack4_bf16(dk[dt][4 * g] * ks_, dk[dt][4 * g + 1] * ks_, dk[dt][4 * g + 2] * ks_, dk[dt][4 * g + 3] * ks_);
                uint2 v2 = pack4_bf16(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(dkp + 32 * dt + 8 * g + 4 * h) = v1;
                *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g + 4 * h) = v2;
            }
    }
}

}  // namespace

static int bwd_check(const void* qkv, const void* o, const void* dO, const float* lse, float* delta_ws, int B, int N, int H, int hd) {
    int rc = attn_check(qkv, B, N, H, hd);
    if (rc) return rc;
    if (!o || !dO || !lse || !delta_ws) return DCV_ERR_NULL;
    return DCV_OK;
}

extern "C" int dcv_attn_bwd_delta(const void* o, const void* dO, const float* lse, float* delta_ws, int B, int N, int H, int head_dim,
                                  void* stream) {  // statistics of ALL query rows
    if (!o || !dO || !lse || !delta_ws) return DCV_ERR_NULL;
    if (B <= 0 || N <= 0 || H <= 0) return DCV_ERR_SHAPE;
    if (head_dim != 64) return DCV_ERR_UNSUPPORTED;
    AttnArgs a{nullptr, (bf16_t*)o, (const bf16_t*)dO, (float*)lse, delta_ws, nullptr, B, N, H, 0.f, N};
    const size_t total = (size_t)B * N * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static int dq_launch(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int Nq, int H,
                     int head_dim, float scale, bool ps, void* stream) {
    int rc = bwd_check(qkv, o, dO, lse, ws, B, N, H, head_dim);
    if (rc) return rc;
    if (!dqkv) return DCV_ERR_NULL;
    if (Nq < 1 || Nq > N) return DCV_ERR_SHAPE;
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, (const bf16_t*)dO, (float*)lse, ws, (bf16_t*)dqkv, B, N, H, scale, Nq};
#if DCV_DQ_FORM == 3
    if (ps && Nq == N) return dcv_dq3_launch(qkv, o, dO, lse, ws, dqkv, B, N, H, scale, (hipStream_t)stream);  // attn_bwd3q.hip (round 5); same sums, same order
#endif
    const dim3 grid(B * H * ((N + 127) / 128));
    if (ps) hipLaunchKernelGGL(attn_bwd_dq2_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq2_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}
static int dkdv_launch(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq, int H, int head_dim,
                       float scale, bool ps, void* stream) {
    int rc = bwd_check(qkv, dO, dO, lse, (float*)ws, B, N, H, head_dim);
    if (rc) return rc;
    if (!dqkv) return DCV_ERR_NULL;
    if (Nq < 1 || Nq > N) return DCV_ERR_SHAPE;
#if DCV_DKDV_FORM == 3
    if (ps) return dcv_dkdv3_launch(qkv, dO, lse, ws, dqkv, B, N, Nq, H, scale, (hipStream_t)stream);  // attn_bwd3.hip (round 5); same sums, same order
#endif
    AttnArgs a{(const bf16_t*)qkv, nullptr, (const bf16_t*)dO, (float*)lse, (float*)ws, (bf16_t*)dqkv, B, N, H, scale, Nq};
    const dim3 grid(B * H * ((N + 127) / 128));
    if (ps) hipLaunchKernelGGL(attn_bwd_dkdv2_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkdv2_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

#if DCV_DQ_FORM == 3
__attribute__((visibility("hidden"))) int dcv_dq2_range(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H, float scale,
                                                        int row_lo, hipStream_t stream) {
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, (const bf16_t*)dO, (float*)lse, ws, (bf16_t*)dqkv, B, N, H, scale, N};
    a.key_lo = row_lo;  // for the dQ kernel the range is one of query rows
    hipLaunchKernelGGL(attn_bwd_dq2_kernel<true>, dim3(B * H * ((N - row_lo + 127) / 128)), dim3(256), 0, stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

#endif

__attribute__((visibility("hidden"))) int dcv_dkdv2_range(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq, int H, float scale, int key_lo,
                    hipStream_t stream) {
    AttnArgs a{(const bf16_t*)qkv, nullptr, (const bf16_t*)dO, (float*)lse, (float*)ws, (bf16_t*)dqkv, B, N, H, scale, Nq};
    a.key_lo = key_lo;
    if (N - key_lo <= 64) hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<true, true>), dim3(B * H), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkdv2_kernel<true>, dim3(B * H * ((N - key_lo + 127) / 128)), dim3(256), 0, stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

extern "C" int dcv_attn_bwd_dq_rows(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N,
                                    int Nq, int H, int head_dim, float scale, void* stream) {
    return dq_launch(qkv, o, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, false, stream);
}

extern "C" int dcv_attn_bwd_dkdv_rows(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq,
                                      int H, int head_dim, float scale, void* stream) {
    return dkdv_launch(qkv, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, false, stream);
}

// Pre-scaled-q forms (the q part of qkv holds q * scale * log2(e), as dcv_attn_fwd_rows_ps reads it): same outputs as the plain entries — dQ is
// the gradient with respect to the UNSCALED q (the operand the qkv input- and weight-gradient GEMMs expect).  The workspace of the _ps pair is
// not interchangeable with the plain pair's (it carries -LSE log2 e).
extern "C" int dcv_attn_bwd_dq_rows_ps(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N,
                                       int Nq, int H, int head_dim, float scale, void* stream) {
    return dq_launch(qkv, o, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, true, stream);
}
extern "C" int dcv_attn_bwd_dkdv_rows_ps(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq,
                                         int H, int head_dim, float scale, void* stream) {
    return dkdv_launch(qkv, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, true, stream);
}
extern "C" int dcv_attn_bwd_rows_ps(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N,
                                    int Nq, int H, int head_dim, float scale, void* stream) {
    int rc = dq_launch(qkv, o, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, true, stream);  // also fills the workspace
    if (rc) return rc;
    return dkdv_launch(qkv, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, true, stream);
}

extern "C" int dcv_attn_bwd_rows(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int Nq,
                                 int H, int head_dim, float scale, void* stream) {
    int rc = dcv_attn_bwd_dq_rows(qkv, o, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, stream);  // also fills the workspace
    if (rc) return rc;
    return dcv_attn_bwd_dkdv_rows(qkv, dO, lse, ws, dqkv, B, N, Nq, H, head_dim, scale, stream);
}

extern "C" int dcv_attn_bwd_dq(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H,
                               int head_dim, float scale, void* stream) {
    return dcv_attn_bwd_dq_rows(qkv, o, dO, lse, ws, dqkv, B, N, N, H, head_dim, scale, stream);
}

extern "C" int dcv_attn_bwd_dkdv(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int H,
                                 int head_dim, float scale, void* stream) {
    return dcv_attn_bwd_dkdv_rows(qkv, dO, lse, ws, dqkv, B, N, N, H, head_dim, scale, stream);
}

extern "C" int dcv_attn_bwd(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H,
                            int head_dim, float scale, void* stream) {
    return dcv_attn_bwd_rows(qkv, o, dO, lse, ws, dqkv, B, N, N, H, head_dim, scale, stream);
}

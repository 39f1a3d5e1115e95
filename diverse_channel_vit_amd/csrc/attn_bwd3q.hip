// Attention backward, dQ, third form (head_dim 64, pre-scaled q, all query rows): the structure of attn_bwd3.hip turned by 90 degrees — ONE wave per
// SIMD, 64 QUERIES per wave (two 32-row blocks), persistent workgroups walking items (batch-head, 256-query block) while the K / V tiles of the
// (batch, head) stream through a 5-stage LDS-DMA ring.  Query on the lane: S'^T = K q'^T - LSE log2e and dP'^T = V dO^T - delta leave their MFMA chains
// with the row constants already in (initial accumulators), dS^T = exp2(S'^T) o dP'^T becomes the B operand of dQ^T += K^T dS^T in registers.
//
// UNIT = one 32-key x 32-query block of one wave, four per 64-key tile in the order (kb 0, qb 0) (kb 0, qb 1) (kb 1, qb 0) (kb 1, qb 1): the K / V row
// fragments of a key slice serve both query blocks.  Step u issues 12 MFMAs — S'^T and dP'^T of unit u + 1 (8), dQ^T of unit u - 1 (4) — with the
// vector work of unit u (16 v_exp, 16 v_mul, 8 v_cvt_pk: only dS is an operand here) and the LDS reads of the steps ahead in the 12 gaps, placed by
// the tables below (<= 24 issue cycles per gap).  Hazard rules, fences and pins: attn_bwd3.hip.
//
// Per item the kernel also produces what the second form's prologue did: delta = rowsum(dO o O) and the two statistics rows of the workspace
// (-delta, -LSE log2e) that the dK / dV kernel starts its accumulators from.  HBM traffic of the neighbouring items moves DURING an item: the next
// item's Q / dO rows by LDS-DMA into a per-wave region R (fragments at the seam), its O fragments and LSE by hidden global loads straight into
// spare accumulator registers, this item's dQ through R as whole 128-byte rows over the next item's first tiles.
//
// RESULT (round 5, profiles/r05_x4_attention_dq_one_wave_per_simd.txt): bit-identical to the second form on every shape tried (dQ and the workspace
// rows) and 1-11 % SLOWER (B64 H6 N1536: 334 against 331 us on random data, 269 against 251 on zeros; N1569: 373 / 360).  Why: this kernel carries
// 3.3 vector instructions + 0.7 LDS reads per MFMA (dK / dV: 3 + 1 per MFMA on 16 MFMAs per unit instead of 12); ONE wave issues a vector
// instruction every 4 cycles (two or more waves of a SIMD together every 2: MI355X_MICROARCH, cycle constants), so its 12 gaps are full (30-32 issue
// cycles of 32) and every stall is exposed, while the second form's three waves per SIMD have twice the vector issue rate.  Compiled only with
// -DDCV_DQ_FORM=3 (variant builds, tools/attn3_check.py): the product library keeps the second form behind dcv_attn_bwd_dq_rows_ps.
#include "attn3_common.hpp"
#if DCV_DQ_FORM == 3

namespace {

constexpr int Q3_ROWS = 256;                                  // query rows per workgroup item
constexpr int Q3_STAGES = 5, Q3_STAGE_BYTES = 16384;          // stage: K tile | V tile (64 keys each)
constexpr int Q3_R_BYTES = 16384;                             // per-wave region R: next item's Q tile | dO tile; this item's dQ tile overlays the Q tile

// vector work of a step by gap (elements 0..15 of the unit's accumulators): exponentials, the multiplies of elements whose exponential is at least one
// gap old, the packing of pairs whose multiplies are done
constexpr int Q3_EXP_AT[12][2] = {{0, 1}, {2, -1}, {3, 4}, {5, -1}, {6, 7}, {8, -1}, {9, 10}, {11, -1}, {12, 13}, {14, -1}, {15, -1}, {-1, -1}};
constexpr int Q3_MUL_AT[12][2] = {{-1, -1}, {0, 1}, {2, -1}, {3, 4}, {5, -1}, {6, 7}, {8, -1}, {9, 10}, {11, -1}, {12, 13}, {14, -1}, {15, -1}};
constexpr int Q3_CVT_AT[12] = {-1, -1, 0, -1, 1, 2, -1, 3, 4, 5, 6, 7};
// even step: the next key slice's row fragments (0-3 K, 4-7 V) go out once the MFMA that last read the register has issued (K[ks]: MFMA 2 ks, V[ks]:
// MFMA 2 ks + 1), this slice's transposed K fragments (first used by the next step's MFMA 8) around them
constexpr int Q3_ROWS_AT[12] = {-1, -1, 0, 4, 1, 5, 2, 6, 3, 7, -1, -1};
constexpr int Q3_TR_AT[12][2] = {{0, -1}, {1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {2, -1}, {3, -1}, {4, 5}, {6, 7}};

// global loads hidden from hipcc's vmcnt bookkeeping (a load it counts brings s_waitcnt vmcnt(0) to wherever it first touches the value — inside the
// tile loop, draining the ring); the kernel's own seam wait covers them
__device__ __forceinline__ void gload_a(bf16x8& d, const bf16_t* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload_f(float& d, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }

struct Q3Item {  // wave-uniform: one (batch-head, query block)
    int b, hh, q0;  // q0: first query row of this WAVE
    bool valid;
};

__attribute__((amdgpu_waves_per_eu(1, 1)))
__global__ __launch_bounds__(256) void attn_bwd_dq3p_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sKV[Q3_STAGES * Q3_STAGE_BYTES + 4 * Q3_R_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, lane_ = lane, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q_hi = a.key_hi;                            // query rows [0, q_hi) are this kernel's (the remainder: the second form)
    const int nqt = (q_hi + Q3_ROWS - 1) / Q3_ROWS;
    const int BH = a.B * a.H;
    const int D = a.H * 64;
    const size_t rs = (size_t)3 * D;
    const size_t BHN = (size_t)a.B * a.H * a.N;
    const int nt = (a.N + 63) / 64;  // key tiles
    const int G = gridDim.x;
    const bool xcd_map = (BH & 7) == 0 && (G & 7) == 0;
    auto item_of = [&](int k) {
        Q3Item it;
        int bh, qt;
        if (xcd_map) {
            const int m = (int)(blockIdx.x >> 3) + k * (G >> 3);
            it.valid = m < (BH >> 3) * nqt;
            bh = (m / nqt) * 8 + (int)(blockIdx.x & 7);
            qt = m % nqt;
        } else {
            const int m = (int)blockIdx.x + k * G;
            it.valid = m < BH * nqt;
            bh = m / nqt;
            qt = m % nqt;
        }
        if (!it.valid) bh = qt = 0;
        it.b = bh / a.H;
        it.hh = bh % a.H;
        it.q0 = qt * Q3_ROWS + wave * 64;
        return it;
    };
    auto q_base = [&](const Q3Item& it) { return a.qkv + (size_t)it.b * a.N * rs + it.hh * 64; };  // q rows of the item's (batch, head); K at + D, V at + 2 D

    const int rowl = 16 * wave + (lane >> 3);
    const int lc8[2] = {((lane & 7) ^ swz64(rowl)) * 8, ((lane & 7) ^ swz64(rowl + 8)) * 8};
    const unsigned vk0 = (unsigned)(((size_t)rowl * rs + lc8[0]) * 2), vk1 = (unsigned)(((size_t)(rowl + 8) * rs + lc8[1]) * 2);
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(sKV));
    const unsigned smem_base = smem0 + 16 * wave * 128;
    // ---- the ring cursor: next stage = key tile c_t of item c_it's (batch, head), at the running scalar pointer c_kp (K rows; V at + D)
    int c_k = 0, c_t = 0, c_slot = 0, n_ahead = 0;
    Q3Item c_it = item_of(0);
    const bf16_t* c_kp = q_base(c_it) + D;
    auto advance = [&]() {
        if (!c_it.valid) return;
        const unsigned sb = smem_base + c_slot * Q3_STAGE_BYTES;
        unsigned k0 = vk0, k1 = vk1;
        if (c_t * 64 + 64 > a.N) {  // partial tile: clamp keys >= N to N - 1 (their P is zeroed in the masked tile body)
            const int r0 = min(c_t * 64 + rowl, a.N - 1) - c_t * 64, r1 = min(c_t * 64 + rowl + 8, a.N - 1) - c_t * 64;
            k0 = (unsigned)(((size_t)r0 * rs + lc8[0]) * 2);
            k1 = (unsigned)(((size_t)r1 * rs + lc8[1]) * 2);
        }
        {
            unsigned keep;
            const bf16_t* c_vp = c_kp + D;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %5\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %3\n\t"
                "s_add_u32 m0, %5, 0x2000\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %4\n\t"
                "s_add_u32 m0, %5, 0x400\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %2, %3\n\t"
                "s_add_u32 m0, %5, 0x2400\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %2, %4\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(k0), "v"(k1), "s"(c_kp), "s"(c_vp), "s"(sb)
                : "memory", "scc");
        }
        c_slot = c_slot == Q3_STAGES - 1 ? 0 : c_slot + 1;
        ++n_ahead;
        c_kp += 64 * rs;
        if (++c_t == nt) {
            c_t = 0;
            c_it = item_of(++c_k);
            c_kp = q_base(c_it) + D;
        }
    };
    auto top = [&](int t) {  // every second tile: drain, barrier, refill the ring up to stage t + 4 (attn_bwd3.hip)
        if (t != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        while (n_ahead < Q3_STAGES && c_it.valid) advance();
    };

    const LaneOffs lo = lane_offs(lane);
    bf16x8 qf[2][4], dof[2][4];  // q' / dO fragments of this wave's 2 x 32 queries (AGPRs)
    bf16x8 nof[2][4];            // the next item's O fragments, in flight during the current item (AGPRs)
    float nlse[2];               // the next item's LSE of this lane's two queries (in flight)
    f32x16 dq[2][2];             // [query block][d half] (AGPRs)
    f32x16 sx[2], dpx[2];        // register set = query block
    f32x16 sinit[2], dpinit[2];  // -LSE log2e / -delta of the lane's query, broadcast: the initial accumulators
    bf16x8 dsf[2][2];            // [set][ss]
    bf16x8 rk[4], rv[4];         // row fragments of the current key slice
    bf16x8 trk[2][2][2];         // [key slice parity][ss][dt]: transposed K fragments
    char* const sR = sKV + Q3_STAGES * Q3_STAGE_BYTES + wave * Q3_R_BYTES;
    const unsigned smemR = smem0 + Q3_STAGES * Q3_STAGE_BYTES + wave * Q3_R_BYTES;
    // rows 8j .. 8j + 7 of item it's Q and dO rows of this wave -> the Q tile and the dO tile of R (swizzled like a ring tile)
    auto qo_dma_pair = [&](const Q3Item& it, int j) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));  // rebuilt per call: hoisted out of the tile loop this arithmetic costs registers the loop does not have
        const int row = 8 * j + (lane >> 3);
        const int q = min(it.q0 + row, a.N - 1);
        const int ch = ((lane & 7) ^ swz64(row)) << 3;
        glds16s(q_base(it), (unsigned)(((size_t)q * rs + ch) * 2), smemR + j * 1024);
        glds16s(a.dO + (size_t)it.b * a.N * D + it.hh * 64, (unsigned)(((size_t)q * D + ch) * 2), smemR + 8192 + j * 1024);
    };
    // the next item's O fragments and LSE of query block qb: hidden loads, landed by the item's seam
    auto o_loads = [&](const Q3Item& it, int qb) {
        int r = r32, hh_ = h;
        asm volatile("" : "+v"(r), "+v"(hh_));
        const int qc = min(it.q0 + 32 * qb + r, a.N - 1);
        const bf16_t* op = a.o + ((size_t)it.b * a.N + qc) * D + it.hh * 64 + 8 * hh_;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) gload_a(nof[qb][ks], op + 16 * ks);
        gload_f(nlse[qb], a.lse + ((size_t)it.b * a.H + it.hh) * a.N + qc);
    };
    // rows 8i .. 8i + 7 of the dQ tile in R -> HBM, 8 whole 128-byte rows per store instruction
    auto store_rows = [&](const Q3Item& it, int i) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const int row = 8 * i + (lane >> 3);
        const uint4 v = lds_read128(sR, row * 128 + (((lane & 7) ^ swz64(row)) << 4));
        if (it.q0 + row < q_hi && it.q0 + row < a.N)
            *reinterpret_cast<uint4*>(a.dqkv + ((size_t)it.b * a.N + it.q0 + row) * rs + it.hh * 64 + (lane & 7) * 8) = v;
    };
    auto acc_to_R = [&]() {  // dQ = scale * dS K: accumulators -> bf16 -> R (row = query, chunks permuted by swz64(row))
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x16& acc = dq[qb][dt];
                const int row = 32 * qb + r32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2 v = make_uint2(pack2_bf16(acc[4 * g] * a.scale, acc[4 * g + 1] * a.scale), pack2_bf16(acc[4 * g + 2] * a.scale, acc[4 * g + 3] * a.scale));
                    *reinterpret_cast<uint2*>(sR + row * 128 + (((4 * dt + g) ^ swz64(row)) << 4) + 8 * h) = v;
                }
            }
    };
    auto load_rows = [&](int so, int kb, int i) {  // i = 0..3: K row fragment ks = i of key slice kb in stage so; 4..7: V
        if (i < 4) rk[i] = as_bf16x8(lds_read128(sKV, so + lo.rows[i] + kb * 4096));
        else rv[i - 4] = as_bf16x8(lds_read128(sKV, so + lo.rows[i - 4] + 8192 + kb * 4096));
    };
    auto load_tr = [&](bf16x8 (&dst)[2][2], int so, int kb, int i, bf16x4 (&half)[2]) {  // i = 0..7 in the order the dQ MFMAs consume: (ss, dt), two reads each
        const int ss = i >> 2, dt = (i >> 1) & 1, hi = i & 1;
        half[hi] = lds_tr_read(sKV, so + lo.cols[dt][hi] + kb * 4096 + ss * 2048);
        if (hi) dst[ss][dt] = join4(half[0], half[1]);
    };

    // one step: J = position in the tile: unit (kb = J >> 1, qb = J & 1); MASKED: keys >= N of this tile carry p = 0
    auto step = [&](auto Jc, auto MASKED, auto LAST, int t, int so, int so_next) {
        constexpr int J = decltype(Jc)::value;
        constexpr int X = J & 1, Y = X ^ 1;      // this unit's register set (= its query block) / the neighbours'
        constexpr int kb = J >> 1;               // this unit's key slice
        constexpr int kbp = ((J + 3) >> 1) & 1;  // key slice of unit u - 1
        const bf16x8 dsfy[2] = {dsf[Y][0], dsf[Y][1]};
        const int lim = a.N - t * 64 - 32 * kb - 4 * h;  // MASKED: accumulator row (= key) index (r & 3) + 8 (r >> 2) >= lim does not exist
        bf16x4 half[2];
        float pv[16], dsv[16];
        unsigned dw[8];
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            // ---- the MFMA of gap g
            if (g < 8) {
                const int ks = g >> 1;
                if constexpr (J == 3 && decltype(LAST)::value) {
                    // no unit follows in this item
                } else if ((g & 1) == 0) {
                    if (ks == 0) mfma_vc(sx[Y], rk[0], qf[Y][0], sinit[Y]);
                    else mfma_vv(sx[Y], rk[ks], qf[Y][ks]);
                } else {
                    if (ks == 0) mfma_vc(dpx[Y], rv[0], dof[Y][0], dpinit[Y]);
                    else mfma_vv(dpx[Y], rv[ks], dof[Y][ks]);
                }
            } else {
                const int i = g - 8, ss = i >> 1, dt = i & 1;
                mfma_aa(dq[Y][dt], trk[kbp][ss][dt], dsfy[ss]);
            }
            // ---- the vector work of this unit
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = Q3_EXP_AT[g][e];
                if (r >= 0) {
                    float p = __builtin_amdgcn_exp2f(sx[X][r]);
                    if constexpr (decltype(MASKED)::value) {
                        if ((r & 3) + 8 * (r >> 2) >= lim) p = 0.f;
                    }
                    P3_PIN(p);
                    pv[r] = p;
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = Q3_MUL_AT[g][e];
                if (r >= 0) {
                    dsv[r] = pv[r] * dpx[X][r];  // dS^T (the 1 / sqrt(d) factor is applied once, to dQ)
                    P3_PIN(dsv[r]);
                }
            }
            if (Q3_CVT_AT[g] >= 0) {
                const int i = Q3_CVT_AT[g];
                dw[i] = pack2_bf16(dsv[2 * i], dsv[2 * i + 1]);
                P3_PIN(dw[i]);
            }
            // ---- the LDS reads of the steps ahead (even steps)
            if ((J & 1) == 0) {
                const int i = Q3_ROWS_AT[g];
                if (i >= 0) {
                    if (J == 0) load_rows(so, 1, i);
                    else if constexpr (!decltype(LAST)::value) load_rows(so_next, 0, i);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (Q3_TR_AT[g][e] >= 0) load_tr(trk[kb], so, kb, Q3_TR_AT[g][e], half);
            }
            P3_FENCE();
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) dsf[X][ss] = as_bf16x8(make_uint4(dw[4 * ss], dw[4 * ss + 1], dw[4 * ss + 2], dw[4 * ss + 3]));
    };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    using J3 = std::integral_constant<int, 3>;

    // HBM traffic of the neighbouring items, a few pieces per even tile (in_loop), or all of it at the seam when an item is too short for that
    const bool in_loop = nt >= 16;
    const int kv_t0 = (nt - 12) & ~1;
    Q3Item prev, nxt;
    bool prev_active = false, nxt_active = false;
    auto trickle = [&](int t) {
        if (!in_loop) return;
        if (prev_active && t < 8) {
            store_rows(prev, t);
            store_rows(prev, t + 1);
        }
        if (nxt_active) {
            if (t == kv_t0 - 2) o_loads(nxt, 0);
            if (t >= kv_t0 && t < kv_t0 + 8) {
                qo_dma_pair(nxt, t - kv_t0);
                qo_dma_pair(nxt, t - kv_t0 + 1);
            }
            if (t == kv_t0 + 8) o_loads(nxt, 1);
        }
    };
    int slot = 0;  // ring slot of the tile being computed
    auto tile = [&](auto LAST, int t) {
        if ((t & 1) == 0) {
            top(t);
            trickle(t);
        }
        --n_ahead;
        const int so = slot * Q3_STAGE_BYTES;
        slot = slot == Q3_STAGES - 1 ? 0 : slot + 1;
        const int so_next = slot * Q3_STAGE_BYTES;
        P3_FENCE();
        step(J0{}, LAST, LAST, t, so, so_next);
        step(J1{}, LAST, LAST, t, so, so_next);
        step(J2{}, LAST, LAST, t, so, so_next);
        step(J3{}, LAST, LAST, t, so, so_next);
    };

    for (int st = 0; st < Q3_STAGES - 1; ++st) advance();
    Q3Item cur = item_of(0);
    prev = cur;
    if (cur.valid && cur.q0 < q_hi) {
        for (int j = 0; j < 8; ++j) qo_dma_pair(cur, j);
        o_loads(cur, 0);
        o_loads(cur, 1);
    }
    for (int k = 0; cur.valid; ++k) {
        const bool active = cur.q0 < q_hi;  // a wave without a single query row of this kernel only keeps the ring going
        nxt = item_of(k + 1);
        nxt_active = nxt.valid && nxt.q0 < q_hi;
        // ---- seam: everything this wave has in flight has landed; behind the barrier that holds for every wave and the previous item's tiles are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (active) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float ndlt = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    qf[qb][ks] = as_bf16x8(lds_read128(sR, lo.rows[ks] + qb * 4096));
                    const bf16x8 d8 = as_bf16x8(lds_read128(sR, lo.rows[ks] + 8192 + qb * 4096));
                    bf16x8 o8 = nof[qb][ks];
                    asm volatile("" : "+a"(o8));  // read below the seam's wait
#pragma unroll
                    for (int e = 0; e < 8; ++e) ndlt -= (float)o8[e] * (float)d8[e];
                    dof[qb][ks] = d8;
                }
                ndlt += __shfl_xor(ndlt, 32, 64);  // the two lane halves hold the two 8-element groups of every 16
                float l = nlse[qb];
                asm volatile("" : "+v"(l));
                const float nl2 = -l * LOG2E;
                const int q = cur.q0 + 32 * qb + r32;
                if (h == 0 && q < q_hi && q < a.N) {  // the workspace rows the dK / dV kernel starts its accumulators from
                    const size_t sidx = ((size_t)cur.b * a.H + cur.hh) * a.N + q;
                    a.delta[sidx] = ndlt;
                    a.delta[BHN + sidx] = nl2;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sinit[qb][r] = nl2;
                    dpinit[qb][r] = ndlt;
                }
            }
            // as in attn_bwd3.hip's kv_frags: the v_accvgpr_writes of the fragments sit above this statement, two wait states before any asm MFMA
            asm volatile("s_nop 1"
                         : "+a"(qf[0][0]), "+a"(qf[0][1]), "+a"(qf[0][2]), "+a"(qf[0][3]), "+a"(qf[1][0]), "+a"(qf[1][1]), "+a"(qf[1][2]), "+a"(qf[1][3]),
                           "+a"(dof[0][0]), "+a"(dof[0][1]), "+a"(dof[0][2]), "+a"(dof[0][3]), "+a"(dof[1][0]), "+a"(dof[1][1]), "+a"(dof[1][2]), "+a"(dof[1][3]));
        }
        if (prev_active) acc_to_R();  // behind the fragment reads: one wave's LDS operations execute in order
        if (!in_loop) {
            if (prev_active)
                for (int i = 0; i < 8; ++i) store_rows(prev, i);
            if (nxt_active) {
                for (int j = 0; j < 8; ++j) qo_dma_pair(nxt, j);
                o_loads(nxt, 0);
                o_loads(nxt, 1);
            }
        }
        if (active) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) zero_acc(dq[qb][dt]);
            asm volatile("s_nop 1" : "+a"(dq[0][0]), "+a"(dq[0][1]), "+a"(dq[1][0]), "+a"(dq[1][1]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dsf[i][j] = trk[1][i][j] = as_bf16x8(make_uint4(0, 0, 0, 0));  // the first step's dQ MFMAs (unit -1) add zero
                }
            // prologue: the row fragments of key slice (tile 0, kb 0), S'^T and dP'^T of unit 0
            const int so = slot * Q3_STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 8; ++i) load_rows(so, 0, i);
            P3_FENCE();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks == 0) {
                    mfma_vc(sx[0], rk[0], qf[0][0], sinit[0]);
                    mfma_vc(dpx[0], rv[0], dof[0][0], dpinit[0]);
                } else {
                    mfma_vv(sx[0], rk[ks], qf[0][ks]);
                    mfma_vv(dpx[0], rv[ks], dof[0][ks]);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            P3_FENCE();
            for (int t = 0; t < nt - 1; ++t) tile(No{}, t);
            tile(Yes{}, nt - 1);
            // dQ^T of the last unit (set 1, key slice parity 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) mfma_aa(dq[1][i & 1], trk[1][i >> 1][i & 1], dsf[1][i >> 1]);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            P3_FENCE();
        } else {
            for (int t = 0; t < nt; ++t) {
                if ((t & 1) == 0) {
                    top(t);
                    trickle(t);
                }
                --n_ahead;
                slot = slot == Q3_STAGES - 1 ? 0 : slot + 1;
            }
        }
        prev = cur;
        prev_active = active;
        cur = nxt;
    }
    if (prev_active) {  // the last item's rows
        acc_to_R();
        for (int i = 0; i < 8; ++i) store_rows(prev, i);
    }
}

int q3_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return n;
}

}  // namespace

// dQ of all query rows (and the workspace rows).  The persistent kernel takes the query blocks of 256; a remainder of at most 128 rows per (batch, head)
// (N = 1569: 33) goes to the second form (attn_bwd.hip), launched behind it on the same stream for exactly those rows — as dcv_dkdv3_launch does for keys.
int dcv_dq3_launch(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H, float scale,
                   hipStream_t stream) {
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)o, (const bf16_t*)dO, (float*)lse, ws, (bf16_t*)dqkv, B, N, H, scale, N};
    const int rem = N % Q3_ROWS;
    const bool split = rem != 0 && rem <= 128 && N > Q3_ROWS;
    a.key_lo = 0;
    a.key_hi = split ? N - rem : N;
    const long items = (long)B * H * ((a.key_hi + Q3_ROWS - 1) / Q3_ROWS);
    const int cus = q3_cus();  // one workgroup per CU (144 KB of LDS, > 256 registers per lane)
    hipLaunchKernelGGL(attn_bwd_dq3p_kernel, dim3((unsigned)(items < cus ? items : cus)), dim3(256), 0, stream, a);
    if (hipGetLastError() != hipSuccess) return DCV_ERR_LAUNCH;
    if (split) return dcv_dq2_range(qkv, o, dO, lse, ws, dqkv, B, N, H, scale, N - rem, stream);
    return DCV_OK;
}
#endif  // DCV_DQ_FORM == 3

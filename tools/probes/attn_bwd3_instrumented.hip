// DIAGNOSTIC TWIN of diverse_channel_vit_amd/csrc/attn_bwd3.hip (snapshot of round 5): the same kernel WITH the timing-only ablation switches (-DDCV_K3_ABL=<mask>, tools/attn3_fit.py;
// results in profiles/r05_x3_*) and the cycle stamps (-DDCV_K3_STAMP; found useless in the persistent form: a pending s_memtime turns hipcc's counted lgkmcnt waits into lgkmcnt(0)).
// Built in place of the product source by _build.build_variant(..., instrumented=("attn_bwd3.hip",)); never part of libdcv_hip.so.
// Attention backward, dK / dV, third form (head_dim 64, pre-scaled q): ONE wave per SIMD owning the whole 512-entry register file, 64 keys per
// wave, persistent workgroups (cdna guide, appendix B "Attention backward" and "4-wave, one-wave-per-SIMD, persistent structure": one workgroup =
// 4 waves = 256 keys of one (batch, head); each wave keeps dK^T and dV^T of its 64 keys in accumulator registers while the workgroup sweeps the
// 32-row query slices).
//
// Why (round 5): the second form (attn_bwd.hip, 32 keys per wave, two waves per SIMD) runs at matrix + vector time (MFMA busy 0.49, co-execution
// 0.13): every 32 x 32 score block re-reads its Q / dO fragments from LDS (32 LDS instructions per 16 MFMAs and wave, eight waves per CU).  With 64
// keys per wave one set of Q / dO fragments (row reads for S and dP, transposed reads for dV^T and dK^T) feeds TWO key blocks: half the LDS reads
// and half the LDS-DMA pieces per score, and the vector work of one 32 x 32 block sits in the MFMA shadow of its neighbours (software pipeline
// below) instead of depending on a second wave's phase.  Same summation order as the second form: bit-identical outputs.
//
// A UNIT is one 32-query x 32-key block of one wave: (tile t, slice qb, key block kb), four units per 64-query tile in the order (0,0) (0,1) (1,0)
// (1,1).  Step u of the stream issues 16 MFMAs — S' and dP' of unit u+1 (8), then dV^T and dK^T of unit u-1 (8) — and, in the 16 gaps between
// them, the vector work of unit u (16 x {v_exp, v_mul}, 16 v_cvt_pk) plus one or two LDS fragment reads (<= 5 single-issue fillers per gap, at
// most one of them a v_exp: MI355X_MICROARCH, cycle constants).  MFMAs are inline asm so that their operands sit where the design needs them —
// dK^T / dV^T (128 registers) and the K / V fragments (64 + 64 of the next item) in the accumulator file, S' / dP' in arch VGPRs where the vector
// instructions read them — and every gap is fenced with sched_barrier(0), every filler value pinned by an empty asm: the order below IS the
// instruction stream.  Hazards hipcc does not see inside asm (cdna guide 5.7): an S' / dP' chain is finished 8 MFMAs (256 cycles) before its
// first vector read; P / dS fragments are written one step before the MFMAs that read them; the accumulators are read behind s_nop.
//
// Persistent: with one workgroup per CU nothing hides a workgroup's prologue (K / V fragments, first Q / dO stages: 11 k cycles measured) and
// epilogue (dK / dV stores at a row stride: 9 k cycles) — 23 % of the non-persistent form.  Here a workgroup walks ITEMS (batch-head, 256-key
// block) and every byte an item needs from or sends to HBM moves DURING an item, a few pieces per tile: the Q / dO ring runs across item seams
// (the cursor issues the next item's first stages during the last tiles); the next item's K / V rows arrive by LDS-DMA in a per-wave 16 KB
// region R (tiles nt-12 .. nt-5) and become register fragments at the seam; dK / dV leave the accumulators at the seam into the same region (bf16,
// row = key) and go to HBM as whole 128-byte rows, two store instructions per tile over the next item's first eight tiles.  (All 256 workgroups
// run in lockstep: with the loads and stores AT the seam every CU hit HBM at once while no CU computed — 68 of 418 us, timing-only ablation.)
#include "attn3_common.hpp"

namespace {

#ifndef DCV_K3_ABL
#define DCV_K3_ABL 0  // timing-only ablations (tools/attn3_fit.py on variant builds): 1 no stores, 2 no K/V DMA, 4 no ring DMA inside items, 8 no wait / barrier per tile,
#endif                 // 16 no vector fillers, 32 no LDS fragment reads in the steps, 64 no MFMAs
constexpr int K3_KEYS = 256;  // keys per workgroup item
constexpr int P3_STAGES = 5, P3_STAGE_BYTES = 16384 + 1024, P3_DMA = 5;  // stage: Q tile | dO tile | -LSE log2e [64] | -delta [64] | 512 B scratch
constexpr int P3_R_BYTES = 16384;  // per-wave region R: next item's K tile | V tile (swizzled like a ring tile), then this item's dK | dV tiles

#ifdef DCV_K3_STAMP  // diagnostic build only ((removed)): cycle stamps of wave 0 of every workgroup; never in the product library
__device__ unsigned long long k3_stamps[1024 * 8];
#define K3_NOW() __builtin_amdgcn_s_memtime()
#endif
// even step, gap g: which of the 16 row / statistics reads (load_rows index: 0-3 stl, 4-7 std_, 8-11 rq, 12-15 rdo) and which of the 8 transposed
// reads go out.  A register is reloaded only after the MFMA that last read it: stl / std_ (C operands of MFMA 0 / 1) from gap 2, rq[ks] after
// MFMA 2 ks, rdo[ks] after MFMA 2 ks + 1.
constexpr int P3_ROWS_AT[16][2] = {{-1, -1}, {-1, -1}, {0, 1}, {2, 3}, {4, 5}, {6, 7}, {8, 9}, {12, 13}, {10, 14}, {11, -1}, {15, -1},
                                   {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};
constexpr int P3_TR0_AT[16][2] = {{0, -1}, {1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1},
                                  {2, -1}, {3, -1}, {4, -1}, {5, -1}, {6, 7}};
struct K3Item {  // wave-uniform description of one (batch-head, key block); pointers are rebuilt from it where needed (SGPR budget)
    int b, hh, key0;  // key0: first key of this WAVE
    bool valid;
};

__attribute__((amdgpu_waves_per_eu(1, 1)))
__global__ __launch_bounds__(256) void attn_bwd_dkdv3p_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sQO[P3_STAGES * P3_STAGE_BYTES + 4 * P3_R_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, lane_ = lane, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = (a.key_hi + K3_KEYS - 1) / K3_KEYS;  // key blocks of a (batch, head): keys [0, key_hi)
    const int BH = a.B * a.H;
    const int D = a.H * 64;
    const size_t rs = (size_t)3 * D;
    const int nt = (a.Nq + 63) / 64;
    const int G = gridDim.x;
    const bool xcd_map = (BH & 7) == 0 && (G & 7) == 0;
    // k-th item of this workgroup.  XCD-aware: workgroups b and b + 8 share an XCD (and its L2): the items of one XCD are the key blocks of
    // the (batch, head) pairs congruent to it, walked in order by its G / 8 workgroups, so that the key blocks of a pair run close in time.
    auto item_of = [&](int k) {
        K3Item it;
        int bh, kt;
        if (xcd_map) {
            const int m = (int)(blockIdx.x >> 3) + k * (G >> 3);
            it.valid = m < (BH >> 3) * nkt;
            bh = (m / nkt) * 8 + (int)(blockIdx.x & 7);
            kt = m % nkt;
        } else {
            const int m = (int)blockIdx.x + k * G;
            it.valid = m < BH * nkt;
            bh = m / nkt;
            kt = m % nkt;
        }
        if (!it.valid) bh = kt = 0;
        it.b = bh / a.H;
        it.hh = bh % a.H;
        it.key0 = kt * K3_KEYS + wave * 64;
        return it;
    };
    auto q_base = [&](const K3Item& it) { return a.qkv + (size_t)it.b * a.N * rs + it.hh * 64; };  // q rows of the item's (batch, head); K at + D, V at + 2 D

    const int rowl = 16 * wave + (lane >> 3);
    const int lc8[2] = {((lane & 7) ^ swz64(rowl)) * 8, ((lane & 7) ^ swz64(rowl + 8)) * 8};
    const unsigned vq0 = (unsigned)(((size_t)rowl * rs + lc8[0]) * 2), vq1 = (unsigned)(((size_t)(rowl + 8) * rs + lc8[1]) * 2);
    const unsigned vo0 = (unsigned)(((size_t)rowl * D + lc8[0]) * 2), vo1 = (unsigned)(((size_t)(rowl + 8) * D + lc8[1]) * 2);
    const unsigned vs = (unsigned)lane * 4;
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(sQO));
    const unsigned smem_base = smem0 + 16 * wave * 128;
    const unsigned stat_dst = smem0 + 16384 + (wave & 1) * 256 + (wave >> 1) * 512;
    // The ring cursor: the next stage to issue is tile c_t of item c_it, at the running scalar pointers c_q / c_o / c_s; it runs P3_STAGES - 1
    // stages ahead of the tile being computed, across item seams.  n_ahead = stages issued and not yet consumed.
    int c_k = 0, c_t = 0, c_slot = 0, n_ahead = 0;
    K3Item c_it = item_of(0);
    const bf16_t* c_q;
    const bf16_t* c_o;
    const float* c_s;
    auto cursor_ptrs = [&]() {
        c_q = q_base(c_it);
        c_o = a.dO + (size_t)c_it.b * a.N * D + c_it.hh * 64;
        // waves 0/2 fetch the -LSE log2e rows, waves 1/3 the -delta rows; waves 2,3 write theirs to the scratch slot (uniform DMA count per wave)
        c_s = a.delta + ((wave & 1) ? 0 : (size_t)a.B * a.H * a.N) + ((size_t)c_it.b * a.H + c_it.hh) * a.N;
    };
    cursor_ptrs();
    auto advance = [&]() {
        if (!c_it.valid) return;
        const unsigned sb = smem_base + c_slot * P3_STAGE_BYTES;
        unsigned q0 = vq0, q1 = vq1, o0 = vo0, o1 = vo1, s0 = vs;
        if (c_t * 64 + 64 > a.Nq) {  // partial tile: clamp query rows >= Nq to Nq-1 (valid data; their P is zeroed in the masked tile body)
            const int r0 = min(c_t * 64 + rowl, a.Nq - 1) - c_t * 64, r1 = min(c_t * 64 + rowl + 8, a.Nq - 1) - c_t * 64;
            q0 = (unsigned)(((size_t)r0 * rs + lc8[0]) * 2);
            q1 = (unsigned)(((size_t)r1 * rs + lc8[1]) * 2);
            o0 = (unsigned)(((size_t)r0 * D + lc8[0]) * 2);
            o1 = (unsigned)(((size_t)r1 * D + lc8[1]) * 2);
            s0 = (unsigned)(min(c_t * 64 + lane, a.Nq - 1) - c_t * 64) * 4;
        }
        {  // the five pieces of this wave in one statement: M0 (the LDS destination) saved and restored once
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %9\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %6\n\t"
                "s_add_u32 m0, %9, 0x2000\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %2, %7\n\t"
                "s_add_u32 m0, %9, 0x400\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %3, %6\n\t"
                "s_add_u32 m0, %9, 0x2400\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %4, %7\n\t"
                "s_mov_b32 m0, %10\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dword %5, %8\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(q0), "v"(o0), "v"(q1), "v"(o1), "v"(s0), "s"(c_q), "s"(c_o), "s"(c_s), "s"(sb), "s"(stat_dst + c_slot * P3_STAGE_BYTES)
                : "memory", "scc");
        }
        c_slot = c_slot == P3_STAGES - 1 ? 0 : c_slot + 1;
        ++n_ahead;
        c_q += 64 * rs;
        c_o += 64 * D;
        c_s += 64;
        if (++c_t == nt) {
            c_t = 0;
            c_it = item_of(++c_k);
            cursor_ptrs();
        }
    };
#ifdef DCV_K3_STAMP
    unsigned long long st_wait = 0, st_issue = 0, st_seam = 0, st_items = 0;
    const unsigned long long st_entry = K3_NOW();
#endif
    // Every second tile (even t): everything this wave has in flight has landed — the stages issued two tiles ago (t + 1, t + 2: tile t + 1 reads the rows of
    // t + 2), the trickled pieces — then, behind the barrier, everyone is done with the tiles before t and the ring is refilled up to stage
    // t + 4.  One barrier and one scalar prologue per TWO tiles: at one per tile they cost 470 cycles of a 2800-cycle tile (timing-only ablation).
    auto top = [&](int t) {
#ifdef DCV_K3_STAMP
        unsigned long long s0 = 0, s1 = 0, s2 = 0;
        if (DCV_K3_STAMP >= 2) s0 = K3_NOW();
#endif
        if (!(DCV_K3_ABL & 8)) {
            if (t != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // t = 0: the item's seam has drained
            __builtin_amdgcn_s_barrier();
        }
#ifdef DCV_K3_STAMP
        if (DCV_K3_STAMP >= 2) s1 = K3_NOW();
#endif
        if (!(DCV_K3_ABL & 4))
            while (n_ahead < P3_STAGES && c_it.valid) advance();
#ifdef DCV_K3_STAMP
        if (DCV_K3_STAMP >= 2) {
            s2 = K3_NOW();
            st_wait += s1 - s0;
            st_issue += s2 - s1;
        }
#endif
    };

    const LaneOffs lo = lane_offs(lane);
    const int sto0 = 16384 + 16 * h;  // this lane-half's 4 consecutive query rows inside an 8-row group
    bf16x8 kf[2][4], vf[2][4];   // K / V fragments of this wave's 64 keys (AGPRs)
    f32x16 dk[2][2], dv[2][2];   // [key block][d half] (AGPRs)
    // register sets: X[0] = even units (kb 0), X[1] = odd units (kb 1)
    f32x16 sx[2], dpx[2];
    bf16x8 pf[2][2], dsf[2][2];  // [set][ss]
    bf16x8 rq[4], rdo[4];        // row fragments of the current slice (S', dP')
    f32x16 stl, std_;            // -LSE log2e / -delta rows of the current slice, accumulator layout
    bf16x8 tr0[2][2][2];         // [slice parity][dt][dO | Q]: transposed fragments, k-step ss = 0
    bf16x8 tr1[2][2];            // the same for ss = 1 (one slice at a time)
    // ---- region R of this wave
    char* const sR = sQO + P3_STAGES * P3_STAGE_BYTES + wave * P3_R_BYTES;
    const unsigned smemR = smem0 + P3_STAGES * P3_STAGE_BYTES + wave * P3_R_BYTES;
    // piece j (0..7) of item it's K and V rows of this wave: keys key0 + 8j .. + 7 (clamped into the tensor), 128 B each, into rows 8j .. of the
    // K tile and of the V tile; source chunks permuted as in a ring tile so that the fragment read below is lane_offs' row read
    // (the lane-dependent address arithmetic of these two is rebuilt per call from an opaque copy of the lane number: hoisted out of the tile loop it
    //  costs registers the loop does not have — and a spilled register's reload brings s_waitcnt vmcnt(0), which drains the ring)
    auto kv_dma_pair = [&](const K3Item& it, int j) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const int row = 8 * j + (lane >> 3);
        const int key = min(it.key0 + row, a.N - 1);
        const unsigned voff = (unsigned)(((size_t)key * rs + (((lane & 7) ^ swz64(row)) << 3)) * 2);
        const bf16_t* kb_ = q_base(it) + D;
        if (DCV_K3_ABL & 2) return;
        glds16s(kb_, voff, smemR + j * 1024);
        glds16s(kb_ + D, voff, smemR + 8192 + j * 1024);
    };
    auto kv_frags = [&]() {  // R -> kf, vf (at the seam, behind the wave's own vmcnt(0): wave-private data, no barrier)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[kb][ks] = as_bf16x8(lds_read128(sR, lo.rows[ks] + kb * 4096));
                vf[kb][ks] = as_bf16x8(lds_read128(sR, lo.rows[ks] + 8192 + kb * 4096));
            }
        // The fragments reach the accumulator file through v_accvgpr_write, which hipcc places where it likes — also directly in front of the asm
        // MFMA that reads them, and it pads no hazard towards an asm statement (seen: key block 1 wrong on single-tile items).  This statement
        // makes all sixteen registers "written" here, two wait states before anything can read them.
        asm volatile("s_nop 1"
                     : "+a"(kf[0][0]), "+a"(kf[0][1]), "+a"(kf[0][2]), "+a"(kf[0][3]), "+a"(kf[1][0]), "+a"(kf[1][1]), "+a"(kf[1][2]), "+a"(kf[1][3]),
                       "+a"(vf[0][0]), "+a"(vf[0][1]), "+a"(vf[0][2]), "+a"(vf[0][3]), "+a"(vf[1][0]), "+a"(vf[1][1]), "+a"(vf[1][2]), "+a"(vf[1][3]));
    };
    // rows 8i .. 8i + 7 of the dK tile and of the dV tile in R -> HBM: 8 whole 128-byte rows per store instruction
    auto store_pair = [&](const K3Item& it, int i) {
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const int row = 8 * i + (lane >> 3);
        const int off = row * 128 + (((lane & 7) ^ swz64(row)) << 4);
        const uint4 v0 = lds_read128(sR, off), v1 = lds_read128(sR, 8192 + off);
#if (DCV_K3_ABL & 1)  // timing-only ablation: no stores
        asm volatile("" ::"v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w));
#else
        if (it.key0 + row < a.key_hi) {
            bf16_t* const dst = a.dqkv + ((size_t)it.b * a.N + it.key0 + row) * rs + D + it.hh * 64 + (lane & 7) * 8;
            *reinterpret_cast<uint4*>(dst) = v0;
            *reinterpret_cast<uint4*>(dst + D) = v1;
        }
#endif
    };
    auto load_rows = [&](int so, int qb, int i) {  // i = 0..15: one ds_read_b128 of the 16 that make (stl, std_, rq, rdo) of slice qb in stage so
        if (i < 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sQO + so + sto0 + (32 * qb + 8 * i) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) stl[4 * i + e] = v[e];
        } else if (i < 8) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sQO + so + sto0 + 256 + (32 * qb + 8 * (i - 4)) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) std_[4 * (i - 4) + e] = v[e];
        } else if (i < 12) {
            rq[i - 8] = as_bf16x8(lds_read128(sQO, so + lo.rows[i - 8] + qb * 4096));
        } else {
            rdo[i - 12] = as_bf16x8(lds_read128(sQO, so + lo.rows[i - 12] + 8192 + qb * 4096));
        }
    };
    // i = 0..7: one transposed read of k-step ss of slice qb; order = the order the dV / dK MFMAs consume them: (dt 0: dO, Q), (dt 1: dO, Q), two reads each
    auto load_tr = [&](bf16x8 (&dst)[2][2], int so, int qb, int ss, int i, bf16x4 (&half)[2]) {
        const int dt = i >> 2, w = (i >> 1) & 1, hi = i & 1;
        const int cc = so + qb * 4096 + ss * 2048 + (w ? 0 : 8192);
        half[hi] = lds_tr_read(sQO, lo.cols[dt][hi] + cc);
        if (hi) dst[dt][w] = join4(half[0], half[1]);
    };

    // one step: J = position in the tile (unit (J >> 1, J & 1)); MASKED: rows >= Nq of this tile carry p = 0
    auto step = [&](auto Jc, auto MASKED, auto LAST, int t, int so, int so_next) {
        constexpr int J = decltype(Jc)::value;
        constexpr int X = J & 1, Y = X ^ 1;      // this unit's register set / the neighbours'
        constexpr int qb = J >> 1;               // this unit's slice
        constexpr int qbp = ((J + 3) >> 1) & 1;  // slice of unit u - 1
        const bf16x8 pfy[2] = {pf[Y][0], pf[Y][1]}, dsfy[2] = {dsf[Y][0], dsf[Y][1]};
        const int lim = a.Nq - t * 64 - 32 * qb - 4 * h;  // MASKED: accumulator row index (r & 3) + 8 (r >> 2) >= lim does not exist
        bf16x4 half[2];
        float pv[16], dsv[16];
        unsigned pw[8], dw[8];  // packed P / dS pairs: word i of k-step ss = i >> 2
        if (DCV_K3_ABL & 16)
            for (int i = 0; i < 8; ++i) pw[i] = dw[i] = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            // ---- the MFMA of gap g
            if (DCV_K3_ABL & 64) {
            } else if (g < 8) {
                const int ks = g >> 1;
                if constexpr (J == 3 && decltype(LAST)::value) {
                    // no unit follows in this item
                } else if ((g & 1) == 0) {
                    if (ks == 0) mfma_vc(sx[Y], rq[0], kf[Y][0], stl);
                    else mfma_vv(sx[Y], rq[ks], kf[Y][ks]);
                } else {
                    if (ks == 0) mfma_vc(dpx[Y], rdo[0], vf[Y][0], std_);
                    else mfma_vv(dpx[Y], rdo[ks], vf[Y][ks]);
                }
            } else {
                const int i = g - 8, ss = i >> 2, dt = (i >> 1) & 1;
                const bf16x8& aT = ss == 0 ? tr0[qbp][dt][i & 1] : tr1[dt][i & 1];
                if ((i & 1) == 0) mfma_aa(dv[Y][dt], aT, pfy[ss]);
                else mfma_aa(dk[Y][dt], aT, dsfy[ss]);
            }
            // ---- fillers: the vector work of this unit, one element per gap; the multiply and the packing trail the exponential by one and two
            // gaps (a dependent instruction right behind a v_exp waits for its result: nothing else is there to issue with one wave per SIMD)
            if (!(DCV_K3_ABL & 16)) {
            float p = __builtin_amdgcn_exp2f(sx[X][g]);
            if constexpr (decltype(MASKED)::value) {
                if ((g & 3) + 8 * (g >> 2) >= lim) p = 0.f;
            }
            P3_PIN(p);  // an empty asm that "rewrites" the value: its producer cannot sink below this gap, its consumers cannot rise above it
            pv[g] = p;
            if (g >= 1) {
                dsv[g - 1] = pv[g - 1] * dpx[X][g - 1];
                P3_PIN(dsv[g - 1]);
            }
            if (g >= 2 && (g & 1) == 0) {
                pw[(g - 2) >> 1] = pack2_bf16(pv[g - 2], pv[g - 1]);
                P3_PIN(pw[(g - 2) >> 1]);
                if (g >= 4) {
                    dw[(g - 4) >> 1] = pack2_bf16(dsv[g - 4], dsv[g - 3]);
                    P3_PIN(dw[(g - 4) >> 1]);
                }
            }
            if (g == 15) {
                dsv[15] = pv[15] * dpx[X][15];
                pw[7] = pack2_bf16(pv[14], pv[15]);
                dw[6] = pack2_bf16(dsv[12], dsv[13]);
                dw[7] = pack2_bf16(dsv[14], dsv[15]);
                P3_PIN(pw[7]);
                P3_PIN(dw[6]);
                P3_PIN(dw[7]);
            }
            }
            // ---- ... and the LDS reads of the steps ahead.  Even step: the next slice's rows and statistics go out as soon as the MFMA that last read
            // each register has issued (tables below) — 13 and more gaps before their first use, so the burst of the four lockstepped waves
            // (64 ds_read_b128 within ~300 cycles: the LDS array is busy for 256 of them) queues without anybody waiting for it — and k-step 0 of this
            // slice's transposed fragments (first used in the next step's second half) fills the rest.  Odd step: k-step 1 (used from MFMA 12 on).
            if (DCV_K3_ABL & 32) {
            } else if ((J & 1) == 0) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = P3_ROWS_AT[g][e];
                    if (i >= 0) {
                        if (J == 0) load_rows(so, 1, i);
                        else if constexpr (!decltype(LAST)::value) load_rows(so_next, 0, i);
                    }
                    const int j = P3_TR0_AT[g][e];
                    if (j >= 0) load_tr(tr0[qb], so, qb, 0, j, half);
                }
            } else if (g < 8) {
                load_tr(tr1, so, qb, 1, g, half);
            }
            P3_FENCE();
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            pf[X][ss] = as_bf16x8(make_uint4(pw[4 * ss], pw[4 * ss + 1], pw[4 * ss + 2], pw[4 * ss + 3]));
            dsf[X][ss] = as_bf16x8(make_uint4(dw[4 * ss], dw[4 * ss + 1], dw[4 * ss + 2], dw[4 * ss + 3]));
        }
    };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    using J3 = std::integral_constant<int, 3>;
    // dK (x 1 / log2 e: q arrived pre-scaled by scale log2 e) and dV of this wave's 64 keys: accumulators -> bf16 -> R (row = key, 16-byte chunks
    // permuted by swz64(row): the 8-byte writes of 16 lanes fall into 16 different bank pairs)
    auto acc_to_R = [&]() {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float mul = which == 0 ? 1.f / LOG2E : 1.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x16& acc = which == 0 ? dk[kb][dt] : dv[kb][dt];
                    const int row = 32 * kb + r32;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 v = make_uint2(pack2_bf16(acc[4 * g] * mul, acc[4 * g + 1] * mul), pack2_bf16(acc[4 * g + 2] * mul, acc[4 * g + 3] * mul));
                        *reinterpret_cast<uint2*>(sR + which * 8192 + row * 128 + (((4 * dt + g) ^ swz64(row)) << 4) + 8 * h) = v;
                    }
                }
        }
    };
    // HBM traffic of the neighbouring items, a few pieces per tile (in_loop), or all of it at the seam when an item is too short for that
    const bool in_loop = nt >= 16;  // 8 tiles of stores, then 8 tiles of K / V rows ending 2-4 tiles before the seam
    K3Item prev, nxt;
    bool prev_active = false, nxt_active = false;
    const int kv_t0 = (nt - 12) & ~1;  // first tile of the K / V window
    auto trickle = [&](int t) {  // at even tiles, behind top(): two pairs each, two tiles in flight before the next top waits for them
        if (!in_loop) return;
        if (prev_active && t < 8) {
            store_pair(prev, t);
            store_pair(prev, t + 1);
        }
        if (nxt_active && t >= kv_t0 && t < kv_t0 + 8) {
            kv_dma_pair(nxt, t - kv_t0);
            kv_dma_pair(nxt, t - kv_t0 + 1);
        }
    };

    int slot = 0;  // ring slot of the tile being computed
    // LAST: the last tile of an item — always the masked body (a full last tile masks nothing) — skips S' / dP' of the unit after it (the next
    // item's first unit is computed in that item's prologue, with ITS K / V fragments)
    auto tile = [&](auto LAST, int t) {
        if ((t & 1) == 0) {
            top(t);
            trickle(t);
        }
        --n_ahead;  // this tile's stage is being consumed
        const int so = slot * P3_STAGE_BYTES;
        slot = slot == P3_STAGES - 1 ? 0 : slot + 1;
        const int so_next = slot * P3_STAGE_BYTES;
        P3_FENCE();
        step(J0{}, LAST, LAST, t, so, so_next);
        step(J1{}, LAST, LAST, t, so, so_next);
        step(J2{}, LAST, LAST, t, so, so_next);
        step(J3{}, LAST, LAST, t, so, so_next);
    };

    for (int st = 0; st < P3_STAGES - 1; ++st) advance();
    K3Item cur = item_of(0);
    prev = cur;
    if (cur.valid && cur.key0 < a.key_hi)
        for (int j = 0; j < 8; ++j) kv_dma_pair(cur, j);
    for (int k = 0; cur.valid; ++k) {
#ifdef DCV_K3_STAMP
        const unsigned long long q0 = K3_NOW();
#endif
        const bool active = cur.key0 < a.key_hi;  // a wave without a single valid key only keeps the ring going
        nxt = item_of(k + 1);
        nxt_active = nxt.valid && nxt.key0 < a.key_hi;
        // ---- seam: everything this wave has in flight has landed (its pieces of the stages issued so far, this item's K / V rows in R, the
        // previous item's stores); after the barrier the stages have landed for every wave, and everyone is done with the previous item's last tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (active) kv_frags();
        if (prev_active) acc_to_R();  // behind the fragment reads: one wave's LDS operations execute in order
        if (!in_loop) {
            if (prev_active)
                for (int i = 0; i < 8; ++i) store_pair(prev, i);
            if (nxt_active)
                for (int j = 0; j < 8; ++j) kv_dma_pair(nxt, j);
        }
        if (active) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    zero_acc(dk[kb][dt]);
                    zero_acc(dv[kb][dt]);
                }
            asm volatile("s_nop 1"  // as in kv_frags: the v_accvgpr_writes of the zeroing sit above this statement, not in front of an asm MFMA
                         : "+a"(dk[0][0]), "+a"(dk[0][1]), "+a"(dk[1][0]), "+a"(dk[1][1]), "+a"(dv[0][0]), "+a"(dv[0][1]), "+a"(dv[1][0]), "+a"(dv[1][1]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    pf[i][j] = dsf[i][j] = tr1[i][j] = as_bf16x8(make_uint4(0, 0, 0, 0));  // the first step's dV / dK MFMAs (unit -1) add zero
                    tr0[1][i][j] = as_bf16x8(make_uint4(0, 0, 0, 0));
                    asm volatile("" : "+v"(pf[i][j]), "+v"(dsf[i][j]), "+v"(tr1[i][j]), "+v"(tr0[1][i][j]));  // pinned: see the product source
                }
            // prologue: the rows of slice (0, 0), S' and dP' of unit 0
            const int so = slot * P3_STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 16; ++i) load_rows(so, 0, i);
            P3_FENCE();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks == 0) {
                    mfma_vc(sx[0], rq[0], kf[0][0], stl);
                    mfma_vc(dpx[0], rdo[0], vf[0][0], std_);
                } else {
                    mfma_vv(sx[0], rq[ks], kf[0][ks]);
                    mfma_vv(dpx[0], rdo[ks], vf[0][ks]);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            P3_FENCE();
        }
#ifdef DCV_K3_STAMP
        const unsigned long long q1 = K3_NOW();
        st_seam += q1 - q0;
        ++st_items;
#endif
        if (active) {
            for (int t = 0; t < nt - 1; ++t) tile(No{}, t);
            tile(Yes{}, nt - 1);
            // dV^T / dK^T of the last unit (set 1, slice parity 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ss = i >> 2, dt = (i >> 1) & 1;
                const bf16x8& aT = ss == 0 ? tr0[1][dt][i & 1] : tr1[dt][i & 1];
                if ((i & 1) == 0) mfma_aa(dv[1][dt], aT, pf[1][ss]);
                else mfma_aa(dk[1][dt], aT, dsf[1][ss]);
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            P3_FENCE();
        } else {
            for (int t = 0; t < nt; ++t) {
                if ((t & 1) == 0) {
                    top(t);
                    trickle(t);
                }
                --n_ahead;
                slot = slot == P3_STAGES - 1 ? 0 : slot + 1;
            }
        }
        prev = cur;
        prev_active = active;
        cur = nxt;
    }
    if (prev_active) {  // the last item's outputs
        acc_to_R();
        for (int i = 0; i < 8; ++i) store_pair(prev, i);
    }
#ifdef DCV_K3_STAMP
    if (tid == 0 && blockIdx.x < 1024) {
        unsigned long long* o = k3_stamps + (size_t)blockIdx.x * 8;
        o[0] = st_entry; o[1] = st_seam; o[2] = st_items; o[3] = K3_NOW(); o[4] = st_wait; o[5] = st_issue; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = nt;
    }
#endif
}

}  // namespace
#ifdef DCV_K3_STAMP
extern "C" int dcv_k3_stamps(void* host_dst, size_t bytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(k3_stamps), bytes < sizeof(k3_stamps) ? bytes : sizeof(k3_stamps)) == hipSuccess ? 0 : 1;
}
#endif

static int k3_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return n;
}

// dK / dV of all keys.  The persistent kernel takes the key blocks of 256; a remainder of at most 128 keys per (batch, head) (N = 1569: 33) would
// cost it a whole item time per pair with one wave of four at work and leave the workgroups unevenly loaded (10.5 items each): that remainder
// goes to the second form (attn_bwd.hip, 128 keys per workgroup), launched behind it on the same stream for exactly those keys.
int dcv_dkdv3_launch(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq, int H, float scale,
                     hipStream_t stream) {
    AttnArgs a{(const bf16_t*)qkv, nullptr, (const bf16_t*)dO, (float*)lse, (float*)ws, (bf16_t*)dqkv, B, N, H, scale, Nq};
    const int rem = N % K3_KEYS;
    const bool split = rem != 0 && rem <= 128 && N > K3_KEYS;
    a.key_lo = 0;
    a.key_hi = split ? N - rem : N;
    const long items = (long)B * H * ((a.key_hi + K3_KEYS - 1) / K3_KEYS);
    const int cus = k3_cus();  // one workgroup per CU (149 KB of LDS, > 256 registers per lane)
    hipLaunchKernelGGL(attn_bwd_dkdv3p_kernel, dim3((unsigned)(items < cus ? items : cus)), dim3(256), 0, stream, a);
    if (hipGetLastError() != hipSuccess) return DCV_ERR_LAUNCH;
    if (split) return dcv_dkdv2_range(qkv, dO, lse, ws, dqkv, B, N, Nq, H, scale, N - rem, stream);
    return DCV_OK;
}

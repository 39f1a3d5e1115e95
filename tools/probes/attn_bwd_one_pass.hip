// RETIRED from the product library in round 5: the one-pass attention backward (dQ by an ordered hand-off between key-block workgroups).
// Parity-green but 28-70 % slower than the dQ + dK/dV pair wherever measured (profiles/r04_x1_*); kept here as the record of that experiment.  It built as
// csrc/attn_bwd_fused.hip against include/dcv.h of round 4 (three entries: dcv_attn_bwd_fused, _ws_bytes, _err_ptr), which no longer declares them.
// Attention backward in ONE pass for head_dim 64 on gfx950: five N x N x 64 products per (batch, head) instead of the seven of the
// dQ + dK/dV pair in attn_bwd.hip, and Q / K / V / dO are read once.  Reference semantics: models/vit.py:121-144 (the backward of
// softmax(q k^T * hd^-0.5) v as autograd computes it).
//
// Structure (cdna guide, Appendix B 'Attention backward'): a workgroup = 8 waves = 256 keys of one (batch, head); every wave keeps
// dK^T and dV^T of its 32 keys in registers while the workgroup sweeps the 64-row query tiles, so dK and dV need no sum across
// workgroups.  Scores are computed with the key on the MFMA lane (S^T and dP^T accumulators are the B operands of the dV^T / dK^T
// products); only dS crosses LDS, once, for dQ = dS K: every wave writes the bf16 dS of its keys to a [256 keys][64 queries] image
// and, one tile later, the eight waves share the 64 x 64 dQ tile of the PREVIOUS query tile (MFMA 16x16x32, K^T from a stationary
// LDS image of the workgroup's keys).
//
// dQ across the J = ceil(N / 256) key blocks of a (batch, head): an ORDERED HAND-OFF with plain stores, no atomics, bit-reproducible.
// The J workgroups of a chain walk the T query tiles in rotated order — workgroup j handles tile (k + 3 j) mod T in its slot k — so at
// any time the J workgroups work on J different tiles and the partial sum of a tile travels from workgroup j + 1 to j (and from 0 to
// J - 1 at the wrap) three slots apart: no pipeline fill, nobody waits in the steady state.  A tile's order of summation is fixed by
// (tile, J, T) alone.  Protocol (cdna guide, Guideline 16 R1 in its write-through form; MI355X_MICROARCH 'Valid forms', row 1):
// the f32 partial tile is stored with sc1 (write-through) stores in a lane-linear layout, so consumer wave w reads exactly the bytes
// producer wave w wrote; each producer wave publishes for itself — after its own s_waitcnt has covered its stores it adds 1 to ITS
// word of the chain's flag block; the consumer wave polls that one word with an sc1 load and then loads the payload with sc1 loads.
// Spins are bounded (error word set, results then undefined but the grid drains).  All members of a chain must be resident together:
// the launch maps a chain to consecutive block ids of one XCD (J <= 32) or to consecutive ids (J <= 256) and relies on in-order
// dispatch for speed only — a member that is not yet resident makes its neighbours wait, never deadlock, as long as one whole
// chain fits on the chip next to it.
//
// Every vector-memory operation inside the tile loop is issued from inline asm and counted by hand (vmcnt counts loads, stores,
// atomics and LDS-DMA together, in issue order): per iteration and wave exactly  P (poll, 1)  D (ring stage, 3)  L (partial in, 2)
// S (partial or final out, 2)  A (publish, 1)  — all unconditional, so the counted waits hold on every path.
#include "attn_common.hpp"

namespace {

constexpr int FB_WAVES = 8, FB_KEYS = 32 * FB_WAVES;  // keys per workgroup
constexpr int FB_STAGES = 3, FB_STAGE_BYTES = 16384 + 1024;  // Q tile | dO tile | lse2[64] | -delta[64] | 512 B scratch
constexpr int FB_RING = FB_STAGES * FB_STAGE_BYTES;
constexpr int FB_KT_OFF = FB_RING;                           // K image of the workgroup's keys [256][64] bf16
constexpr int FB_DS_OFF = FB_KT_OFF + FB_KEYS * 128;         // two dS images [256 keys][64 queries] bf16
constexpr int FB_DS_BYTES = FB_KEYS * 128;
constexpr int FB_SMEM = FB_DS_OFF + 2 * FB_DS_BYTES;         // 150 528 B: one workgroup per CU
constexpr unsigned FB_SPIN_LIMIT = 1u << 21;                 // polls (~1 us each with the sleep) before a wave gives up

struct FusedArgs {
    const bf16_t* qkv;   // [B,N,3,H,64]
    const bf16_t* dO;    // [B,N,H*64]
    bf16_t* dqkv;        // [B,N,3,H,64]
    const float* stats;  // [2][B*H][64 T]: -delta | LSE * log2(e); rows >= N hold 0 | 1e30 (their probabilities come out as exactly 0)
    float* part;         // [B*H][J][T][4096]: f32 partial dQ tiles in hand-off (lane-linear) layout
    unsigned* flags;     // [B*H][J][8]: tiles published by wave w of workgroup j
    unsigned* err;       // set to 1 by a wave that gave up waiting
    float* dump;         // 1 KB that lanes without an output row store to
    int B, N, H, J, T, lag;
    float scale;
};

// LDS images read by ds_read_b64_tr_b16 as 16x16x32 operands (a 16-lane group reads 4 keys x 16 columns): 8-byte granule g of key row
// `key` lives at granule g ^ f(key).  fK keeps bit 0 clear (the image is filled by 16-byte LDS-DMA pieces); fS also permutes within a
// 16-byte pair so that the eight-byte ds_write_b64 of 16 consecutive keys hit 16 different bank pairs.  Brute-forced against the bank
// model of MI355X_MICROARCH 'LDS': no conflict in either direction.
__device__ __forceinline__ int fbK(int key) { return (((key >> 2) & 1) << 1) | (((key >> 1) & 1) << 2) | (((key >> 3) & 1) << 3); }
__device__ __forceinline__ int fbS(int key) { return fbK(key) | (key & 1); }

#define FB_WAITVM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")


__device__ __forceinline__ unsigned fb_poll_issue(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 fb_ld16_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void fb_st16_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void fb_st8(void* p, uint2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// the wave's flag word (it has ONE writer) <- v, from lane 0 whatever EXEC is: exactly one vector-memory operation per call
__device__ __forceinline__ void fb_flag_publish(unsigned* p, unsigned v) {
    unsigned long long keep;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_store_dword %1, %2, off sc0 sc1\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(keep)
        : "v"(p), "v"(v)
        : "memory");
}

// Row statistics for the fused kernel, padded to whole query tiles, and the chain flags zeroed (a kernel boundary orders both before
// the main kernel).  One thread per (b, padded row, head).
__global__ __launch_bounds__(256) void attn_fused_stats_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dO,
                                                               const float* __restrict__ lse, float* __restrict__ stats, unsigned* flags,
                                                               int nflags, unsigned* err, int B, int N, int H, int Tp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (size_t)nflags) flags[idx] = 0u;
    if (idx == 0) *err = 0u;
    const size_t total = (size_t)B * Tp * H;
    if (idx >= total) return;
    const int hh = idx % H;
    const size_t bn = idx / H;
    const int n = bn % Tp, b = bn / Tp;
    const size_t so = ((size_t)b * H + hh) * Tp + n;
    const size_t BHT = (size_t)B * H * Tp;
    if (n >= N) {
        stats[so] = 0.f;
        stats[BHT + so] = 1e30f;
        return;
    }
    const int D = H * 64;
    const bf16_t* po = o + ((size_t)b * N + n) * D + hh * 64;
    const bf16_t* pd = dO + ((size_t)b * N + n) * D + hh * 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bf16x8 x = as_bf16x8(*reinterpret_cast<const uint4*>(po + 8 * i));
        bf16x8 y = as_bf16x8(*reinterpret_cast<const uint4*>(pd + 8 * i));
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)x[e] * (float)y[e];
    }
    stats[so] = -s;
    stats[BHT + so] = lse[((size_t)b * H + hh) * N + n] * LOG2E;
}

DCV_WAVES_PER_SIMD(2)
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[FB_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int BH = a.B * a.H, J = a.J, T = a.T, c = a.lag;
    int bh, j;
    if ((BH & 7) == 0 && J <= 32) {  // a chain on consecutive slots of one XCD (blocks b and b + 8 share one): hand-off reads are L2 hits
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / J) * 8 + xcd;
        j = slot % J;
    } else {
        bh = blockIdx.x / J;
        j = blockIdx.x % J;
    }
    const int b = bh / a.H, hh = bh % a.H;
    const int D = a.H * 64;
    const size_t rs = (size_t)3 * D;
    const bf16_t* Qb = a.qkv + (size_t)b * a.N * rs + hh * 64;
    const bf16_t* Kb = Qb + D;
    const bf16_t* Vb = Qb + 2 * D;
    const bf16_t* dOb = a.dO + (size_t)b * a.N * D + hh * 64;
    const int Tp = T * 64;
    // wave 0 stages LSE*log2e, wave 1 -delta, the others re-read the LSE rows into the scratch bytes (same DMA count for every wave)
    const float* statb = a.stats + (wave == 1 ? (size_t)0 : (size_t)BH * Tp) + (size_t)bh * Tp;

    const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const int rowl = 8 * wave + (lane >> 3);
    const int lc8 = ((lane & 7) ^ swz64(rowl)) * 8;
    const unsigned vq0 = (unsigned)(((size_t)rowl * rs + lc8) * 2), vo0 = (unsigned)(((size_t)rowl * D + lc8) * 2);
    const unsigned vs = (unsigned)lane * 4;
    const unsigned ring_dst = smem0 + 8 * wave * 128;
    const unsigned stat_dst = smem0 + 16384 + (wave == 0 ? 0 : (wave == 1 ? 256 : 512));
    // ring stage `slot` <- query tile tau: exactly three vector-memory operations per wave
    auto issue = [&](int tau, int slot) {
        const unsigned sb = ring_dst + slot * FB_STAGE_BYTES;
        const bf16_t* qt = Qb + (size_t)tau * 64 * rs;  // scalar bases
        const bf16_t* ot = dOb + (size_t)tau * 64 * D;
        const float* st = statb + tau * 64;
        unsigned q0 = vq0, o0 = vo0;
        if (tau * 64 + 64 > a.N) {  // partial tile: rows >= N re-read row N-1 (their statistics make P exactly 0)
            const int r0 = min(tau * 64 + rowl, a.N - 1) - tau * 64;
            q0 = (unsigned)(((size_t)r0 * rs + lc8) * 2);
            o0 = (unsigned)(((size_t)r0 * D + lc8) * 2);
        }
        glds16s(qt, q0, sb);
        glds16s(ot, o0, sb + 8192);
        glds4s(st, vs, stat_dst + slot * FB_STAGE_BYTES);
    };

    const int key0 = j * FB_KEYS;                     // first key of the workgroup
    const int nvalid = min(FB_KEYS, a.N - key0);      // >= 1
    const int nks = (nvalid + 31) >> 5;               // waves (= 32-key steps of the dQ product) with at least one key
    const int key = key0 + wave * 32 + r32;           // this lane's key
    const int kc = min(key, a.N - 1);
    const bool active = wave < nks;
    const bool wave_partial = active && (key0 + wave * 32 + 32 > a.N);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Kb + (size_t)kc * rs + 16 * ks + 8 * h));
        vf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Vb + (size_t)kc * rs + 16 * ks + 8 * h));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(kf[ks]), "v"(vf[ks]));  // hipcc's wait for these loads sits here, before any DMA

    // K image of the workgroup's keys: wave w fills its own 32 rows (4 pieces of 8 rows x 128 B; the swizzle goes to the source chunk)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int rl = 32 * wave + 8 * q4 + (lane >> 3);
        const int gk = min(key0 + rl, a.N - 1);
        const int clog = (lane & 7) ^ (fbK(rl) >> 1);
        glds16(Kb + (size_t)gk * rs + clog * 8, smem0 + FB_KT_OFF + (32 * wave + 8 * q4) * 128);
    }
    // slot k of this workgroup handles query tile (k + c j) mod T
    int tau_issue = (c * j) % T;  // tile of the next stage to issue
    auto next_tau = [&](int t) { return t + 1 == T ? 0 : t + 1; };
    issue(tau_issue, 0);
    tau_issue = next_tau(tau_issue);
    issue(T > 1 ? tau_issue : 0, 1);  // a one-tile problem re-reads its tile (never consumed)
    tau_issue = next_tau(tau_issue);

    const LaneOffs lo = lane_offs(lane);
    const float cs = a.scale * LOG2E;
    f32x16 dk[2], dv[2];
    zero_acc(dk[0]);
    zero_acc(dk[1]);
    zero_acc(dv[0]);
    zero_acc(dv[1]);

    // dS image write base of this lane's key: granule (C + h) ^ fS = C ^ (fS ^ h) for even C, and the base's bits 3..6 are free
    const int kw = 32 * wave + r32;
    const int wbase = FB_DS_OFF + kw * 128 + ((fbS(kw) ^ h) << 3);
    // dQ product (16x16x32): wave w owns query rows 16 (w >> 1) .. +15 and columns 32 (w & 1) .. +31 of the 64 x 64 tile
    const int i16 = lane & 15, g4 = lane >> 4;
    const int qblk = wave >> 1, db0 = 2 * (wave & 1);
    // per-lane read offsets of the dQ product: keys kl0 (+4 for the second half of a fragment), granule (i16 & 3) of the lane's 16 columns.
    // The other five offsets differ from these two by XOR constants (key + 4: granule bit 1 and byte 512; column block + 1: granule
    // bit 2) and are rebuilt inside phase B, where registers are plentiful, instead of living through phase A.
    const int kl0 = 8 * g4 + (i16 >> 2);
    const int offS0 = FB_DS_OFF + kl0 * 128 + (((4 * qblk + (i16 & 3)) ^ fbS(kl0)) << 3);
    const int offK0 = FB_KT_OFF + kl0 * 128 + (((4 * db0 + (i16 & 3)) ^ fbK(kl0)) << 3);
    const size_t lane_part = ((size_t)(wave * 2) * 64 + lane) * 4;  // floats: [wave][db][lane] float4
    float* const my_part = a.part + ((size_t)bh * J + j) * T * 4096;
    unsigned* const my_flag = a.flags + ((size_t)bh * J + j) * 8 + wave;
    const int jp = (j < J - 1) ? j + 1 : 0;  // the workgroup this one receives partial sums from (when the slot has a predecessor)
    const float* const pred_part = a.part + ((size_t)bh * J + jp) * T * 4096;
    const unsigned* const pred_flag = a.flags + ((size_t)bh * J + jp) * 8 + wave;
    const int wrap0 = T - c * (J - 1);  // workgroup J-1: first slot whose tile workgroup 0 has already handled

    // ---- phase A of a slot: S^T, dP^T, P, dS for this wave's 32 keys x the tile's 64 queries; dV^T, dK^T; dS -> LDS image `buf` ----
    auto phaseA = [&](int slot, int buf, auto QB) {
        constexpr int qb = decltype(QB)::value;
        const int so = slot * FB_STAGE_BYTES;
        int ro[4], co[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ro[ks] = lo.rows[ks] + so;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            co[dt][0] = lo.cols[dt][0] + so;
            co[dt][1] = lo.cols[dt][1] + so;
        }
        const int sto = so + 16384 + 16 * h;  // this lane-half's 4 consecutive query rows inside an 8-row group
        const int wb = wbase + buf * FB_DS_BYTES;
        {
            f32x16 s, dp;
            f32x4 l4[4];
            zero_acc(s);
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // the dP accumulator starts at -delta of its query rows
                l4[g] = *reinterpret_cast<const f32x4*>(smem + sto + (32 * qb + 8 * g) * 4);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(smem + sto + 256 + (32 * qb + 8 * g) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dp[4 * g + e] = d4[e];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(as_bf16x8(lds_read128(smem, ro[ks] + qb * 4096)), kf[ks], s);
                dp = mfma32(as_bf16x8(lds_read128(smem, ro[ks] + 8192 + qb * 4096)), vf[ks], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(s[4 * g + e] * cs - l4[g][e]);
                    s[4 * g + e] = p;
                    dp[4 * g + e] = p * dp[4 * g + e];  // dS = P (dP - delta)
                }
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 pf = acc_to_frag(s, ss);
                bf16x8 dsf = acc_to_frag(dp, ss);
                const int cc = qb * 4096 + ss * 2048;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma32(join4(lds_tr_read(smem, co[dt][0] + 8192 + cc), lds_tr_read(smem, co[dt][1] + 8192 + cc)), pf, dv[dt]);
                    dk[dt] = mfma32(join4(lds_tr_read(smem, co[dt][0] + cc), lds_tr_read(smem, co[dt][1] + cc)), dsf, dk[dt]);
                }
                U128 u;
                u.h = dsf;
                // elements 0..3: query rows 32 qb + 16 ss + 4 h + (0..3) -> granule 8 qb + 4 ss + h; elements 4..7: 8 rows further
                *reinterpret_cast<uint2*>(smem + (wb ^ ((8 * qb + 4 * ss) << 3))) = u.d[0];
                *reinterpret_cast<uint2*>(smem + (wb ^ ((8 * qb + 4 * ss + 2) << 3))) = u.d[1];
            }
        }
    };
    // ---- phase B of a slot (one iteration later): dQ tile = dS K over the workgroup's keys, + predecessor, -> successor or output ----
    auto phaseB_mma = [&](int buf, f32x4& dq0, f32x4& dq1) {
        const int bo = buf * FB_DS_BYTES;
        int oS = offS0, oK = offK0;
        asm volatile("" : "+v"(oS), "+v"(oK));  // opaque: keeps hipcc from hoisting the derived offsets out of the tile loop
        const int offS[2] = {oS, oS ^ (16 | 512)};
        const int offK[2][2] = {{oK, oK ^ (16 | 512)}, {oK ^ 32, oK ^ (32 | 16 | 512)}};
        auto step = [&](int ko) {
            const bf16x8 bf = join4(lds_tr_read(smem, offS[0] + bo + ko), lds_tr_read(smem, offS[1] + bo + ko));
            const bf16x8 a0 = join4(lds_tr_read(smem, offK[0][0] + ko), lds_tr_read(smem, offK[0][1] + ko));
            const bf16x8 a1 = join4(lds_tr_read(smem, offK[1][0] + ko), lds_tr_read(smem, offK[1][1] + ko));
            dq0 = mfma16(a0, bf, dq0);  // rows: head-dim columns 16 db + 4 g4 + (0..3); column: query 16 qblk + i16
            dq1 = mfma16(a1, bf, dq1);
        };
        if (nks == FB_WAVES) {
#pragma unroll
            for (int ks = 0; ks < FB_WAVES; ++ks) step(ks * 4096);
        } else {
            for (int ks = 0; ks < nks; ++ks) step(ks * 4096);
        }
    };

    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    int tau = (c * j) % T;  // tile of the current slot
    int tau_prev = 0;
    bool gave_up = false;
    // hand-off role of a slot (decided when its poll is issued, used one iteration later when its phase B runs)
    bool has_pred = false, is_last = false;
    unsigned need = 0;
    auto role_of = [&](int k, bool& hp, bool& il, unsigned& nd) {
        int kp;
        if (j < J - 1) { hp = k >= c; kp = k - c; }
        else { hp = (J > 1) && k >= wrap0; kp = k - wrap0; }
        il = (j >= 1) ? (k + c >= T) : (k >= c * (J - 1));
        nd = hp ? (unsigned)(kp + 1) : 0u;
    };
    auto zero_invalid_k_rows = [&]() {
        // keys >= N of this wave are clamped duplicates of key N-1 (their dK / dV are never stored); their dS must not reach dQ: their rows
        // of the K image become zero.  This wave's own DMA pieces of the image have landed, the image is first read after the next barrier.
        if (wave_partial && key >= a.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(smem + FB_KT_OFF + (32 * wave + r32) * 128 + 64 * h + 16 * i) = make_uint4(0u, 0u, 0u, 0u);
        }
    };

    // Per iteration and wave, in issue order:  P (1)  D (3)  L (2)  S (2)  A (1);  iteration 0 has only D.  Waits (vmcnt counts in issue order):
    //   top of iteration it >= 3 : vmcnt(14) -> the ring stage issued two iterations ago has landed (younger: that iteration's L, S, A and
    //                              the whole previous iteration); it == 2: vmcnt(9); it <= 1: vmcnt(3);
    //   behind phase A           : vmcnt(3)  -> the poll has landed (younger: D);
    //   before the add           : vmcnt(0)  -> L has landed; the previous iteration's S are older, so A may publish their slot.
    // Measured alternatives (profiles/r04_x1_*): L issued before phase A with the poll one iteration ahead, and the publication moved to the
    // middle of the next phase A, both made the chains stall MORE (the poll then reads an older flag; the mid-iteration wait for the stores
    // stalls): 1260-1500 us against 1107 us for this order at the headline shape.
    for (int it = 0; it <= T; ++it) {
        if (it >= 3) FB_WAITVM(14);
        else if (it == 2) FB_WAITVM(9);
        else FB_WAITVM(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's dS writes of the previous slot
        __builtin_amdgcn_s_barrier();
        const int k = it - 1;  // slot whose phase B runs in this iteration
        unsigned pv = 0;
        if (it >= 1) {
            role_of(k, has_pred, is_last, need);
            pv = fb_poll_issue(has_pred ? pred_flag : my_flag);  // P
        }
        // D: beyond the last tile a copy nobody reads goes into the free buffer — every iteration issues the same operations
        issue(it + 2 < T ? tau_issue : (it < T ? tau : tau_prev), (it + 2) % FB_STAGES);
        tau_issue = next_tau(tau_issue);
        if (it == 0) zero_invalid_k_rows();
        if (it < T && active) {  // ONE block for both halves: hipcc overlaps the second half's loads and MFMAs with the first half's vector work
            phaseA(it % FB_STAGES, it & 1, Q0{});
            phaseA(it % FB_STAGES, it & 1, Q1{});
        }
        if (it >= 1) {
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(pv)::"memory");  // P landed (younger: D).  ONE statement on every path: hipcc must not copy pv before it
            unsigned got = __builtin_amdgcn_readfirstlane(pv);
            if (got < need && !gave_up) {  // the predecessor is late: slow spin, bounded
                unsigned spins = 0;
                do {
                    __builtin_amdgcn_s_sleep(8);
                    unsigned v;
                    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(pred_flag) : "memory");
                    got = __builtin_amdgcn_readfirstlane(v);
                } while (got < need && ++spins < FB_SPIN_LIMIT);
                if (got < need) {
                    gave_up = true;
                    if (lane == 0) *a.err = 1u;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            const float* src = (has_pred ? pred_part : my_part) + (size_t)tau_prev * 4096 + lane_part;
            f32x4 l0 = fb_ld16_sc1(src), l1 = fb_ld16_sc1(src + 256);  // L
            f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
            phaseB_mma(k & 1, dq0, dq1);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(l0), "+v"(l1)::"memory");  // L landed (nothing younger)
            const float m = has_pred ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dq0[e] = fmaf(m, has_pred ? l0[e] : 0.f, dq0[e]);
                dq1[e] = fmaf(m, has_pred ? l1[e] : 0.f, dq1[e]);
            }
            if (is_last) {  // S: the finished rows, bf16, into dqkv slot 0
                const int q = tau_prev * 64 + 16 * qblk + i16;
                char* dst = (q < a.N) ? (char*)(a.dqkv + ((size_t)b * a.N + q) * rs + hh * 64 + 16 * db0 + 4 * g4) : (char*)a.dump + lane * 8;
                const uint2 v0 = pack4_bf16(dq0[0] * a.scale, dq0[1] * a.scale, dq0[2] * a.scale, dq0[3] * a.scale);
                const uint2 v1 = pack4_bf16(dq1[0] * a.scale, dq1[1] * a.scale, dq1[2] * a.scale, dq1[3] * a.scale);
                fb_st8(dst, v0);
                fb_st8(dst + ((q < a.N) ? 32 : 512), v1);
            } else {  // S: the running sum for the next workgroup of the chain
                float* dstp = my_part + (size_t)tau_prev * 4096 + lane_part;
                fb_st16_sc1(dstp, dq0);
                fb_st16_sc1(dstp + 256, dq1);
            }
            // A: the stores of the PREVIOUS iteration (slot k - 1) are complete (they are older than L, which has landed): flag = slots published
            fb_flag_publish(my_flag, it >= 2 ? (unsigned)(it - 1) : 0u);
        }
        tau_prev = tau;
        tau = next_tau(tau);
    }
    // the last slot's stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fb_flag_publish(my_flag, (unsigned)T);

    if (active && key < a.N) {
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

struct FusedLayout {
    int J, T, lag;
    size_t stats_floats, nflags, part_floats;
    size_t off_flags, off_err, off_dump, off_part, bytes;
};
inline FusedLayout fused_layout(int B, int N, int H) {
    FusedLayout L;
    L.T = (N + 63) / 64;
    L.J = (N + FB_KEYS - 1) / FB_KEYS;
    int c = L.T / L.J;
    L.lag = c < 1 ? 1 : (c > 3 ? 3 : c);
    const size_t BH = (size_t)B * H;
    L.stats_floats = 2 * BH * L.T * 64;
    L.nflags = BH * L.J * 8;
    L.part_floats = BH * L.J * L.T * 4096;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.off_flags = up(L.stats_floats * 4);
    L.off_err = up(L.off_flags + L.nflags * 4);
    L.off_dump = L.off_err + 256;
    L.off_part = L.off_dump + 1024;
    L.bytes = L.off_part + L.part_floats * 4;
    return L;
}

}  // namespace

extern "C" size_t dcv_attn_bwd_fused_ws_bytes(int B, int N, int H) {
    if (B <= 0 || N <= 0 || H <= 0) return 0;
    return fused_layout(B, N, H).bytes;
}

extern "C" int dcv_attn_bwd_fused(const void* qkv, const void* o, const void* dO, const float* lse, void* ws, void* dqkv, int B, int N,
                                  int H, int head_dim, float scale, void* stream) {
    int rc = attn_check(qkv, B, N, H, head_dim);
    if (rc) return rc;
    if (!o || !dO || !lse || !ws || !dqkv) return DCV_ERR_NULL;
    if (((uintptr_t)ws & 255) || ((uintptr_t)dqkv & 15) || ((uintptr_t)dO & 15) || ((uintptr_t)o & 15)) return DCV_ERR_ALIGN;
    const FusedLayout L = fused_layout(B, N, H);
    if (L.J > 256) return DCV_ERR_UNSUPPORTED;  // a chain must fit on the chip
    char* w = (char*)ws;
    FusedArgs a{(const bf16_t*)qkv, (const bf16_t*)dO, (bf16_t*)dqkv, (const float*)w, (float*)(w + L.off_part),
                (unsigned*)(w + L.off_flags), (unsigned*)(w + L.off_err), (float*)(w + L.off_dump), B, N, H, L.J, L.T, L.lag, scale};
    const size_t total = (size_t)B * L.T * 64 * H;
    const size_t nthreads = total > L.nflags ? total : L.nflags;
    hipLaunchKernelGGL(attn_fused_stats_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)o, (const bf16_t*)dO, lse, (float*)w, a.flags, (int)L.nflags, a.err, B, N, H, L.T * 64);
    DCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3((unsigned)(B * H * L.J)), dim3(512), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// the error word of a workspace (1 = some wave gave up waiting for its predecessor: results undefined); a device pointer the caller may copy
extern "C" const void* dcv_attn_bwd_fused_err_ptr(const void* ws, int B, int N, int H) {
    if (!ws || B <= 0 || N <= 0 || H <= 0) return nullptr;
    return (const char*)ws + fused_layout(B, N, H).off_err;
}

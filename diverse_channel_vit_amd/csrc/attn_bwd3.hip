// Attention backward, dK / dV, third form (head_dim 64, pre-scaled q): ONE wave per SIMD owning the whole 512-entry register file,
// 64 keys per wave (cdna guide, appendix B "Attention backward": one workgroup = 4 waves = 256 keys of one (batch, head); each wave keeps
// dK^T and dV^T of its 64 keys in accumulator registers while the workgroup sweeps the 32-row query slices).
//
// Why (round 5): the second form (attn_bwd.hip, 32 keys per wave, two waves per SIMD) runs at matrix + vector time (MFMA busy 0.49, co-execution
// 0.13): every 32 x 32 score block re-reads its Q / dO fragments from LDS (32 LDS instructions per 16 MFMAs and wave, eight waves per CU).  With 64
// keys per wave one set of Q / dO fragments (row reads for S and dP, transposed reads for dV^T and dK^T) feeds TWO key blocks: half the LDS reads
// and half the LDS-DMA pieces per score, and the score / softmax-gradient arithmetic of one 32 x 32 block sits in the MFMA shadow of its
// neighbours (software pipeline below) instead of depending on a second wave's phase.
#include "attn_common.hpp"

namespace {

constexpr int K3_STAGES = 4, K3_STAGE_BYTES = 16384 + 1024;  // Q tile | dO tile | -LSE log2e [64] | -delta[64] | 512 B scratch
constexpr int K3_DMA_PER_WAVE = 5;
constexpr int K3_KEYS = 256;  // keys per workgroup

#ifndef DCV_K3_PIPE
#define DCV_K3_PIPE 1
#endif

__attribute__((amdgpu_waves_per_eu(1, 1)))
__global__ __launch_bounds__(256) void attn_bwd_dkdv3_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sQO[K3_STAGES * K3_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = (a.N + K3_KEYS - 1) / K3_KEYS;
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
    // waves 0/2 fetch the -LSE log2e rows, waves 1/3 the -delta rows; waves 2,3 write theirs to the scratch slot (uniform DMA count per wave)
    const float* statb = a.delta + ((wave & 1) ? 0 : (size_t)a.B * a.H * a.N) + ((size_t)b * a.H + hh) * a.N;
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
        const unsigned sb = smem_base + slot * K3_STAGE_BYTES;
        const bf16_t* qt = Qb + (size_t)t * 64 * rs;  // scalar bases
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
        glds4s(st, s0, stat_dst + slot * K3_STAGE_BYTES);
    };

    const int key0 = kt * K3_KEYS + wave * 64;  // this wave's first key
    const bool active = key0 < a.N;             // a wave without a single valid key only keeps the ring going
    bf16x8 kf[2][4], vf[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int kc = min(key0 + 32 * kb + r32, a.N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[kb][ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Kb + (size_t)kc * rs + 16 * ks + 8 * h));
            vf[kb][ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Vb + (size_t)kc * rs + 16 * ks + 8 * h));
        }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(kf[kb][ks]), "v"(vf[kb][ks]));  // hipcc's wait for these loads sits here
    for (int st = 0; st < K3_STAGES - 1; ++st)
        if (st < nt) issue(st, st);
    const LaneOffs lo = lane_offs(lane);
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            zero_acc(dk[kb][dt]);
            zero_acc(dv[kb][dt]);
        }

    auto tile = [&](auto MASKED, auto COMPUTE, int t, int slot) {
        const int rem = nt - 1 - t;  // younger stages in flight: min(rem, 2)
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K3_DMA_PER_WAVE) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K3_DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage t landed for everyone; everyone is done with tile t-1 -> its buffer is free
        if (t + 3 < nt) issue(t + 3, (slot + 3) & 3);
        if constexpr (!decltype(COMPUTE)::value) return;
        const int so = slot * K3_STAGE_BYTES;
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
            f32x16 s[2], dp[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sQO + sto + (32 * qb + 8 * g) * 4);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sQO + sto + 256 + (32 * qb + 8 * g) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[0][4 * g + e] = l4[e];
                    s[1][4 * g + e] = l4[e];
                    dp[0][4 * g + e] = d4[e];
                    dp[1][4 * g + e] = d4[e];
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qf = as_bf16x8(lds_read128(sQO, ro[ks] + qb * 4096));
                const bf16x8 dof = as_bf16x8(lds_read128(sQO, ro[ks] + 8192 + qb * 4096));
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[kb] = mfma32(qf, kf[kb][ks], s[kb]);
                    dp[kb] = mfma32(dof, vf[kb][ks], dp[kb]);
                }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(s[kb][r]);
                    if constexpr (decltype(MASKED)::value) {
                        if (t * 64 + 32 * qb + acc_row(r, h) >= a.Nq) p = 0.f;  // query row does not exist
                    }
                    s[kb][r] = p;
                    dp[kb][r] = p * dp[kb][r];  // dS = P * (dP - delta)
                }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int cc = qb * 4096 + ss * 2048;
                bf16x8 pf[2], dsf[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    pf[kb] = acc_to_frag(s[kb], ss);
                    dsf[kb] = acc_to_frag(dp[kb], ss);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 doT = join4(lds_tr_read(sQO, co[dt][0] + 8192 + cc), lds_tr_read(sQO, co[dt][1] + 8192 + cc));
                    const bf16x8 qT = join4(lds_tr_read(sQO, co[dt][0] + cc), lds_tr_read(sQO, co[dt][1] + cc));
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        dv[kb][dt] = mfma32(doT, pf[kb], dv[kb][dt]);
                        dk[kb][dt] = mfma32(qT, dsf[kb], dk[kb][dt]);
                    }
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

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = key0 + 32 * kb + r32;
        if (key < a.N) {
            bf16_t* dkp = a.dqkv + ((size_t)b * a.N + key) * rs + D + hh * 64;
            bf16_t* dvp = dkp + D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float ks_ = 1.f / LOG2E;
                    uint2 v1 = pack4_bf16(dk[kb][dt][4 * g] * ks_, dk[kb][dt][4 * g + 1] * ks_, dk[kb][dt][4 * g + 2] * ks_, dk[kb][dt][4 * g + 3] * ks_);
                    uint2 v2 = pack4_bf16(dv[kb][dt][4 * g], dv[kb][dt][4 * g + 1], dv[kb][dt][4 * g + 2], dv[kb][dt][4 * g + 3]);
                    *reinterpret_cast<uint2*>(dkp + 32 * dt + 8 * g + 4 * h) = v1;
                    *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g + 4 * h) = v2;
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The software-pipelined form.  A UNIT is one 32-query x 32-key block of one wave: (tile t, slice qb, key block kb), four units per 64-query
// tile in the order (0,0) (0,1) (1,0) (1,1).  Step u of the stream issues 16 MFMAs — S' and dP' of unit u+1 (8), then dV^T and dK^T of unit
// u-1 (8) — and, in the 16 gaps between them, the vector work of unit u (16 x {v_exp, v_mul}, 16 v_cvt_pk) plus one or two LDS fragment reads
// (<= 5 single-issue fillers per gap, at most one of them a v_exp: MI355X_MICROARCH, cycle constants).  MFMAs are inline asm so that their
// accumulators sit where the design needs them — dK^T / dV^T (128 registers) and the K / V fragments (64) in the accumulator file, S' / dP' in
// arch VGPRs where the vector instructions read them — and every gap is fenced with sched_barrier(0): the order below IS the instruction stream.
// Hazards hipcc does not see inside asm (cdna guide 5.7): an S' / dP' chain is finished 8 MFMAs (256 cycles) before its first vector read; P / dS
// fragments are written one step before the MFMAs that read them; the accumulators are read after the loop behind s_nop.
constexpr int P3_STAGES = 5, P3_STAGE_BYTES = 16384 + 1024, P3_DMA = 5;

__device__ __forceinline__ void mfma_vv(f32x16& d, const bf16x8& a, const bf16x8& b) {  // d += a b   (d in arch VGPRs, b = K / V fragment in AGPRs)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_vc(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c) {  // d = a b + c
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma_aa(f32x16& d, const bf16x8& a, const bf16x8& b) {  // d += a b   (d in AGPRs)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
#ifdef DCV_K3_STAMP  // diagnostic build only (tools/attn3_stamps.py): cycle stamps of wave 0 of every workgroup; never in the product library
__device__ unsigned long long k3_stamps[8192 * 8];
#define K3_NOW() __builtin_amdgcn_s_memtime()
#endif
#define P3_FENCE() __builtin_amdgcn_sched_barrier(0)
#define P3_PIN(x) asm volatile("" : "+v"(x))
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {  // one v_cvt_pk_bf16_f32
    bf16x2_t v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, v);
}

__attribute__((amdgpu_waves_per_eu(1, 1)))
__global__ __launch_bounds__(256) void attn_bwd_dkdv3p_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sQO[P3_STAGES * P3_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = (a.N + K3_KEYS - 1) / K3_KEYS;
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
    const float* statb = a.delta + ((wave & 1) ? 0 : (size_t)a.B * a.H * a.N) + ((size_t)b * a.H + hh) * a.N;
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
        const unsigned sb = smem_base + slot * P3_STAGE_BYTES;
        const bf16_t* qt = Qb + (size_t)t * 64 * rs;  // scalar bases
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
        glds4s(st, s0, stat_dst + slot * P3_STAGE_BYTES);
    };
    // top of tile t: stage t + 1 has landed (the step that reads it is two steps away), everyone is done with tile t - 1 -> its slot takes stage t + S - 1
#ifdef DCV_K3_STAMP
    unsigned long long st_wait = 0, st_issue = 0;
    const unsigned long long st_entry = K3_NOW();
#endif
    auto top = [&](int t, int slot) {
#ifdef DCV_K3_STAMP
        const unsigned long long s0 = K3_NOW();
#endif
        const int rem = nt - 1 - t;
        if (rem >= P3_STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P3_STAGES - 3) * P3_DMA) : "memory");
        else if (rem == P3_STAGES - 3 && P3_STAGES >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P3_STAGES - 4) * P3_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef DCV_K3_STAMP
        const unsigned long long s1 = K3_NOW();
#endif
        if (t + P3_STAGES - 1 < nt) issue(t + P3_STAGES - 1, slot == 0 ? P3_STAGES - 1 : slot - 1);
#ifdef DCV_K3_STAMP
        const unsigned long long s2 = K3_NOW();
        st_wait += s1 - s0;
        st_issue += s2 - s1;
#endif
    };

    const int key0 = kt * K3_KEYS + wave * 64;  // this wave's first key
    const bool active = key0 < a.N;             // a wave without a single valid key only keeps the ring going
    for (int st = 0; st < P3_STAGES - 1; ++st)
        if (st < nt) issue(st, st);
    if (!active) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int t = 0; t < nt; ++t) {
            top(t, slot);
            slot = slot == P3_STAGES - 1 ? 0 : slot + 1;
        }
        return;
    }
    bf16x8 kf[2][4], vf[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int kc = min(key0 + 32 * kb + r32, a.N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[kb][ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Kb + (size_t)kc * rs + 16 * ks + 8 * h));
            vf[kb][ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Vb + (size_t)kc * rs + 16 * ks + 8 * h));
        }
    }
    // hipcc's wait for these loads sits HERE: left to the first use it lands inside the tile loop, as s_waitcnt vmcnt(7 .. 0) in front of the first
    // MFMAs of every tile — and vmcnt(0) there drains the LDS-DMA ring (measured: 75 cycles per MFMA instead of 35)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"a"(kf[kb][ks]), "a"(vf[kb][ks]));
    const LaneOffs lo = lane_offs(lane);
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            zero_acc(dk[kb][dt]);
            zero_acc(dv[kb][dt]);
        }

    // register sets: X[0] = even units (kb 0), X[1] = odd units (kb 1)
    f32x16 sx[2], dpx[2];
    bf16x8 pf[2][2], dsf[2][2];  // [set][ss]
    bf16x8 rq[4], rdo[4];        // row fragments of the current slice (S', dP')
    f32x16 stl, std_;            // -LSE log2e / -delta rows of the current slice, accumulator layout
    bf16x8 tr0[2][2][2];         // [slice parity][dt][dO | Q]: transposed fragments, k-step ss = 0
    bf16x8 tr1[2][2];            // the same for ss = 1 (one slice at a time)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pf[i][j] = dsf[i][j] = tr1[i][j] = as_bf16x8(make_uint4(0, 0, 0, 0));
            tr0[1][i][j] = as_bf16x8(make_uint4(0, 0, 0, 0));
        }
    const int sto0 = 16384 + 16 * h;  // this lane-half's 4 consecutive query rows inside an 8-row group
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
    auto step = [&](auto Jc, auto MASKED, int t, int so, int so_next) {
        constexpr int J = decltype(Jc)::value;
        constexpr int X = J & 1, Y = X ^ 1;      // this unit's register set / the neighbours'
        constexpr int qb = J >> 1;               // this unit's slice
        constexpr int qbn = ((J + 1) >> 1) & 1;  // slice of unit u + 1 (rows already in rq / rdo)
        constexpr int qbp = ((J + 3) >> 1) & 1;  // slice of unit u - 1
        const bf16x8 pfy[2] = {pf[Y][0], pf[Y][1]}, dsfy[2] = {dsf[Y][0], dsf[Y][1]};
        const int lim = a.Nq - t * 64 - 32 * qb - 4 * h;  // MASKED: accumulator row index (r & 3) + 8 (r >> 2) >= lim does not exist
        bf16x4 half[2];
        float pv[16], dsv[16];
        unsigned pw[8], dw[8];  // packed P / dS pairs: word i of k-step ss = i >> 2
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            // ---- the MFMA of gap g
            if (g < 8) {
                const int ks = g >> 1;
                if ((g & 1) == 0) {
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
            // ---- ... and the LDS reads of the steps ahead
            if (g < 8) {
                if ((J & 1) == 0) load_tr(tr0[qb], so, qb, 0, g, half);  // even step: k-step 0 of this slice's transposed fragments (first used next step)
                else load_tr(tr1, so, qb, 1, g, half);                   // odd step: k-step 1 (used from MFMA 12 of this step)
            } else if ((J & 1) == 0) {                                   // even step, after the last MFMA that read rq / rdo / stl / std_: the next slice's
                const int i = 2 * (g - 8);
                if (J == 0) {
                    load_rows(so, 1, i);
                    load_rows(so, 1, i + 1);
                } else {
                    load_rows(so_next, 0, i);
                    load_rows(so_next, 0, i + 1);
                }
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

    // prologue: stage 0, the rows of slice (0, 0), S' and dP' of unit 0
    {
        const int inflight = min(nt, P3_STAGES - 1) - 1;  // stages younger than stage 0
        if (inflight >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * P3_DMA) : "memory");
        else if (inflight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P3_DMA) : "memory");
        else if (inflight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * P3_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) load_rows(0, 0, i);
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
    auto tile = [&](auto MASKED, int t, int slot) {
        top(t, slot);
        const int so = slot * P3_STAGE_BYTES;
        const int so_next = (slot == P3_STAGES - 1 ? 0 : slot + 1) * P3_STAGE_BYTES;
        P3_FENCE();
        step(J0{}, MASKED, t, so, so_next);
        step(J1{}, MASKED, t, so, so_next);
        step(J2{}, MASKED, t, so, so_next);
        step(J3{}, MASKED, t, so, so_next);
    };
    const int nfull = a.Nq / 64;
    int slot = 0;
#ifdef DCV_K3_STAMP
    const unsigned long long st_loop0 = K3_NOW();
#endif
    for (int t = 0; t < nfull; ++t) {
        tile(No{}, t, slot);
        slot = slot == P3_STAGES - 1 ? 0 : slot + 1;
    }
    if (nfull < nt) tile(Yes{}, nfull, slot);
#ifdef DCV_K3_STAMP
    const unsigned long long st_loop1 = K3_NOW();
#endif
    // epilogue: dV^T / dK^T of the last unit (set 1, slice parity 1)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ss = i >> 2, dt = (i >> 1) & 1;
        const bf16x8& aT = ss == 0 ? tr0[1][dt][i & 1] : tr1[dt][i & 1];
        if ((i & 1) == 0) mfma_aa(dv[1][dt], aT, pf[1][ss]);
        else mfma_aa(dk[1][dt], aT, dsf[1][ss]);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    P3_FENCE();

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = key0 + 32 * kb + r32;
        if (key < a.N) {
            bf16_t* dkp = a.dqkv + ((size_t)b * a.N + key) * rs + D + hh * 64;
            bf16_t* dvp = dkp + D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float ks_ = 1.f / LOG2E;
                    uint2 v1 = pack4_bf16(dk[kb][dt][4 * g] * ks_, dk[kb][dt][4 * g + 1] * ks_, dk[kb][dt][4 * g + 2] * ks_, dk[kb][dt][4 * g + 3] * ks_);
                    uint2 v2 = pack4_bf16(dv[kb][dt][4 * g], dv[kb][dt][4 * g + 1], dv[kb][dt][4 * g + 2], dv[kb][dt][4 * g + 3]);
                    *reinterpret_cast<uint2*>(dkp + 32 * dt + 8 * g + 4 * h) = v1;
                    *reinterpret_cast<uint2*>(dvp + 32 * dt + 8 * g + 4 * h) = v2;
                }
        }
    }
#ifdef DCV_K3_STAMP
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* o = k3_stamps + (size_t)blockIdx.x * 8;
        o[0] = st_entry; o[1] = st_loop0; o[2] = st_loop1; o[3] = K3_NOW(); o[4] = st_wait; o[5] = st_issue; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = nt;
    }
#endif
}

}  // namespace
#ifdef DCV_K3_STAMP
extern "C" int dcv_k3_stamps(void* host_dst, size_t bytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(k3_stamps), bytes < sizeof(k3_stamps) ? bytes : sizeof(k3_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int dcv_attn_bwd_dkdv_rows_ps3(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq,
                                          int H, int head_dim, float scale, void* stream) {
    int rc = attn_check(qkv, B, N, H, head_dim);
    if (rc) return rc;
    if (!dO || !lse || !ws || !dqkv) return DCV_ERR_NULL;
    if (Nq < 1 || Nq > N) return DCV_ERR_SHAPE;
    AttnArgs a{(const bf16_t*)qkv, nullptr, (const bf16_t*)dO, (float*)lse, (float*)ws, (bf16_t*)dqkv, B, N, H, scale, Nq};
    const dim3 grid(B * H * ((N + K3_KEYS - 1) / K3_KEYS));
    if (DCV_K3_PIPE) hipLaunchKernelGGL(attn_bwd_dkdv3p_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkdv3_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

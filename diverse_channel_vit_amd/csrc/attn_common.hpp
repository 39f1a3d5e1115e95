// Shared pieces of the attention kernels (attn.hip: forward; attn_bwd.hip: delta, dQ, dK/dV).
#pragma once
#include <type_traits>
#include "dcv_common.hpp"
#include "../../include/dcv.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// amdgpu_waves_per_eu(n, n) tells hipcc how many waves per SIMD the kernel is launched for, i.e. how many VGPRs it MAY use
// (512 / n).  Without it the scheduler aims at a lower register count than the occupancy needs and issues every LDS fragment
// read just before the MFMA that consumes it (ds_read; s_waitcnt lgkmcnt(0); v_mfma — the full LDS latency exposed per MFMA).
#define DCV_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#ifndef DCV_WPE_FWD
#define DCV_WPE_FWD 3
#endif
#ifndef DCV_WPE_DQ
#define DCV_WPE_DQ 3
#endif
#ifndef DCV_WPE_DKDV
#define DCV_WPE_DKDV 2
#endif

struct AttnArgs {
    const bf16_t* qkv;  // [B,N,3,H,64]
    bf16_t* o;          // [B,N,H*64]           (fwd out / bwd in)
    const bf16_t* dO;   // [B,N,H*64]
    float* lse;         // [B,H,N]
    float* delta;       // [B,H,N]
    bf16_t* dqkv;       // [B,N,3,H,64]
    int B, N, H;
    float scale;
    int Nq;  // query rows [0, Nq) of every (batch, head) are processed (Nq = N: all; Nq = 1: the CLS row of the last block)
    int key_lo = 0, key_hi = 0;  // dK / dV kernels launched for a key range only: keys [key_lo, key_hi) (0, 0: all)
};

// blockIdx -> (batch-head, 128-query tile) of the forward and dQ kernels.  An XCD gets all tiles of a batch-head (its K / V stay in that XCD's L2).
// DCV_ATTN_LIGHT_LAST (variant builds): when the last query tile is ragged (N = 1569: 33 of 128 rows, two of its four waves idle), those light workgroups
// are numbered AFTER all full ones, so that the full tiles fill whole rounds of resident workgroups and the light ones share the tail round.  Measured: no
// change (forward 300 -> 306 us, dQ 385 -> 380, inside the run-to-run spread; profiles/r05_x8_* (c)); off in the product.
#ifndef DCV_ATTN_LIGHT_LAST
#define DCV_ATTN_LIGHT_LAST 0
#endif
__device__ __forceinline__ void attn_block_to_tile(int bid, int BH, int nqt, int n_rows, int& bh, int& qt) {
    if ((BH & 7) != 0) {
        bh = bid / nqt;
        qt = bid % nqt;
        return;
    }
    const bool light = DCV_ATTN_LIGHT_LAST && (n_rows & 127) != 0 && (n_rows & 127) <= 64 && nqt > 1;
    const int nf = light ? nqt - 1 : nqt;
    if (bid < BH * nf) {
        const int xcd = bid & 7, slot = bid >> 3;
        bh = (slot / nf) * 8 + xcd;
        qt = slot % nf;
    } else {
        bh = bid - BH * nf;  // (bh & 7 = the XCD of its full tiles)
        qt = nqt - 1;
    }
}

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

// max(x of lane l, x of lane l ^ 32) by one v_permlane32_swap (gfx950) instead of __shfl_xor's ds_bpermute_b32: no LDS round trip and no s_waitcnt
// lgkmcnt(0) — which in the forward loop also waited for every fragment read in flight — in the middle of a tile.  After the swap of x with itself one
// result holds the lower half's values in both halves, the other the upper half's.
__device__ __forceinline__ float max_halves(float x) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float m;  // asm: fmaxf on values that arrive as integers makes hipcc canonicalise both operands first (two v_max x, x)
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
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
// Tile rings: stages of (8 KB K + 8 KB V) or (8 KB Q + 8 KB dO), filled by LDS-DMA two or three tiles ahead.
// A register-staged single-tile prefetch left every key tile waiting ~1.5 us for its loads (measured: 3400 cycles
// per tile for 512 cycles of MFMA).  Each wave issues 4 DMA instructions per stage: rows [16w,16w+16) of either tile.
constexpr int KV_STAGE_BYTES = 16384;  // one ring stage: a 64-row K (or Q) tile + a 64-row V (or dO) tile
constexpr int FWD_WAVES = 4, FWD_QTILE = 32 * FWD_WAVES;  // query rows per workgroup: K/V re-reads scale with 1/FWD_QTILE
constexpr int KV_DMA_PER_WAVE = 16 / FWD_WAVES;            // DMA instructions per stage per wave (8 rows x 128 B each; K: 8, V: 8)


inline int attn_check(const void* qkv, int B, int N, int H, int hd) {
    if (!qkv) return DCV_ERR_NULL;
    if (B <= 0 || N <= 0 || H <= 0) return DCV_ERR_SHAPE;
    if (hd != 64) return DCV_ERR_UNSUPPORTED;
    if ((uintptr_t)qkv & 15) return DCV_ERR_ALIGN;
    return DCV_OK;
}

}  // namespace

// attn_bwd.hip: the second dK / dV form (pre-scaled q) for keys [key_lo, N) only; key_lo a multiple of 128
__attribute__((visibility("hidden"))) int dcv_dkdv2_range(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq, int H, float scale, int key_lo,
                    hipStream_t stream);
// attn_bwd3.hip: the third dK / dV form (pre-scaled q): persistent, one wave per SIMD; all keys (the remainder of < 129 keys through dcv_dkdv2_range)
__attribute__((visibility("hidden"))) int dcv_dkdv3_launch(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq,
                                                           int H, float scale, hipStream_t stream);
// the dQ counterparts (attn_bwd.hip for query rows [row_lo, N), row_lo a multiple of 128; attn_bwd3q.hip for all rows, Nq == N)
__attribute__((visibility("hidden"))) int dcv_dq2_range(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H, float scale,
                                                        int row_lo, hipStream_t stream);
__attribute__((visibility("hidden"))) int dcv_dq3_launch(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H,
                                                         float scale, hipStream_t stream);
#ifndef DCV_DQ_FORM
#define DCV_DQ_FORM 2  // 3 (variant builds only): the one-wave-per-SIMD dQ kernel of attn_bwd3q.hip — bit-identical, measured 1-11 % SLOWER (profiles/r05_x4_*)
#endif
#ifndef DCV_DKDV_FORM
#define DCV_DKDV_FORM 3  // 2: dcv_attn_bwd_dkdv_rows_ps keeps the second form (A/B builds)
#endif

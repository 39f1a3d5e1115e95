// Shared pieces of the one-wave-per-SIMD attention backward kernels (attn_bwd3.hip: dK / dV; attn_bwd3q.hip: dQ): inline-asm MFMAs with explicit
// register classes, the scheduling fence and the value pin, bf16 packing.  See attn_bwd3.hip for the design and the hazard rules.
#pragma once
#include "attn_common.hpp"

namespace {

#define P3_FENCE() __builtin_amdgcn_sched_barrier(0)
#define P3_PIN(x) asm volatile("" : "+v"(x))
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {  // one v_cvt_pk_bf16_f32
    bf16x2_t v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void mfma_vv(f32x16& d, const bf16x8& a, const bf16x8& b) {  // d += a b   (d in arch VGPRs, b = K / V fragment in AGPRs)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_vc(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c) {  // d = a b + c
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma_aa(f32x16& d, const bf16x8& a, const bf16x8& b) {  // d += a b   (d in AGPRs)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}

}  // namespace

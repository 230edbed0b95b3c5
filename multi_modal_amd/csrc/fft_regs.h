// Register-resident radix building blocks for gfx950 (wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace mmfe {

// The fused multiply-adds of the helpers below are written out (fmaf) with the compiler's own contraction off inside
// them: which products get fused decides the low bits of a spectrum, and left to the optimiser that choice moves
// with instruction scheduling (two builds of one kernel then differ in the sixth digit of a log-mel value).

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
#pragma clang fp contract(off)
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// 4-point forward DFT in place: (a0,a1,a2,a3) -> (X0,X1,X2,X3), W4 = -i.
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = make_float2(t1.x + t3.y, t1.y - t3.x);
    a3 = make_float2(t1.x - t3.y, t1.y + t3.x);
}

// x *= W16^M  (W16 = exp(-2*pi*i/16)), M a compile-time constant in [0, 9].
template <int M>
__device__ __forceinline__ float2 mul_w16(float2 v) {
#pragma clang fp contract(off)
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    if constexpr (M == 0) return v;
    else if constexpr (M == 1) return make_float2(fmaf(v.x, C1, v.y * S1), fmaf(v.y, C1, -(v.x * S1)));
    else if constexpr (M == 2) return make_float2((v.x + v.y) * H, (v.y - v.x) * H);
    else if constexpr (M == 3) return make_float2(fmaf(v.x, S1, v.y * C1), fmaf(v.y, S1, -(v.x * C1)));
    else if constexpr (M == 4) return make_float2(v.y, -v.x);
    else if constexpr (M == 6) return make_float2((v.y - v.x) * H, -(v.x + v.y) * H);
    else {
        static_assert(M == 9, "unsupported W16 power");
        return make_float2(-fmaf(v.x, C1, v.y * S1), fmaf(v.x, S1, -(v.y * C1)));   // W16^9 = (-C1, +S1)
    }
}

// 16-point forward DFT in place.  Input x[n] in natural order; output X[k] is left at
// x[fft16_pos(k)] (digit-reversed positions), so callers index with fft16_pos() at compile time.
__host__ __device__ constexpr int fft16_pos(int k) { return 4 * (k & 3) + (k >> 2); }

// 4-point DFT whose inputs 1 and 3 arrive as H*p1 and H*p3 (H = sqrt(1/2), the W16^2 / W16^6 twiddles): the H rides
// on the fused multiply-adds of the last butterfly level.
__device__ __forceinline__ void dft4_h13(float2& a0, float2 p1, float2& a2, float2 p3, float2& o1, float2& o3) {
#pragma clang fp contract(off)
    constexpr float H = 0.70710678118654752f;
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), s = cadd(p1, p3), d = csub(p1, p3);
    a0 = make_float2(fmaf(H, s.x, t0.x), fmaf(H, s.y, t0.y));
    a2 = make_float2(fmaf(-H, s.x, t0.x), fmaf(-H, s.y, t0.y));
    o1 = make_float2(fmaf(H, d.y, t1.x), fmaf(-H, d.x, t1.y));
    o3 = make_float2(fmaf(-H, d.y, t1.x), fmaf(H, d.x, t1.y));
}

// 4-point DFT whose input 2 arrives as H*p2.
__device__ __forceinline__ void dft4_h2(float2& a0, float2& a1, float2& a2, float2 p2, float2& a3) {
#pragma clang fp contract(off)
    constexpr float H = 0.70710678118654752f;
    const float2 t0 = make_float2(fmaf(H, p2.x, a0.x), fmaf(H, p2.y, a0.y)), t1 = make_float2(fmaf(-H, p2.x, a0.x), fmaf(-H, p2.y, a0.y));
    const float2 t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = make_float2(t1.x + t3.y, t1.y - t3.x);
    a3 = make_float2(t1.x - t3.y, t1.y + t3.x);
}

__device__ __forceinline__ void fft16(float2 (&x)[16]) {
    // radix-4 over n1 for each n2: x[4*n1 + n2] -> A[n2][k1] stored at x[4*k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
    // twiddles W16^(n2*k1), then radix-4 over n2 for each k1: x[4*k1 + k2] = X[k1 + 4*k2].  W16^2 = H*(1 - i) and
    // W16^6 = -H*(1 + i): those products are left as (sum, difference) pairs and their H is applied inside the butterfly.
    dft4(x[0], x[1], x[2], x[3]);
    {   // k1 = 1: twiddles 1, W^1, W^2, W^3
        x[5] = mul_w16<1>(x[5]);
        x[7] = mul_w16<3>(x[7]);
        const float2 p2 = make_float2(x[6].x + x[6].y, x[6].y - x[6].x);
        dft4_h2(x[4], x[5], x[6], p2, x[7]);
    }
    {   // k1 = 2: twiddles 1, W^2, W^4, W^6
        const float2 p1 = make_float2(x[9].x + x[9].y, x[9].y - x[9].x);
        const float2 p3 = make_float2(x[11].y - x[11].x, -(x[11].x + x[11].y));
        x[10] = mul_w16<4>(x[10]);
        dft4_h13(x[8], p1, x[10], p3, x[9], x[11]);
    }
    {   // k1 = 3: twiddles 1, W^3, W^6, W^9
        x[13] = mul_w16<3>(x[13]);
        x[15] = mul_w16<9>(x[15]);
        const float2 p2 = make_float2(x[14].y - x[14].x, -(x[14].x + x[14].y));
        dft4_h2(x[12], x[13], x[14], p2, x[15]);
    }
}

// ---- DPP helpers (16-lane rows) ----------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {   // lanes without a source read 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_upd(float old, float v) {   // lanes without a source keep `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int DPP_ROW_SHR1 = 0x111;        // lane i <- lane i-1 of the same 16-lane row
constexpr int DPP_ROW_ROR1 = 0x121;        // lane i <- lane (i-1) mod 16
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <- lane 15-i
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <- lane 7-i within each half row

// Sum over each 16-lane row; every lane of the row ends with the same total.
__device__ __forceinline__ float row16_sum(float s) {
    s += dpp_mov<DPP_QUAD_XOR1>(s);
    s += dpp_mov<DPP_QUAD_XOR2>(s);
    s += dpp_mov<DPP_ROW_HALF_MIRROR>(s);
    s += dpp_mov<DPP_ROW_MIRROR>(s);
    return s;
}

__device__ __forceinline__ float read_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Maximum / minimum over each 16-lane row, then over the wave's four rows (lanes 0, 16, 32, 48 read back as
// scalars): every lane ends with the wave-wide extreme, no LDS traffic.
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    const float r0 = read_lane(v, 0), r1 = read_lane(v, 16), r2 = read_lane(v, 32), r3 = read_lane(v, 48);
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = fminf(v, dpp_mov<DPP_QUAD_XOR2>(v));
    v = fminf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = fminf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    const float r0 = read_lane(v, 0), r1 = read_lane(v, 16), r2 = read_lane(v, 32), r3 = read_lane(v, 48);
    return fminf(fminf(r0, r1), fminf(r2, r3));
}

// Value of lane (l + 32) for lanes l < 32 (v_permlane32_swap, a VALU operation); lanes 32..63 get their own value back.
__device__ __forceinline__ float upper_half_value(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[1]);
}

// Orders LDS traffic between the lanes of one wave (no instruction is emitted for it beyond what
// the memory model requires: a wave's DS operations already execute in program order).
__device__ __forceinline__ void wave_lds_fence() {
#ifndef MMFE_NO_WAVE_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Block barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter,
// i.e. it would wait for the drain phase's global stores to reach memory at every phase change.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace mmfe

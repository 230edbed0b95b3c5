// Log compression + affine output stage shared by the fused radix kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#include "mmfe.h"

namespace mmfe {

// The log kind is encoded as DATA, never as control flow: hipcc turns selects on wave-uniform
// conditions into scalar branches, and a taken branch per value costs more than the log itself.
struct LogAffine {
    float max_with;     // floor for the max kinds, -inf otherwise
    float add_to;       // epsilon for the add kind, 0 otherwise
    float mul;          // log-base change * out_scale
    float add;          // out_shift * out_scale
    float floor_cmp;    // floor for the max kinds, -inf otherwise
    float floor_val;    // value of floored cells
    float out_shift, out_scale;
    float none_flag;    // 1 when there is no log at all
    int kind;           // MMFE_LOG_*: for the block form below, which branches once per block of values
};

// Branch-free log + affine stage: result = (log(x) + out_shift) * out_scale, with the log taken by the
// hardware base-2 logarithm (v_log_f32, about 1 ulp of log2) and the base change folded into one fma:
//   result = log2(arg) * (log_b(2) * out_scale) + out_shift * out_scale.
// Absolute error of the log is below 4e-6 over the range of spectra (|log| <= 30).  Floored cells
// (x <= floor) take `floor_val` = (log(floor) + out_shift) * out_scale computed on the host from the
// host libm's log, the value a CPU front-end produces.  `none_flag` must be a per-lane (VGPR) copy of
// LogAffine::none_flag that the optimiser cannot see through, so the final select stays a v_cndmask.
__device__ __forceinline__ float apply_log_affine(float x, const LogAffine& L, float none_flag) {
    const float arg = fmaxf(x, L.max_with) + L.add_to;            // max kinds: max(x, floor) + 0;  add kind: x + eps
    float v = fmaf(__builtin_amdgcn_logf(arg), L.mul, L.add);
    v = x <= L.floor_cmp ? L.floor_val : v;                       // floor_cmp = -inf unless a max kind
    const float lin = (x + L.out_shift) * L.out_scale;
    return none_flag != 0.f ? lin : v;
}

// Block form: N values per lane behind ONE wave-uniform branch on the log kind (a branch per block costs a few
// cycles; the straight-line variants are 4 / 3 / 2 instructions per value instead of 10).  Max kinds take the log of
// the raw value: cells at or below the floor are replaced by `floor_val` anyway, whatever the log returned for them.
template <int N>
__device__ __forceinline__ void apply_log_affine_block(float (&x)[N], const LogAffine& L) {
    if (L.kind == MMFE_LOG_LN_MAX || L.kind == MMFE_LOG_LOG10_MAX) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float v = fmaf(__builtin_amdgcn_logf(x[i]), L.mul, L.add);
            x[i] = x[i] <= L.floor_cmp ? L.floor_val : v;
        }
    } else if (L.kind == MMFE_LOG_LN_ADD) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = fmaf(__builtin_amdgcn_logf(x[i] + L.add_to), L.mul, L.add);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = (x[i] + L.out_shift) * L.out_scale;
    }
}

// Host side.  `defer_affine` leaves the plain log in the output (shift 0, scale 1) for kernels that
// apply the affine map in a later pass.
inline LogAffine make_log_affine(const mmfe_spec& s, bool defer_affine = false) {
    const float shift = defer_affine ? 0.f : s.out_shift, scale = defer_affine ? 1.f : s.out_scale;
    const bool is_max = s.log_kind == MMFE_LOG_LN_MAX || s.log_kind == MMFE_LOG_LOG10_MAX;
    const double base = s.log_kind == MMFE_LOG_LOG10_MAX ? 0.30102999566398119521 : 0.69314718055994530942;   // log10(2), ln(2)
    const float lof = s.log_kind == MMFE_LOG_LOG10_MAX ? (float)std::log10((double)s.log_floor) : (float)std::log((double)s.log_floor);
    LogAffine la;
    la.max_with = is_max ? s.log_floor : -INFINITY;
    la.add_to = s.log_kind == MMFE_LOG_LN_ADD ? s.log_floor : 0.0f;
    la.mul = (float)(base * (double)scale);
    la.add = (float)((double)shift * (double)scale);
    la.floor_cmp = is_max ? s.log_floor : -INFINITY;
    la.floor_val = (lof + shift) * scale;
    la.out_shift = shift;
    la.out_scale = scale;
    la.none_flag = s.log_kind == MMFE_LOG_NONE ? 1.0f : 0.0f;
    la.kind = s.log_kind;
    return la;
}

}  // namespace mmfe

// Internal declarations shared by the translation units of libmmfe.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "mmfe.h"

namespace mmfe {

void set_error(const char* fmt, ...);

#define MMFE_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::mmfe::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MMFE_EHIP;                                                             \
        }                                                                                 \
    } while (0)

// Where the rows of a batch live.  Either a strided 2-D array, or slices of packed recordings.
struct RowSource {
    const void* base;          // strided: row r starts r*stride elements in.  slices: packed recordings.  Elements are
                               // floats or int16_t PCM, as the plan's in_dtype says
    int64_t stride;
    const int64_t* rec_start;  // slices only (device)
    const int32_t* rec_index;  // slices only (device)
    const int64_t* offsets;    // slices only (device)
    const int32_t* row_len;    // ragged batches only (device): samples of row r, clamped to [0, n_samples]; NULL = every row has n_samples
};

// Samples of row r of a batch whose longest row has n_samples.
__device__ __forceinline__ int row_samples(const RowSource& s, int64_t r, int n_samples) {
    if (s.row_len == nullptr) return n_samples;
    const int n = s.row_len[r];
    return n < 0 ? 0 : (n > n_samples ? n_samples : n);
}

// Frames an n-sample row yields under mmfe_spec.frames_mode (before drop_last / out_frames).
__host__ __device__ inline int64_t natural_frame_count(int frames_mode, int win_len, int hop, int n_fft, int64_t n) {
    switch (frames_mode) {
        case MMFE_FRAMES_SNIP: return n < win_len ? 0 : 1 + (n - win_len) / hop;
        case MMFE_FRAMES_CENTER: {
            const int64_t padded = n + 2 * (int64_t)(n_fft / 2);
            return padded < n_fft ? 0 : 1 + (padded - n_fft) / hop;
        }
        case MMFE_FRAMES_SCIPY: {
            const int64_t ext = n + 2 * (int64_t)(win_len / 2);
            if (ext < win_len) return 0;
            int64_t nadd = (-(ext - win_len)) % hop;
            if (nadd < 0) nadd += hop;
            nadd %= win_len;
            return 1 + (ext + nadd - win_len) / hop;
        }
        case MMFE_FRAMES_CEIL: return n <= win_len ? 1 : 1 + (n - win_len + hop - 1) / hop;
        default: return -1;
    }
}

template <typename S>
__device__ __forceinline__ const S* row_ptr_as(const RowSource& s, int64_t r) {
    const S* b = static_cast<const S*>(s.base);
    if (s.offsets == nullptr) return b + r * s.stride;
    return b + s.rec_start[s.rec_index[r]] + s.offsets[r];
}
__device__ __forceinline__ const float* row_ptr(const RowSource& s, int64_t r) { return row_ptr_as<float>(s, r); }

// Value of a stored sample: floats as they are, 16-bit PCM as s / 32768 (exact in float32).
__device__ __forceinline__ float sample_value(float v) { return v; }
__device__ __forceinline__ float sample_value(int16_t v) { return (float)v * (1.0f / 32768.0f); }

// Sample i of a row of n samples under the plan's padding rule (i may lie outside the row).
template <typename S>
__device__ __forceinline__ float fetch_sample(const S* row, int i, int n, int pad_mode) {
    if (i >= 0 && i < n) return sample_value(row[i]);
    if (n <= 0) return 0.f;
    if (pad_mode == MMFE_PAD_REFLECT) {
        if (n == 1) return sample_value(row[0]);
        const int period = 2 * (n - 1);
        int m = i % period;
        if (m < 0) m += period;
        return sample_value(row[m < n ? m : period - m]);
    }
    return 0.f;
}

// monotone map float -> int so that atomicMax on ints orders floats
__device__ __forceinline__ int float_to_ordered(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

// ---------------------------------------------------------------------------------------------
// mel filter bank in the layout the radix kernels consume: filters are taken 16 at a time
// ("pass" j holds filters 16j .. 16j+15, one per lane of a 16-lane frame group); inside a pass
// every lane walks the same number of taps (the longest band of the pass), shorter bands are
// padded with zero weights.
struct MelPassTable {
    static constexpr int kMaxPass = 16;      // up to 256 filters
    int n_pass = 0;
    int niter[kMaxPass] = {0};               // taps walked in pass j
    int woff[kMaxPass] = {0};                // first tap row of pass j in `weights` (rows of 16 floats)
    int total_rows = 0;
    std::vector<int32_t> start;              // n_pass*16 first bins (clamped so start+niter stays in the spectrum buffer)
    std::vector<float> weights;              // total_rows*16
};

struct Plan {
    mmfe_spec spec;
    int device = -1;
    int n_bins = 0;
    // host copies
    std::vector<float> window;        // fp32 view of the window
    std::vector<double> window64;     // the window as given (float64 when the caller supplied doubles)
    std::vector<int32_t> band_start, band_len;
    std::vector<float> band_weights;
    MelPassTable mel16;
    // device tables (plan lifetime)
    float* d_window = nullptr;        // n_fft floats: window placed at win_off, zeros elsewhere (pre-scaled where a kernel says so)
    float* d_tw_a = nullptr;          // kernel-specific twiddle table A
    float* d_tw_b = nullptr;          // kernel-specific twiddle table B
    double* d_tw64 = nullptr;         // float64 twiddle table (any-length DFT kernel)
    double* d_window64 = nullptr;     // float64 window (any-length DFT kernel)
    int32_t* d_mel_start = nullptr;
    float* d_mel_w = nullptr;
    int32_t* d_mel_niter = nullptr;
    int32_t* d_mel_woff = nullptr;
    int mel4_nga = 0, mel4_ngb = 0;   // matrix-pipe mel form: groups of 4 bins per filter set (0 = not used)
    int32_t* d_mel4_kstart = nullptr;
    float* d_mel4_w = nullptr;
    int r400_shape = -1;              // n_fft = 400 kernel: index of the instantiated mel stream shape
    int32_t* d_r400_dst = nullptr;    // n_fft = 400 kernel: (filter, frame group) served by each lane of each stream
    int32_t* d_band_start = nullptr;  // plain CSR copy (generic DFT kernel)
    int32_t* d_band_len = nullptr;
    int32_t* d_band_off = nullptr;
    float* d_band_w = nullptr;
    float* d_rowmax = nullptr;        // scratch for clamp_below_max: per-row maxima [0, capacity) and minima [capacity, 2*capacity)
    int64_t rowmax_capacity = 0;
    // tuning overrides, read from the environment once at plan creation (0 = the kernel's own choice)
    int tune_blocks_per_cu = 0;       // n_fft 512 kernel: upper limit on resident blocks per CU; STFT-32 kernel: rows per group
    int tune_chunk = 0;               // n_fft 512 kernel: consecutive frame quads a wave takes per visit; STFT-32 kernel: mover waves (-1: phase-serial form)
    int tune_nt = 1;                  // STFT-32 kernel: output stores bypass the caches
    // resampler, wave-uniform-phase kernel (rs_L == 0: not applicable, the general kernel serves the plan)
    int rs_q = 4, rs_L = 0, rs_D = 0, rs_nb = 0, rs_groups = 0, rs_wlen = 0, rs_wchunks = 0, rs_tile_in = 0, rs_lds_in = 0, rs_pad = 0, rs_round_out = 0,
        rs_ls = 0, rs_threads = 0;
    // the same for 16-bit PCM rows, two blocks per CU (rs16_threads == 0: not applicable)
    int rs16_q = 0, rs16_L = 0, rs16_D = 0, rs16_groups = 0, rs16_wlen = 0, rs16_tile_in = 0, rs16_ls = 0, rs16_threads = 0, rs16_round_out = 0;
    int impl = 0;                     // which kernel family serves this plan
};

enum Impl { IMPL_FBANK512 = 1, IMPL_STFT32 = 2, IMPL_DFT_GENERIC = 3, IMPL_FRAME = 4, IMPL_RFFT400 = 5, IMPL_RESAMPLE = 6 };

// Where a clamp_below_max call keeps its per-row extremes (ordered-int maxima at [0, capacity), minima at
// [capacity, 2*capacity)): the plan's own buffer, or the library's static device array (mmfe_run_noalloc).
struct ClampScratch {
    float* base;
    int64_t capacity;
};

// launchers (one per translation unit)
int fbank512_prepare(Plan& plan);
int fbank512_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                    int64_t out_frames, void* out, hipStream_t stream);

int stft32_prepare(Plan& plan);
int stft32_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames,
                  float* out, hipStream_t stream);

int dftgen_prepare(Plan& plan);
int dftgen_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                  int64_t out_frames, float* out, const ClampScratch& cs, hipStream_t stream);

// per-row extremes + clamp pass of clamp_below_max: out = (max(out, rowmax - c) + out_shift) * out_scale
int rowmax_init_launch(const ClampScratch& cs, int64_t n_rows, hipStream_t stream);
int clamp_affine_launch(const Plan& plan, const ClampScratch& cs, float* out, int64_t n_rows, int64_t per_row, hipStream_t stream);

bool rfft400_supports(const mmfe_spec& spec);
int rfft400_prepare(Plan& plan);
int rfft400_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                   int64_t out_frames, float* out, const ClampScratch& cs, hipStream_t stream);

int frame_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames,
                 void* out, hipStream_t stream);

// polyphase resampler (MMFE_KIND_RESAMPLE): d_window holds the [up][taps per phase] bank
int resample_prepare(Plan& plan);
int64_t resample_out_len(const mmfe_spec& spec, int64_t n_samples);
int resample_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_out_natural, int64_t out_len,
                    float* out, hipStream_t stream);

// ragged batches: row r is padded from frame min(natural frames of its own length - drop_last, out_frames) on
int fill_pad_rows_ragged(void* out, int out_dtype, int64_t n_rows, int64_t out_frames, int64_t n_feat, int layout, float value,
                         const mmfe_spec& spec, const int32_t* row_len, int64_t n_samples, hipStream_t stream);

int fill_pad_rows(void* out, int out_dtype, int64_t n_rows, int64_t out_frames, int64_t first_pad_frame, int64_t n_feat, int layout,
                  float value, hipStream_t stream);

int feature_sums_launch(const float* x, int64_t n, double* sums, hipStream_t stream);

// Output element conversions (round to nearest even), as raw 16-bit patterns.
__device__ __forceinline__ unsigned short f16_bits(float v) {
    const _Float16 h = static_cast<_Float16>(v);
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned short bf16_bits(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Two values rounded and packed into one 32-bit word (low half = first value); bfloat16 uses gfx950's packed convert.
__device__ __forceinline__ unsigned pack2_bits(float a, float b, int out_dtype) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    if (out_dtype == MMFE_OUT_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
    return (unsigned)f16_bits(a) | (unsigned)f16_bits(b) << 16;
}

// Launch-geometry helpers (thread-safe, cached per kernel and device).
//   kernel_prepare: raises the kernel's dynamic-LDS limit to the CU's 160 KB on the current device (once) and
//   returns how many blocks of `threads` threads with `lds` bytes of dynamic LDS are resident per CU, and the CU count.
int kernel_prepare(const void* fn, int threads, size_t lds, int* blocks_per_cu, int* cus);

template <typename T>
int upload(T** dptr, const std::vector<T>& host) {
    *dptr = nullptr;
    if (host.empty()) return MMFE_OK;
    MMFE_HIP_CHECK(hipMalloc((void**)dptr, host.size() * sizeof(T)));
    MMFE_HIP_CHECK(hipMemcpy(*dptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return MMFE_OK;
}

}  // namespace mmfe

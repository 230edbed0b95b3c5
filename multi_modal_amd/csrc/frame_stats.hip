// Layout-only helpers of the front-end: framing with a window (no transform), padded-frame fill,
// and the feature statistics reduction.
//
//   frame_kernel       frame_signal(signal, frame_length, hop_length, window)
//                      (classification/vggsound/utils/transforms.py:19-61): centre zero padding split
//                      floor/ceil, strided gather, multiply by the window.
//   fill_kernel        the ZeroPad2d rows appended when a clip has fewer than out_frames frames
//                      (classification/vggsound/dataset/vggsound.py:52-54), already normalised.
//   sums_kernel        sum and sum of squares of a feature tensor, the arithmetic of the two passes of
//                      cal_norm (classification/vggsound/dataset/vggsound.py:85-107), in float64.
#include "mmfe_internal.h"

#include <algorithm>
#include <cstring>

namespace mmfe {

namespace {

struct FrameArgs {
    RowSource src;
    void* out;              // elements of the input's type
    const void* window;     // doubles for float64 rows, floats otherwise
    int64_t n_rows;
    int n_samples, n_frames, frame_len, hop, left_pad;
};

// Element types of frame_signal.  The reference multiplies in the dtype of its operands (transforms.py:58-59):
// float32 and float64 products are single correctly rounded operations; a float16 / bfloat16 product is the exact product
// (at most 22 significant bits, it fits float32) rounded once to 16 bits, which is what torch computes for those dtypes.
template <int IN_DTYPE> struct FrameElem;
template <> struct FrameElem<MMFE_IN_F32> {
    typedef float store; typedef float win;
    static __device__ __forceinline__ float mul(float v, float w) { return v * w; }
};
template <> struct FrameElem<MMFE_IN_F64> {
    typedef double store; typedef double win;
    static __device__ __forceinline__ double mul(double v, double w) { return v * w; }
};
template <> struct FrameElem<MMFE_IN_F16> {
    typedef _Float16 store; typedef float win;
    static __device__ __forceinline__ _Float16 mul(_Float16 v, float w) { return static_cast<_Float16>((float)v * w); }
};
template <> struct FrameElem<MMFE_IN_BF16> {
    typedef unsigned short store; typedef float win;
    static __device__ __forceinline__ unsigned short mul(unsigned short v, float w) { return bf16_bits(__uint_as_float((unsigned)v << 16) * w); }
};

template <int IN_DTYPE>
__global__ void __launch_bounds__(256) frame_kernel(const FrameArgs p) {
    typedef FrameElem<IN_DTYPE> E;
    typedef typename E::store T;
    const typename E::win* window = static_cast<const typename E::win*>(p.window);
    T* out = static_cast<T*>(p.out);
    const int64_t per_row = (int64_t)p.n_frames * p.frame_len;
    const int64_t total = p.n_rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / per_row;
        const int rem = (int)(i - row * per_row);
        const int f = rem / p.frame_len;
        const int j = rem - f * p.frame_len;
        const int idx = f * p.hop + j - p.left_pad;
        T v = T(0);
        if (idx >= 0 && idx < p.n_samples) v = row_ptr_as<T>(p.src, row)[idx];
        out[i] = E::mul(v, window[j]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) fill_kernel(T* out, int64_t n_rows, int64_t out_frames, int64_t first_pad,
                                                   int64_t n_feat, int layout, T value) {
    const int64_t pad_frames = out_frames - first_pad;
    const int64_t per_row = pad_frames * n_feat;
    const int64_t total = n_rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / per_row;
        const int64_t rem = i - row * per_row;
        int64_t o;
        if (layout == MMFE_LAYOUT_TIME_MAJOR) {
            o = (row * out_frames + first_pad) * n_feat + rem;          // padded frames are one contiguous tail
        } else {
            const int64_t m = rem / pad_frames, t = rem - m * pad_frames;
            o = (row * n_feat + m) * out_frames + first_pad + t;
        }
        out[o] = value;
    }
}

// Ragged batches: one block per row (grid-stride); row r is padded from its own first missing frame on.
struct RaggedFill {
    const int32_t* row_len;
    int64_t n_rows, out_frames, n_feat;
    int n_samples, layout, frames_mode, win_len, hop, n_fft, drop_last;
};
template <typename T>
__global__ void __launch_bounds__(256) fill_ragged_kernel(T* out, const RaggedFill p, T value) {
    for (int64_t row = blockIdx.x; row < p.n_rows; row += gridDim.x) {
        int n = p.row_len[row];
        n = n < 0 ? 0 : (n > p.n_samples ? p.n_samples : n);
        int64_t first = natural_frame_count(p.frames_mode, p.win_len, p.hop, p.n_fft, n);
        if (p.drop_last && first > 0) first -= 1;
        if (first >= p.out_frames) continue;
        const int64_t pad_frames = p.out_frames - first;
        const int64_t per_row = pad_frames * p.n_feat;
        for (int64_t rem = threadIdx.x; rem < per_row; rem += blockDim.x) {
            int64_t o;
            if (p.layout == MMFE_LAYOUT_TIME_MAJOR) {
                o = (row * p.out_frames + first) * p.n_feat + rem;
            } else {
                const int64_t m = rem / pad_frames, t = rem - m * pad_frames;
                o = (row * p.n_feat + m) * p.out_frames + first + t;
            }
            out[o] = value;
        }
    }
}

__global__ void __launch_bounds__(256) sums_kernel(const float* x, int64_t n, double* sums) {
    double s = 0.0, q = 0.0;
    const int64_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        const float v = x[n4 * 4 + threadIdx.x];
        s += v;
        q += (double)v * v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    __shared__ double sh[2][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = s; sh[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(&sums[1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

int grid_for(int64_t total, int per_block) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return (int)std::max<int64_t>(1, std::min<int64_t>((total + per_block - 1) / per_block, (int64_t)cus * 8));
}

}  // namespace

int frame_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames,
                 void* out, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    if (n_rows <= 0 || n_frames <= 0) return MMFE_OK;
    if (n_samples >= (int64_t)1 << 30 || n_frames * (int64_t)s.win_len >= (int64_t)1 << 31) {
        set_error("frame: row too long for one launch");
        return MMFE_EINVAL;
    }
    FrameArgs a{};
    a.src = src;
    a.out = out;
    a.window = s.in_dtype == MMFE_IN_F64 ? static_cast<const void*>(plan.d_window64) : static_cast<const void*>(plan.d_window);
    a.n_rows = n_rows;
    a.n_samples = (int)n_samples;
    a.n_frames = (int)n_frames;
    a.frame_len = s.win_len;
    a.hop = s.hop;
    const int64_t pad_len = (n_frames - 1) * s.hop + s.win_len;
    a.left_pad = pad_len > n_samples ? (int)((pad_len - n_samples) / 2) : 0;
    const int64_t total = n_rows * n_frames * s.win_len;
    const dim3 grid(grid_for(total, 256));
    switch (s.in_dtype) {
        case MMFE_IN_F32: hipLaunchKernelGGL(frame_kernel<MMFE_IN_F32>, grid, dim3(256), 0, stream, a); break;
        case MMFE_IN_F64: hipLaunchKernelGGL(frame_kernel<MMFE_IN_F64>, grid, dim3(256), 0, stream, a); break;
        case MMFE_IN_F16: hipLaunchKernelGGL(frame_kernel<MMFE_IN_F16>, grid, dim3(256), 0, stream, a); break;
        case MMFE_IN_BF16: hipLaunchKernelGGL(frame_kernel<MMFE_IN_BF16>, grid, dim3(256), 0, stream, a); break;
        default: set_error("frame: unsupported in_dtype %d", s.in_dtype); return MMFE_EINVAL;
    }
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

// the 16-bit pattern of the rounded value, computed on the host with the kernels' round-to-nearest-even rule
static unsigned short half_bits(float value, int out_dtype) {
    unsigned short bits;
    if (out_dtype == MMFE_OUT_F16) {
        const _Float16 h = static_cast<_Float16>(value);
        std::memcpy(&bits, &h, 2);
    } else {
        unsigned u;
        std::memcpy(&u, &value, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) u = (u | 0x400000u);
        else u += 0x7fffu + ((u >> 16) & 1u);
        bits = (unsigned short)(u >> 16);
    }
    return bits;
}

int fill_pad_rows_ragged(void* out, int out_dtype, int64_t n_rows, int64_t out_frames, int64_t n_feat, int layout, float value,
                         const mmfe_spec& spec, const int32_t* row_len, int64_t n_samples, hipStream_t stream) {
    if (n_rows <= 0 || out_frames <= 0 || n_feat <= 0) return MMFE_OK;
    RaggedFill p{row_len, n_rows, out_frames, n_feat, (int)n_samples, layout, spec.frames_mode, spec.win_len, spec.hop, spec.n_fft, spec.drop_last};
    const int grid = grid_for(n_rows, 1);
    if (out_dtype == MMFE_OUT_F32)
        hipLaunchKernelGGL(fill_ragged_kernel<float>, dim3(grid), dim3(256), 0, stream, static_cast<float*>(out), p, value);
    else
        hipLaunchKernelGGL(fill_ragged_kernel<unsigned short>, dim3(grid), dim3(256), 0, stream, static_cast<unsigned short*>(out), p,
                           half_bits(value, out_dtype));
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

int fill_pad_rows(void* out, int out_dtype, int64_t n_rows, int64_t out_frames, int64_t first_pad_frame, int64_t n_feat, int layout,
                  float value, hipStream_t stream) {
    if (n_rows <= 0 || first_pad_frame >= out_frames) return MMFE_OK;
    const int64_t total = n_rows * (out_frames - first_pad_frame) * n_feat;
    if (out_dtype == MMFE_OUT_F32) {
        hipLaunchKernelGGL(fill_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, stream, static_cast<float*>(out), n_rows,
                           out_frames, first_pad_frame, n_feat, layout, value);
    } else {
        const unsigned short bits = half_bits(value, out_dtype);
        hipLaunchKernelGGL(fill_kernel<unsigned short>, dim3(grid_for(total, 256)), dim3(256), 0, stream,
                           static_cast<unsigned short*>(out), n_rows, out_frames, first_pad_frame, n_feat, layout, bits);
    }
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

int feature_sums_launch(const float* x, int64_t n, double* sums, hipStream_t stream) {
    if (n <= 0) return MMFE_OK;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0) {
        set_error("feature_sums: input must be 16-byte aligned");
        return MMFE_EINVAL;
    }
    hipLaunchKernelGGL(sums_kernel, dim3(grid_for(n / 4 + 1, 256 * 8)), dim3(256), 0, stream, x, n, sums);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace mmfe

// Short-window STFT magnitude for n_fft = 32 on gfx950 (many tiny FFTs, write-dominated).
//
// Stands behind  np.abs(scipy.signal.stft(x, fs=100, nperseg=32, noverlap=28)[-1])  as the reference
// calls it (classification/har/adbox.py:20): zero extension by nperseg/2 on both sides, zero padding
// to a whole number of hops, periodic-Hann window, one-sided 32-point DFT scaled by 1/sum(window),
// magnitude, output laid out (row, bin, frame).
//
// Mapping: one lane owns one frame.  A block stages a group of consecutive rows in LDS with zero
// margins (coalesced 16-byte global reads), every lane pulls its 32 samples with eight 16-byte LDS
// reads, runs the 32-point real FFT as a 16-point complex FFT plus recombination entirely in
// registers, and stores 17 magnitudes.  Lanes of a wave hold consecutive frames of a row, so each
// of the 17 stores of a wave covers contiguous runs of the (row, bin, frame) output.
#include "mmfe_internal.h"
#include "fft_regs.h"

#include <algorithm>
#include <cmath>

namespace mmfe {

namespace {

constexpr int kThreads = 256;
constexpr int kN = 32;

struct StftArgs {
    RowSource src;
    float* out;
    int n_rows;
    int n_samples;
    int n_frames;        // S
    int hop;
    int rows_per_block;  // R
    int row_lds;         // floats of LDS per staged row (16 + n_samples + tail, multiple of 4)
    int power;           // 1 = magnitude, 2 = power
    float win[kN];       // window pre-multiplied by 0.5 * spec_scale
};

// -i * exp(-2 pi i k / 32), k = 0..8
__device__ __forceinline__ float2 tw32(int k) {
    constexpr float S[9] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                            0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.f};
    return make_float2(-S[k], -S[8 - k]);
}

__global__ void __launch_bounds__(kThreads) stft32_kernel(const StftArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_in = reinterpret_cast<float*>(smem);

    const int R = p.rows_per_block;
    const int S = p.n_frames;
    const int T = p.n_samples;
    const int n_groups = (p.n_rows + R - 1) / R;

    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int row0 = grp * R;
        const int rows = min(R, p.n_rows - row0);
        __syncthreads();   // previous group's readers are done with s_in
        // stage: zero margins + samples
        for (int i = threadIdx.x; i < rows * p.row_lds; i += kThreads) {
            const int r = i / p.row_lds;
            const int j = i - r * p.row_lds - 16;          // sample index
            float v = 0.f;
            if (j >= 0 && j < T) v = row_ptr(p.src, row0 + r)[j];
            s_in[i] = v;
        }
        __syncthreads();

        const int n_fr = rows * S;
        for (int f = threadIdx.x; f < n_fr; f += kThreads) {
            const int r = f / S;
            const int s = f - r * S;
            const float* xp = s_in + r * p.row_lds + s * p.hop;    // frame s covers samples [s*hop-16, s*hop+16)
            float2 z[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                z[n] = make_float2(xp[2 * n] * p.win[2 * n], xp[2 * n + 1] * p.win[2 * n + 1]);
            }
            fft16(z);
            float* o = p.out + ((int64_t)(row0 + r) * 17) * S + s;
#pragma unroll
            for (int k = 0; k <= 8; ++k) {
                const float2 zk = z[fft16_pos(k)];
                const float2 zp = z[fft16_pos((16 - k) & 15)];
                const float2 tw = tw32(k);
                const float2 E = make_float2(zk.x + zp.x, zk.y - zp.y);
                const float2 D = make_float2(zk.x - zp.x, zk.y + zp.y);
                const float2 G = cmul(tw, D);
                const float2 xa = cadd(E, G), xb = csub(E, G);
                float pa = xa.x * xa.x + xa.y * xa.y;
                float pb = xb.x * xb.x + xb.y * xb.y;
                if (p.power == 1) { pa = sqrtf(pa); pb = sqrtf(pb); }
                o[(int64_t)k * S] = pa;
                if (k != 8) o[(int64_t)(16 - k) * S] = pb;
            }
        }
    }
}

}  // namespace

int stft32_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    if (s.n_fft != kN || s.win_len != kN || s.win_off != 0 || s.hop < 1 || s.hop > kN || s.n_mel != 0 ||
        s.log_kind != MMFE_LOG_NONE || s.layout != MMFE_LAYOUT_FEAT_MAJOR || s.frames_mode != MMFE_FRAMES_SCIPY ||
        s.remove_dc || s.preemph != 0.f || s.out_frames != 0 || s.out_shift != 0.f || s.out_scale != 1.f) {
        set_error("stft32: needs n_fft == win_len == 32, 1 <= hop <= 32, scipy framing, feature-major magnitude/power output, no mel/log/affine");
        return MMFE_EINVAL;
    }
    return MMFE_OK;
}

int stft32_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames,
                  float* out, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    if (n_rows <= 0 || n_frames <= 0) return MMFE_OK;
    if (n_rows >= (int64_t)1 << 31 || n_samples >= (int64_t)1 << 24) {
        set_error("stft32: batch too large for one launch");
        return MMFE_EINVAL;
    }
    StftArgs a{};
    a.src = src;
    a.out = out;
    a.n_rows = (int)n_rows;
    a.n_samples = (int)n_samples;
    a.n_frames = (int)n_frames;
    a.hop = s.hop;
    a.power = s.power;
    for (int i = 0; i < kN; ++i) a.win[i] = plan.window[i] * (0.5f * s.spec_scale);
    // LDS row: 16 leading zeros + samples + zeros up to the end of the last frame
    const int need = (int)((n_frames - 1) * s.hop + kN);
    a.row_lds = (std::max(need, (int)n_samples + 16) + 3) & ~3;
    // rows per block: fill the 256 threads well without exceeding ~48 KB of LDS
    const int max_rows = std::max(1, std::min<int>(64, (48 * 1024) / (a.row_lds * 4)));
    int best_r = 1;
    double best_eff = 0.0;
    for (int r = 1; r <= max_rows; ++r) {
        const double fr = (double)r * (double)n_frames;
        const double eff = fr / (std::ceil(fr / kThreads) * kThreads);
        if (eff > best_eff + 1e-9) { best_eff = eff; best_r = r; }
    }
    a.rows_per_block = best_r;
    const size_t lds = (size_t)a.rows_per_block * a.row_lds * 4;
    if (lds > 64 * 1024) {
        set_error("stft32: a row of %lld samples does not fit the LDS staging buffer", (long long)n_samples);
        return MMFE_EINVAL;
    }
    int dev = 0, cus = 256;
    MMFE_HIP_CHECK(hipGetDevice(&dev));
    MMFE_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t groups = (n_rows + a.rows_per_block - 1) / a.rows_per_block;
    const int grid = (int)std::min<int64_t>(groups, (int64_t)cus * 8);
    hipLaunchKernelGGL(stft32_kernel, dim3(grid), dim3(kThreads), lds, stream, a);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace mmfe

// Rational-ratio polyphase FIR resampler on gfx950 (MMFE_KIND_RESAMPLE): the step the reference runs offline, on the CPU,
// in front of the spectrogram path: librosa.load(file, sr=16000, duration=10) turns the 44.1 / 48 kHz VGGSound tracks into
// the 16 kHz clips the dataset stores (classification/vggsound/dataset/vggsound.py:114), and ffmpeg -ar 24000 does the same
// for EPIC (classification/recognition/dataset/extract_audio.py:6-13).
//
// out[m] = sum_j bank[phase(m)][j] * x[i(m) - j],   t = (m + pre) * down,  i(m) = t / up,  phase(m) = t % up
// with bank[p][j] = h_padded[p + j * up]: upsample by `up`, FIR-filter with h, keep every `down`-th sample, with the
// alignment and output count of scipy.signal.resample_poly (h centred on the output grid, ceil(n * up / down) outputs,
// zeros outside the row).
//
// A block takes a tile of consecutive outputs of one row.  The input samples under the tile are staged once in LDS by
// coalesced loads (16-bit PCM is decoded there), the polyphase bank sits in LDS as well (rows padded to whole 16-byte
// groups, so a lane fetches four taps per read); each thread then walks its outputs.  HBM traffic is the algorithmic
// minimum plus the J - 1 samples of history per tile.
#include "mmfe_internal.h"

#include <algorithm>
#include <vector>

namespace mmfe {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxTileIn = 12288;        // floats of staged input per tile (48 KiB)
constexpr int kMaxBankBytes = 96 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ResampleArgs {
    RowSource src;
    float* out;
    int n_rows;
    int n_in;                  // samples per row (ragged batches: of the longest row)
    int64_t n_out;             // outputs computed per row
    int64_t out_stride;        // floats between output rows
    int up, down;
    int taps;                  // J: taps per phase
    int taps_padded;           // J rounded up to a multiple of 4
    int pre;                   // outputs of the full filter response dropped in front (n_pre_remove)
    int tile_out;              // outputs per tile
    int tile_in;               // staged input samples per tile
    int tiles_per_row;
    const float* bank;         // [up][taps_padded]
};

template <typename SMP, bool BANK_IN_LDS>
__global__ void __launch_bounds__(kThreads) resample_kernel(const ResampleArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bank = reinterpret_cast<float*>(smem);
    float* s_x = s_bank + (BANK_IN_LDS ? p.up * p.taps_padded : 0);
    if (BANK_IN_LDS) {
        const int n4 = p.up * p.taps_padded / 4;
        for (int i = threadIdx.x; i < n4; i += kThreads) reinterpret_cast<f32x4*>(s_bank)[i] = reinterpret_cast<const f32x4*>(p.bank)[i];
    }
    const float* bank = BANK_IN_LDS ? s_bank : p.bank;
    const int64_t n_tiles = (int64_t)p.n_rows * p.tiles_per_row;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = (int)(tile / p.tiles_per_row);
        const int64_t m0 = (tile - (int64_t)row * p.tiles_per_row) * p.tile_out;
        const SMP* x = row_ptr_as<SMP>(p.src, row);
        const int n_row = row_samples(p.src, row, p.n_in);
        // staged range: from the oldest sample the tile's first output reads
        const int64_t i_lo = ((m0 + p.pre) * p.down) / p.up - (p.taps_padded - 1);
        __syncthreads();                                   // the previous tile's readers are done (and the bank is in place)
        for (int i = threadIdx.x; i < p.tile_in; i += kThreads) {
            const int64_t g = i_lo + i;
            s_x[i] = (g >= 0 && g < n_row) ? sample_value(x[g]) : 0.f;
        }
        __syncthreads();
        float* orow = p.out + (int64_t)row * p.out_stride;
        for (int k = threadIdx.x; k < p.tile_out; k += kThreads) {
            const int64_t m = m0 + k;
            if (m >= p.n_out) break;
            const int64_t t = (m + p.pre) * p.down;
            const int64_t im = t / p.up;
            const int phase = (int)(t - im * p.up);
            const float* xp = s_x + (int)(im - i_lo);      // x[i(m)]; tap j pairs with xp[-j]
            const float* bp = bank + (size_t)phase * p.taps_padded;
            float acc = 0.f;
            for (int j = 0; j < p.taps_padded; j += 4) {    // padded taps are zero and pair with staged (finite) samples
                const f32x4 w = *reinterpret_cast<const f32x4*>(bp + j);
                acc = fmaf(w.x, xp[-j], acc);
                acc = fmaf(w.y, xp[-j - 1], acc);
                acc = fmaf(w.z, xp[-j - 2], acc);
                acc = fmaf(w.w, xp[-j - 3], acc);
            }
            __builtin_nontemporal_store(acc, orow + m);
        }
    }
}

}  // namespace

int resample_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    const int up = s.n_fft, down = s.hop, n_taps = s.win_len;
    if (up < 1 || down < 1 || up > 4096 || down > 65536) { set_error("RESAMPLE: up (n_fft) must lie in [1, 4096], down (hop) in [1, 65536]"); return MMFE_EINVAL; }
    int a = up, b = down;
    while (b) { const int t = a % b; a = b; b = t; }
    if (a != 1) { set_error("RESAMPLE: up and down must be coprime (divide both by %d)", a); return MMFE_EINVAL; }
    if (n_taps < 1 || n_taps % 2 == 0) { set_error("RESAMPLE: the filter needs an odd number of taps (win_len), got %d", n_taps); return MMFE_EINVAL; }
    if (s.n_mel != 0 || s.log_kind != MMFE_LOG_NONE || s.out_dtype != MMFE_OUT_F32 || s.clamp_below_max > 0.f || s.drop_last ||
        s.layout != MMFE_LAYOUT_TIME_MAJOR) {
        set_error("RESAMPLE: no mel bank, log, clamp or 16-bit output");
        return MMFE_EINVAL;
    }
    // alignment of scipy.signal.resample_poly: h is padded in front so that the kept outputs start on the input grid
    const int half = (n_taps - 1) / 2;
    const int pre_pad = down - half % down;
    const int padded_len = pre_pad + n_taps;
    const int taps = (padded_len + up - 1) / up;
    const int taps_padded = (taps + 3) & ~3;
    std::vector<float> bank((size_t)up * taps_padded, 0.f);
    for (int p = 0; p < up; ++p)
        for (int j = 0; j < taps; ++j) {
            const int idx = p + j * up - pre_pad;
            if (idx >= 0 && idx < n_taps) bank[(size_t)p * taps_padded + j] = plan.window[idx];
        }
    // the staged input of even a one-thread-per-output tile must fit
    if ((int64_t)kThreads * down / up + taps_padded + 8 > kMaxTileIn) {
        set_error("RESAMPLE: ratio %d/%d with %d taps per phase needs more than %d staged samples per tile", up, down, taps, kMaxTileIn);
        return MMFE_EINVAL;
    }
    return upload(&plan.d_window, bank);
}

int64_t resample_out_len(const mmfe_spec& s, int64_t n) {
    const int64_t v = n * s.n_fft;
    return v / s.hop + (v % s.hop ? 1 : 0);
}

int resample_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_out_natural, int64_t out_len,
                    float* out, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    const int64_t n_out = std::min(n_out_natural, out_len);
    if (n_rows <= 0 || n_out <= 0) return MMFE_OK;
    if (n_samples >= ((int64_t)1 << 31) / 2 || n_rows >= (int64_t)1 << 31) { set_error("RESAMPLE: row too long for one launch"); return MMFE_EINVAL; }
    const int up = s.n_fft, down = s.hop, n_taps = s.win_len;
    const int half = (n_taps - 1) / 2;
    const int pre_pad = down - half % down;
    ResampleArgs a{};
    a.src = src;
    a.out = out;
    a.n_rows = (int)n_rows;
    a.n_in = (int)n_samples;
    a.n_out = n_out;
    a.out_stride = out_len;
    a.up = up;
    a.down = down;
    a.taps = (pre_pad + n_taps + up - 1) / up;
    a.taps_padded = (a.taps + 3) & ~3;
    a.pre = (half + pre_pad) / down;
    // largest tile (a multiple of the block size, at most 16 outputs per thread) whose staged input fits
    int per_thread = 16;
    auto tile_in_of = [&](int tile_out) { return (int)(((int64_t)tile_out * down + up - 1) / up) + a.taps_padded + 8; };
    while (per_thread > 1 && tile_in_of(per_thread * kThreads) > kMaxTileIn) per_thread /= 2;
    a.tile_out = per_thread * kThreads;
    a.tile_in = tile_in_of(a.tile_out);
    a.tiles_per_row = (int)((n_out + a.tile_out - 1) / a.tile_out);
    a.bank = plan.d_window;
    const size_t bank_bytes = (size_t)up * a.taps_padded * 4;
    const bool bank_lds = bank_bytes <= (size_t)kMaxBankBytes;
    const size_t lds = (bank_lds ? bank_bytes : 0) + (size_t)a.tile_in * 4;
    const bool pcm = s.in_dtype == MMFE_IN_S16;
    const void* fn = pcm ? (bank_lds ? reinterpret_cast<const void*>(&resample_kernel<int16_t, true>) : reinterpret_cast<const void*>(&resample_kernel<int16_t, false>))
                         : (bank_lds ? reinterpret_cast<const void*>(&resample_kernel<float, true>) : reinterpret_cast<const void*>(&resample_kernel<float, false>));
    int per_cu = 1, cus = 256;
    const int rc = kernel_prepare(fn, kThreads, lds, &per_cu, &cus);
    if (rc) return rc;
    const int64_t n_tiles = n_rows * a.tiles_per_row;
    const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cus * std::max(1, per_cu));
    if (pcm) {
        if (bank_lds) hipLaunchKernelGGL((resample_kernel<int16_t, true>), dim3(grid), dim3(kThreads), lds, stream, a);
        else hipLaunchKernelGGL((resample_kernel<int16_t, false>), dim3(grid), dim3(kThreads), lds, stream, a);
    } else {
        if (bank_lds) hipLaunchKernelGGL((resample_kernel<float, true>), dim3(grid), dim3(kThreads), lds, stream, a);
        else hipLaunchKernelGGL((resample_kernel<float, false>), dim3(grid), dim3(kThreads), lds, stream, a);
    }
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace mmfe

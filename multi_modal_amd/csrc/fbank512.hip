// Fused kaldi-style log-mel filterbank for n_fft = 512 on gfx950.
//
// Stands behind torchaudio.compliance.kaldi.fbank as the reference calls it
// (classification/vggsound/dataset/vggsound.py:45-47) plus the pad/cut/normalise of
// vggsound.py:48-59: per frame  DC removal -> pre-emphasis -> window -> zero-pad to 512 ->
// real FFT -> |X|^2 -> banded mel -> ln(max(.,floor)) -> (x+shift)*scale, stored time-major.
//
// Mapping: 16 lanes own one frame, so a wave64 carries 4 frames through the whole pipeline.
//   * the 512-point real FFT is a 256-point complex FFT of z[n] = y[2n] + i*y[2n+1], split 16 x 16:
//     lane b runs a register-resident 16-point FFT over z[16a+b], twiddles by W256^(b*c), the 16
//     lanes transpose through a padded LDS tile, lane c runs the second 16-point FFT and holds
//     Z[c+16d];
//   * Z goes back to LDS in natural order, each lane picks up the pairs (k, 256-k) it needs for the
//     real-FFT recombination, shares E = Z[k]+conj(Z[256-k]) and D = Z[k]-conj(Z[256-k]) between the
//     two bins, and leaves |X|^2 in LDS;
//   * the mel projection walks the bands 16 filters at a time (one filter per lane) with zero-padded
//     per-pass tap tables, so the spectrum never leaves the CU.
// HBM traffic is the algorithmic minimum: every kept sample is read once (8-byte lane loads that
// cover each frame's 128-byte lines exactly once) and every output float is written once.
#include "mmfe_internal.h"
#include "fft_regs.h"

#include <algorithm>
#include <cmath>

namespace mmfe {

namespace {

constexpr int kFrameLds = 16 * 17 * 8;       // bytes of LDS work space per frame (padded 16x16 complex tile)
constexpr int kFrameFloats = kFrameLds / 4;
constexpr int kWavesPerBlock = 4;
constexpr int kTwPostEntries = 144;          // -i*W512^k for k = 0..143
constexpr int kMaxPass = MelPassTable::kMaxPass;

struct FbankArgs {
    RowSource src;
    float* out;
    int n_rows;
    int frames_valid;      // frames computed per row (<= natural frame count, <= out_rows)
    int out_rows;          // frames per row in the output tensor
    int hop;
    int win_len;
    int n_mel;
    int n_pass;
    int mel_rows;          // total tap rows (of 16 weights)
    int niter[kMaxPass];
    int woff[kMaxPass];
    float preemph;
    int remove_dc;
    float inv_win_len;
    int log_kind;
    float log_floor;
    float log_of_floor;
    float out_shift, out_scale;
    const float* window;       // 512 floats, pre-halved, zero beyond win_len
    const float2* tw_stage;    // 256: W256^n
    const float2* tw_post;     // 144: -i*W512^k
    const int32_t* mel_start;  // n_pass*16
    const float* mel_w;        // mel_rows*16
};

// Loads the 2-sample slices (32a+2b, 32a+2b+1), a = 0..NA-1, of one frame.
template <int NA, bool VEC2>
__device__ __forceinline__ void load_frame(float2 (&v)[NA], const float* fp, bool valid, int b, int win_len) {
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int idx = 32 * a + 2 * b;
        float2 r = make_float2(0.f, 0.f);
        if (VEC2) {
            if (valid && idx + 1 < win_len) r = *reinterpret_cast<const float2*>(fp + idx);
            else if (valid && idx < win_len) r.x = fp[idx];
        } else {
            if (valid && idx < win_len) r.x = fp[idx];
            if (valid && idx + 1 < win_len) r.y = fp[idx + 1];
        }
        v[a] = r;
    }
}

// `log_of_floor` is log(floor) rounded on the host, so floored cells carry the same value the
// reference's CPU libm produces (device logf is within 1 ulp, not correctly rounded).
__device__ __forceinline__ float apply_log(float x, int kind, float floor_, float log_of_floor) {
    switch (kind) {
        case MMFE_LOG_LN_MAX: return x <= floor_ ? log_of_floor : logf(x);
        case MMFE_LOG_LN_ADD: return logf(x + floor_);
        case MMFE_LOG_LOG10_MAX: return x <= floor_ ? log_of_floor : log10f(x);
        default: return x;
    }
}

template <int NA /* ceil(win_len/32) */, bool VEC2>
__global__ void __launch_bounds__(64 * kWavesPerBlock, 3) fbank512_kernel(const FbankArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* s_tw_stage = reinterpret_cast<float2*>(smem);                      // 256
    float2* s_tw_post = s_tw_stage + 256;                                      // 144
    float* s_win = reinterpret_cast<float*>(s_tw_post + kTwPostEntries);       // 512
    int32_t* s_mel_start = reinterpret_cast<int32_t*>(s_win + 512);            // kMaxPass*16
    float* s_mel_w = reinterpret_cast<float*>(s_mel_start + kMaxPass * 16);    // mel_rows*16
    char* s_work = reinterpret_cast<char*>(s_mel_w + p.mel_rows * 16);         // 16-byte aligned: rows of 16 floats

    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_tw_stage[i] = p.tw_stage[i];
    for (int i = threadIdx.x; i < kTwPostEntries; i += blockDim.x) s_tw_post[i] = p.tw_post[i];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) s_win[i] = p.window[i];
    for (int i = threadIdx.x; i < p.n_pass * 16; i += blockDim.x) s_mel_start[i] = p.mel_start[i];
    for (int i = threadIdx.x; i < p.mel_rows * 16; i += blockDim.x) s_mel_w[i] = p.mel_w[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane >> 4;      // frame slot within the wave
    const int b = lane & 15;       // lane within the frame group
    float2* fbuf = reinterpret_cast<float2*>(s_work + (wave * 4 + fl) * kFrameLds);
    float* pbuf = reinterpret_cast<float*>(fbuf);

    const unsigned quads_per_row = (unsigned)(p.frames_valid + 3) >> 2;
    const unsigned total_quads = (unsigned)p.n_rows * quads_per_row;
    const unsigned qstride = gridDim.x * kWavesPerBlock;
    unsigned quad = blockIdx.x * kWavesPerBlock + wave;

    // frame owned by this lane group inside quad q (q is wave-uniform)
    auto locate = [&](unsigned q, unsigned& row, int& t) -> bool {
        row = q / quads_per_row;
        t = (int)(q - row * quads_per_row) * 4 + fl;
        return q < total_quads && t < p.frames_valid;
    };

    float2 nxt[NA];
    {
        unsigned row; int t;
        const bool valid = locate(quad, row, t);
        const float* fp = valid ? row_ptr(p.src, row) + (int64_t)t * p.hop : nullptr;
        load_frame<NA, VEC2>(nxt, fp, valid, b, p.win_len);
    }

    for (; quad < total_quads; quad += qstride) {
        unsigned row; int t;
        const bool valid = locate(quad, row, t);

        float2 g[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) g[a] = nxt[a];

        {   // prefetch the next quad of this wave; the loads stay in flight under the arithmetic below
            unsigned nrow; int nt;
            const bool nvalid = locate(quad + qstride, nrow, nt);
            const float* nfp = nvalid ? row_ptr(p.src, nrow) + (int64_t)nt * p.hop : nullptr;
            load_frame<NA, VEC2>(nxt, nfp, nvalid, b, p.win_len);
        }

        // ---- per-frame DC removal (kaldi remove_dc_offset) ------------------------------------
        if (p.remove_dc) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < NA; ++a) s += g[a].x + g[a].y;
            const float mean = row16_sum(s) * p.inv_win_len;
#pragma unroll
            for (int a = 0; a < NA; ++a) { g[a].x -= mean; g[a].y -= mean; }
        }

        // ---- pre-emphasis y[j] = g[j] - c*g[max(j-1,0)], then window.  The window table is zero
        //      beyond win_len, which also clears the lanes that sit in the zero padding. ---------
        float2 y[16];
        {
            const float c = p.preemph;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                // the sample before g[a].x lives in lane b-1 (.y, same a); for lane 0 it is lane 15's
                // .y of slice a-1; the very first sample of the frame uses itself (replicate padding)
                float edge;
                if (a == 0) edge = g[0].x;
                else edge = dpp_mov<DPP_ROW_ROR1>(g[a - 1 < 0 ? 0 : a - 1].y);
                const float prev = dpp_upd<DPP_ROW_SHR1>(edge, g[a].y);
                const float y0 = g[a].x - c * prev;
                const float y1 = g[a].y - c * g[a].x;
                const float2 w = *reinterpret_cast<const float2*>(s_win + 32 * a + 2 * b);
                y[a] = make_float2(y0 * w.x, y1 * w.y);
            }
#pragma unroll
            for (int a = NA; a < 16; ++a) y[a] = make_float2(0.f, 0.f);
        }

        // ---- 256-point complex FFT, first half: 16-point FFT over a, twiddle, transpose --------
        fft16(y);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 val = y[fft16_pos(c)];
            if (c > 0) val = cmul(val, s_tw_stage[b * c]);
            fbuf[c * 17 + b] = val;
        }
        wave_lds_fence();
        float2 u[16];
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) u[bb] = fbuf[b * 17 + bb];
        wave_lds_fence();

        // ---- second half: lane c = b now owns Z[c + 16 d]; park Z in natural order --------------
        fft16(u);
#pragma unroll
        for (int d = 0; d < 16; ++d) fbuf[b + 16 * d] = u[fft16_pos(d)];
        wave_lds_fence();

        // ---- real-FFT recombination on pairs (k, 256-k), power spectrum ------------------------
        float pk[9], pq[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = b + 16 * i;                   // i == 8 is only meaningful for b == 0 (k = 128)
            const float2 zk = fbuf[k];
            const float2 zp = fbuf[(256 - k) & 255];
            const float2 tw = s_tw_post[k];
            const float2 E = make_float2(zk.x + zp.x, zk.y - zp.y);
            const float2 D = make_float2(zk.x - zp.x, zk.y + zp.y);
            const float2 G = cmul(tw, D);
            const float2 xa = cadd(E, G), xb = csub(E, G);
            pk[i] = xa.x * xa.x + xa.y * xa.y;          // |X[k]|^2
            pq[i] = xb.x * xb.x + xb.y * xb.y;          // |X[256-k]|^2
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = b + 16 * i;
            pbuf[k] = pk[i];
            pbuf[256 - k] = pq[i];
        }
        if (b == 0) pbuf[128] = pk[8];
        wave_lds_fence();

        // ---- banded mel projection, log, affine, store ----------------------------------------
        float* orow = p.out + ((int64_t)row * p.out_rows + t) * p.n_mel;
        for (int j = 0; j < p.n_pass; ++j) {
            const int m = 16 * j + b;
            const float* pp = pbuf + s_mel_start[m];
            const float* wp = s_mel_w + p.woff[j] * 16 + b;
            const int n = p.niter[j];
            float acc = 0.f;
            for (int i = 0; i < n; ++i) acc = fmaf(wp[16 * i], pp[i], acc);
            float val = apply_log(acc, p.log_kind, p.log_floor, p.log_of_floor);
            val = (val + p.out_shift) * p.out_scale;
            if (valid && m < p.n_mel) orow[m] = val;
        }
        wave_lds_fence();
    }
}

template <int NA, bool VEC2>
int launch_variant(const FbankArgs& args, int grid, size_t lds, hipStream_t stream) {
    static bool attr_set = false;   // raising the dynamic-LDS limit is idempotent; racing threads set the same value
    if (!attr_set) {
        MMFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fbank512_kernel<NA, VEC2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((fbank512_kernel<NA, VEC2>), dim3(grid), dim3(64 * kWavesPerBlock), lds, stream, args);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace

int fbank512_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    if (s.n_fft != 512 || s.win_off != 0 || s.win_len > 512 || s.win_len < 2) {
        set_error("fbank512: needs n_fft == 512, win_off == 0, 2 <= win_len <= 512");
        return MMFE_EINVAL;
    }
    if (s.n_mel <= 0 || s.n_mel > 16 * kMaxPass) {
        set_error("fbank512: n_mel must be in [1, %d]", 16 * kMaxPass);
        return MMFE_EINVAL;
    }
    // window: pre-halved (folds the 1/2 of the real-FFT recombination; exact in binary fp), zero padded
    std::vector<float> win(512, 0.f);
    for (int i = 0; i < s.win_len; ++i) win[i] = 0.5f * plan.window[i];
    int rc = upload(&plan.d_window, win);
    if (rc) return rc;
    std::vector<float> tws(2 * 256), twp(2 * kTwPostEntries);
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < 256; ++n) {
        tws[2 * n] = (float)std::cos(2.0 * pi * n / 256.0);
        tws[2 * n + 1] = (float)-std::sin(2.0 * pi * n / 256.0);
    }
    for (int k = 0; k < kTwPostEntries; ++k) {   // -i * exp(-2 pi i k / 512)
        twp[2 * k] = (float)-std::sin(2.0 * pi * k / 512.0);
        twp[2 * k + 1] = (float)-std::cos(2.0 * pi * k / 512.0);
    }
    if ((rc = upload(&plan.d_tw_a, tws))) return rc;
    if ((rc = upload(&plan.d_tw_b, twp))) return rc;

    // mel taps, 16 filters per pass
    MelPassTable& mt = plan.mel16;
    mt.n_pass = (s.n_mel + 15) / 16;
    mt.start.assign(mt.n_pass * 16, 0);
    int rows = 0;
    std::vector<int> boff(s.n_mel + 1, 0);
    for (int m = 0; m < s.n_mel; ++m) boff[m + 1] = boff[m] + plan.band_len[m];
    for (int j = 0; j < mt.n_pass; ++j) {
        int longest = 0;
        for (int l = 0; l < 16; ++l) {
            const int m = 16 * j + l;
            if (m < s.n_mel) longest = std::max(longest, (int)plan.band_len[m]);
        }
        mt.niter[j] = longest;
        mt.woff[j] = rows;
        rows += longest;
    }
    mt.total_rows = rows;
    mt.weights.assign((size_t)std::max(rows, 1) * 16, 0.f);
    for (int j = 0; j < mt.n_pass; ++j) {
        for (int l = 0; l < 16; ++l) {
            const int m = 16 * j + l;
            if (m >= s.n_mel) continue;
            int st = plan.band_start[m];
            const int len = plan.band_len[m];
            if (len > 0 && (st < 0 || st + len > 257)) {
                set_error("fbank512: mel band %d = [%d, %d) leaves the 257-bin spectrum", m, st, st + len);
                return MMFE_EINVAL;
            }
            if (len == 0) st = 0;
            // every lane walks niter[j] taps: the walk must stay inside the frame's LDS work space
            if (st + mt.niter[j] > kFrameFloats) {
                set_error("fbank512: mel pass %d too long", j);
                return MMFE_EINVAL;
            }
            mt.start[m] = st;
            for (int i = 0; i < len; ++i) mt.weights[(size_t)(mt.woff[j] + i) * 16 + l] = plan.band_weights[boff[m] + i];
        }
    }
    if ((rc = upload(&plan.d_mel_start, mt.start))) return rc;
    if ((rc = upload(&plan.d_mel_w, mt.weights))) return rc;
    return MMFE_OK;
}

int fbank512_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                    int64_t out_frames, float* out, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    const int64_t frames_valid = std::min(n_frames_natural, out_frames);
    if (n_rows <= 0 || frames_valid <= 0) return MMFE_OK;
    const int64_t quads = n_rows * ((frames_valid + 3) / 4);
    if (quads >= (int64_t)1 << 31 || n_rows >= (int64_t)1 << 31) {
        set_error("fbank512: batch too large for one launch (%lld frame quads)", (long long)quads);
        return MMFE_EINVAL;
    }
    FbankArgs a{};
    a.src = src;
    a.out = out;
    a.n_rows = (int)n_rows;
    a.frames_valid = (int)frames_valid;
    a.out_rows = (int)out_frames;
    a.hop = s.hop;
    a.win_len = s.win_len;
    a.n_mel = s.n_mel;
    a.n_pass = plan.mel16.n_pass;
    a.mel_rows = std::max(plan.mel16.total_rows, 1);
    for (int j = 0; j < kMaxPass; ++j) { a.niter[j] = plan.mel16.niter[j]; a.woff[j] = plan.mel16.woff[j]; }
    a.preemph = s.preemph;
    a.remove_dc = s.remove_dc;
    a.inv_win_len = 1.0f / (float)s.win_len;
    a.log_kind = s.log_kind;
    a.log_floor = s.log_floor;
    a.log_of_floor = s.log_kind == MMFE_LOG_LOG10_MAX ? (float)std::log10((double)s.log_floor) : (float)std::log((double)s.log_floor);
    a.out_shift = s.out_shift;
    a.out_scale = s.out_scale;
    a.window = plan.d_window;
    a.tw_stage = reinterpret_cast<const float2*>(plan.d_tw_a);
    a.tw_post = reinterpret_cast<const float2*>(plan.d_tw_b);
    a.mel_start = plan.d_mel_start;
    a.mel_w = plan.d_mel_w;

    const size_t lds = 256 * 8 + kTwPostEntries * 8 + 512 * 4 + kMaxPass * 16 * 4 + (size_t)a.mel_rows * 64 +
                       (size_t)kWavesPerBlock * 4 * kFrameLds;
    if (lds > 160 * 1024) {
        set_error("fbank512: mel tap table does not fit in LDS (%zu bytes)", lds);
        return MMFE_EINVAL;
    }
    int dev = 0, cus = 256;
    MMFE_HIP_CHECK(hipGetDevice(&dev));
    MMFE_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t want = (quads + kWavesPerBlock - 1) / kWavesPerBlock;
    const int grid = (int)std::min<int64_t>(want, (int64_t)cus * 3);

    // 8-byte lane loads need every frame start to be 8-byte aligned
    bool vec2 = (s.hop % 2 == 0) && (reinterpret_cast<uintptr_t>(src.base) % 8 == 0);
    if (src.offsets == nullptr) vec2 = vec2 && (src.stride % 2 == 0);
    else vec2 = false;   // slice offsets are data: take the scalar-load variant
    const int na = (s.win_len + 31) / 32;
    if (na <= 13) return vec2 ? launch_variant<13, true>(a, grid, lds, stream) : launch_variant<13, false>(a, grid, lds, stream);
    return vec2 ? launch_variant<16, true>(a, grid, lds, stream) : launch_variant<16, false>(a, grid, lds, stream);
}

}  // namespace mmfe

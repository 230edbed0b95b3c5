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
// registers, and writes its 17 magnitudes into an LDS image of the group's output block.  That
// block is contiguous in global memory, so it is drained with 16-byte coalesced stores: the kernel
// is bound by HBM writes (81 % of its traffic), and partial-line stores would waste them.
//
// Two kernels share that arithmetic.  stft32_ws_kernel (packed batches, hop a multiple of 4: the regular case) splits the
// waves of a block by role: compute waves turn the staged rows of group g into the output image of group g while a
// few mover waves store the image of group g - 1 and stage the samples of group g + 1, both buffers doubled in LDS,
// one barrier per group.  A wave that waits for the memory system to accept its stores then never holds up
// arithmetic, and the stores of a CU leave as one steady stream.  stft32_kernel (any alignment, slices) runs the
// phases one after the other with every wave taking part in each.
#include "mmfe_internal.h"
#include "fft_regs.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace mmfe {

namespace {

constexpr int kN = 32;
constexpr int kMaxThreads = 1024;

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
    int vec_in;          // rows are contiguous, 16-byte aligned and n_samples % 4 == 0: stage with 16-byte loads
    int vec_out;         // every group's output block starts 16-byte aligned: drain with 16-byte stores
    int mover_waves;     // stft32_ws_kernel: the last `mover_waves` waves of a block store and stage, the others compute
    unsigned inv_frames; // ceil(2^32 / n_frames): (f * inv_frames) >> 32 == f / n_frames for the frame indices of a group
    unsigned inv_samples; // ceil(2^32 / n_samples), for the sample indices of a group (stft32_ws_kernel, n_samples % 4 != 0)
    float win[kN];       // window pre-multiplied by 0.5 * spec_scale
};

// -i * exp(-2 pi i k / 32), k = 0..8
__device__ __forceinline__ float2 tw32(int k) {
    constexpr float S[9] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                            0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.f};
    return make_float2(-S[k], -S[8 - k]);
}

constexpr int kPrefetch = 4;     // float4 registers per thread for the next group's samples

// One frame: 32 samples at xp (LDS) -> 17 magnitudes (or powers) at o[k * S] (LDS image of the output block).
// `o` is addressed with 32-bit offsets (k * S is wave-uniform), the magnitude / power choice is a compile-time constant: as run-time
// values they cost a 64-bit multiply-add per stored bin and a select per stored bin.
template <bool ALIGNED_HOP, bool MAGNITUDE>
__device__ __forceinline__ void stft32_frame(const StftArgs& p, const float* xp, float* o, unsigned S) {
    float x[32];
    if (ALIGNED_HOP) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(xp + 4 * q);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int n = 0; n < 32; ++n) x[n] = xp[n];
    }
    float2 z[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) z[n] = make_float2(x[2 * n] * p.win[2 * n], x[2 * n + 1] * p.win[2 * n + 1]);
    fft16(z);
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float* ol = (lds_float*)o;          // LDS pointer: 32-bit address arithmetic
    // the optimiser must not see how `ol` was formed: it would fold k into the row term of the index ((17 r + k) * S + s) and pay a
    // 64-bit multiply-add per stored bin, where k * S is a scalar and the sum is one addition
    asm volatile("" : "+v"(ol));
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
        if (MAGNITUDE) { pa = __builtin_amdgcn_sqrtf(pa); pb = __builtin_amdgcn_sqrtf(pb); }
        ol[(unsigned)k * S] = pa;
        if (k != 8) ol[(unsigned)(16 - k) * S] = pb;
    }
}


// One group = R consecutive rows, three phases:
//   stage  the group's samples (prefetched into registers two groups ahead, coalesced 16-byte reads) are
//          dropped into LDS rows whose zero margins (scipy's boundary extension + padding) are written once;
//          this happens between the previous group's work and drain, so neither the loads nor the
//          drain's stores are waited for right after they are issued
//   work   one lane per frame: eight 16-byte LDS reads, 32-point real FFT in registers, 17 magnitudes
//          written to an LDS image of the group's output block, laid out exactly like global memory
//   drain  the output block is contiguous in global memory: 16-byte coalesced stores, left in flight while
//          the next group is staged and worked on (barriers order LDS only)
template <bool ALIGNED_HOP>
__global__ void __launch_bounds__(kMaxThreads) stft32_kernel(const StftArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int R = p.rows_per_block;
    const int S = p.n_frames;
    const int T = p.n_samples;
    const int T4 = T >> 2;
    float* s_in = reinterpret_cast<float*>(smem);                    // R * row_lds
    float* s_out = s_in + R * p.row_lds;                             // R * 17 * S
    const int n_groups = (p.n_rows + R - 1) / R;
    const int nthr = blockDim.x;
    const bool prefetch = p.vec_in && R * T4 <= kPrefetch * nthr;

    for (int i = threadIdx.x; i < R * (p.row_lds - T); i += nthr) {   // zero margins: never overwritten
        const int r = i / (p.row_lds - T);
        int j = i - r * (p.row_lds - T);
        j = j < 16 ? j : j + T;
        s_in[r * p.row_lds + j] = 0.f;
    }

    float4 pf[kPrefetch];
    auto fetch = [&](int grp) {
        const int row0 = grp * R;
        const int rows = min(R, p.n_rows - row0);
        const float4* g4 = reinterpret_cast<const float4*>(static_cast<const float*>(p.src.base) + (int64_t)row0 * T);
#pragma unroll
        for (int q = 0; q < kPrefetch; ++q) {
            const int i = threadIdx.x + q * nthr;
            pf[q] = (grp < n_groups && i < rows * T4) ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // put group `grp`'s samples into the LDS rows (from the prefetch registers when they hold them)
    auto stage = [&](int grp) {
        if (grp >= n_groups) return;
        const int row0 = grp * R;
        const int rows = min(R, p.n_rows - row0);
        if (prefetch) {
#pragma unroll
            for (int q = 0; q < kPrefetch; ++q) {
                const int i = threadIdx.x + q * nthr;
                if (i < rows * T4) {
                    const int r = i / T4;
                    const int j4 = i - r * T4;
                    *reinterpret_cast<float4*>(s_in + r * p.row_lds + 16 + 4 * j4) = pf[q];
                }
            }
        } else if (p.vec_in) {
            const float4* g4 = reinterpret_cast<const float4*>(static_cast<const float*>(p.src.base) + (int64_t)row0 * T);
            for (int i = threadIdx.x; i < rows * T4; i += nthr) {
                const int r = i / T4;
                const int j4 = i - r * T4;
                *reinterpret_cast<float4*>(s_in + r * p.row_lds + 16 + 4 * j4) = g4[i];
            }
        } else {
            for (int i = threadIdx.x; i < rows * T; i += nthr) {
                const int r = i / T;
                const int j = i - r * T;
                s_in[r * p.row_lds + 16 + j] = row_ptr(p.src, row0 + r)[j];
            }
        }
    };
    if (prefetch) fetch(blockIdx.x);
    stage(blockIdx.x);
    if (prefetch) fetch(blockIdx.x + gridDim.x);
    lds_barrier();

    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int row0 = grp * R;
        const int rows = min(R, p.n_rows - row0);
        // ---- work: one frame per lane ---------------------------------------------------------------
        const int n_fr = rows * S;
        for (int f = threadIdx.x; f < n_fr; f += nthr) {
            const int r = f / S;
            const int s = f - r * S;
            if (p.power == 1) stft32_frame<ALIGNED_HOP, true>(p, s_in + r * p.row_lds + s * p.hop, s_out + r * 17 * S + s, (unsigned)S);
            else stft32_frame<ALIGNED_HOP, false>(p, s_in + r * p.row_lds + s * p.hop, s_out + r * 17 * S + s, (unsigned)S);    // frame s covers samples [s*hop-16, s*hop+16)
        }
        lds_barrier();      // output image complete; every wave is done reading s_in
        // ---- stage the next group (its loads were issued a whole work phase ago), refill the prefetch --------
        stage(grp + gridDim.x);
        if (prefetch) fetch(grp + 2 * gridDim.x);
        // ---- drain ----------------------------------------------------------------------------------
        const int n_out = rows * 17 * S;
        float* gout = p.out + (int64_t)row0 * 17 * S;
        if (p.vec_out) {
            // the output tensor is far larger than the caches and is not read back here: stream it past them
            typedef float vf4 __attribute__((ext_vector_type(4)));
            const int n4 = n_out >> 2;
            vf4* g4 = reinterpret_cast<vf4*>(gout);
            const vf4* l4 = reinterpret_cast<const vf4*>(s_out);
            for (int i = threadIdx.x; i < n4; i += nthr) __builtin_nontemporal_store(l4[i], g4 + i);
            for (int i = 4 * n4 + threadIdx.x; i < n_out; i += nthr) gout[i] = s_out[i];
        } else {
            for (int i = threadIdx.x; i < n_out; i += nthr) gout[i] = s_out[i];
        }
        lds_barrier();      // next group's samples visible; every wave is done reading s_out
    }
}

#ifndef MMFE_STFT_STAGE_FIRST
#define MMFE_STFT_STAGE_FIRST 1
#endif
constexpr int kWsPrefetch = 6;   // float4 registers per mover lane for the samples of the group after next

// Wave-specialised form (rows packed back to back from a 16-byte aligned base, hop a multiple of 4, group output blocks 16-byte
// aligned).
// LDS holds two input buffers and two output images.  In step t a block's compute waves work on its t-th group
// (input buffer t & 1 -> image t & 1) while its mover waves store image (t - 1) & 1, drop the samples of group t + 1
// (requested one step earlier) into input buffer (t + 1) & 1 and request those of group t + 2; one barrier ends the step.
template <bool NT, bool MAGNITUDE, bool SPLIT /* n_samples % 4 != 0: a 16-byte piece may straddle two rows */>
__global__ void __launch_bounds__(kMaxThreads) stft32_ws_kernel(const StftArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const int R = p.rows_per_block;
    const int S = p.n_frames;
    const int T = p.n_samples;
    const int in_floats = R * p.row_lds;
    const int out_floats = R * 17 * S;
    float* s_in0 = reinterpret_cast<float*>(smem);                   // 2 * R * row_lds
    float* s_out0 = s_in0 + 2 * in_floats;                           // 2 * R * 17 * S
    const int n_groups = (p.n_rows + R - 1) / R;
    const int n_compute = blockDim.x - 64 * p.mover_waves;           // threads that compute
    const int n_mover = 64 * p.mover_waves;
    const bool mover = (int)threadIdx.x >= n_compute;
    int ml = (int)threadIdx.x - n_compute;                           // mover lane index
    const float* gin = static_cast<const float*>(p.src.base);

    for (int i = threadIdx.x; i < 2 * R * (p.row_lds - T); i += blockDim.x) {   // zero margins of both input buffers: never overwritten
        const int r = i / (p.row_lds - T);
        int j = i - r * (p.row_lds - T);
        j = j < 16 ? j : j + T;
        s_in0[r * p.row_lds + j] = 0.f;
    }

    // A group's samples are one contiguous run of rows * T floats that starts 16-byte aligned (R is a multiple of 4), whatever T
    // is; lanes take whole 16-byte pieces of the run.  With T % 4 != 0 a piece may straddle two rows, and the batch's last piece
    // may be partial: that one is read float by float when it is staged, so that nothing is read beyond the batch.
    constexpr bool rows_split_pieces = SPLIT;
    vf4 pf[kWsPrefetch];
    auto fetch = [&](int grp) {                                      // movers: request group grp's samples
        const int row0 = grp * R;
        const int n_el = grp < n_groups ? min(R, p.n_rows - row0) * T : 0;
        const vf4* g4 = reinterpret_cast<const vf4*>(gin + (int64_t)row0 * T);
#pragma unroll
        for (int q = 0; q < kWsPrefetch; ++q) {
            const int i = ml + q * n_mover;
            pf[q] = 4 * i + 4 <= n_el ? __builtin_nontemporal_load(g4 + i) : vf4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](int grp, float* s_in) {                         // movers: registers -> LDS rows
        if (grp >= n_groups) return;
        const int n_el = min(R, p.n_rows - grp * R) * T;
#pragma unroll
        for (int q = 0; q < kWsPrefetch; ++q) {
            const int e = 4 * (ml + q * n_mover);
            if (e >= n_el) continue;
            if constexpr (!rows_split_pieces) {
                const unsigned r = __umulhi((unsigned)e, p.inv_samples);
                *reinterpret_cast<vf4*>(s_in + r * (unsigned)p.row_lds + 16 + ((unsigned)e - r * (unsigned)T)) = pf[q];
            } else {
                unsigned r = __umulhi((unsigned)e, p.inv_samples);
                unsigned j = (unsigned)e - r * (unsigned)T;
                float* d = s_in + r * (unsigned)p.row_lds + 16;
                if (e + 4 > n_el) {                                  // the batch's last, partial piece (one lane, once per launch)
                    const float* g = gin + (int64_t)grp * R * T + e;
                    for (int k = 0; k < n_el - e; ++k) {
                        d[j] = g[k];
                        if (++j == (unsigned)T) { j = 0; d += p.row_lds; }
                    }
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d[j] = pf[q][k];
                    if (++j == (unsigned)T) { j = 0; d += p.row_lds; }
                }
            }
        }
    };
    if (mover) {
        fetch(blockIdx.x);
        stage(blockIdx.x, s_in0);
        fetch(blockIdx.x + gridDim.x);
    }
    lds_barrier();

    // one step beyond the block's last group: the movers store its image
    for (int grp = blockIdx.x, t = 0; grp - (int)gridDim.x < n_groups; grp += gridDim.x, ++t) {
        if (!mover) {
            if (grp < n_groups) {
                const int rows = min(R, p.n_rows - grp * R);
                const float* s_in = s_in0 + (t & 1) * in_floats;
                float* s_out = s_out0 + (t & 1) * out_floats;
                const int n_fr = rows * S;
                for (int f = threadIdx.x; f < n_fr; f += n_compute) {
                    const unsigned r = __umulhi((unsigned)f, p.inv_frames);
                    const unsigned fs = (unsigned)f - r * (unsigned)S;
                    stft32_frame<true, MAGNITUDE>(p, s_in + (r * (unsigned)p.row_lds + fs * (unsigned)p.hop), s_out + (r * 17u * (unsigned)S + fs), (unsigned)S);
                }
            }
        } else {
            // A mover lane's piece indices, LDS rows and offsets are the same in every step.  Left visible, the optimiser keeps them
            // all in registers across the loop (a dozen values per lane, spilled to scratch at this block size); recomputed per step
            // they cost one multiply-high per piece.
            asm volatile("" : "+v"(ml));
#if MMFE_STFT_STAGE_FIRST
            // The samples requested one step ago are dropped into LDS and the next request goes out BEFORE the previous image is stored:
            // loads and stores return in order, so the wait for the samples would otherwise also be a wait for stores issued a moment
            // before it; in this order everything that wait covers is a whole step old.
            stage(grp + gridDim.x, s_in0 + ((t + 1) & 1) * in_floats);
            fetch(grp + 2 * gridDim.x);
#endif
            if (t > 0) {                                              // store the previous group's image
                const int prev = grp - (int)gridDim.x;
                const int rows = min(R, p.n_rows - prev * R);
                const int n_out = rows * 17 * S;
                const float* s_out = s_out0 + ((t - 1) & 1) * out_floats;
                float* gout = p.out + (int64_t)prev * R * 17 * S;
                const int n4 = n_out >> 2;
                vf4* g4 = reinterpret_cast<vf4*>(gout);
                const vf4* l4 = reinterpret_cast<const vf4*>(s_out);
                int i = ml;
                for (; i + 3 * n_mover < n4; i += 4 * n_mover) {      // four 16-byte LDS reads in flight per lane
                    const vf4 a = l4[i], b = l4[i + n_mover], c = l4[i + 2 * n_mover], d = l4[i + 3 * n_mover];
                    if (NT) {
                        __builtin_nontemporal_store(a, g4 + i);
                        __builtin_nontemporal_store(b, g4 + i + n_mover);
                        __builtin_nontemporal_store(c, g4 + i + 2 * n_mover);
                        __builtin_nontemporal_store(d, g4 + i + 3 * n_mover);
                    } else {
                        g4[i] = a; g4[i + n_mover] = b; g4[i + 2 * n_mover] = c; g4[i + 3 * n_mover] = d;
                    }
                }
                for (; i < n4; i += n_mover) { if (NT) __builtin_nontemporal_store(l4[i], g4 + i); else g4[i] = l4[i]; }
                for (int j = 4 * n4 + ml; j < n_out; j += n_mover) gout[j] = s_out[j];
            }
#if !MMFE_STFT_STAGE_FIRST
            stage(grp + gridDim.x, s_in0 + ((t + 1) & 1) * in_floats);
            fetch(grp + 2 * gridDim.x);
#endif
        }
        lds_barrier();
    }
}

}  // namespace

int stft32_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    if (s.n_fft != kN || s.win_len != kN || s.win_off != 0 || s.hop < 1 || s.hop > kN || s.n_mel != 0 ||
        s.log_kind != MMFE_LOG_NONE || s.layout != MMFE_LAYOUT_FEAT_MAJOR || s.frames_mode != MMFE_FRAMES_SCIPY ||
        s.remove_dc || s.preemph != 0.f || s.out_frames != 0 || s.out_shift != 0.f || s.out_scale != 1.f ||
        s.pad_mode != MMFE_PAD_ZERO || s.clamp_below_max > 0.f || s.drop_last) {
        set_error("stft32: needs n_fft == win_len == 32, 1 <= hop <= 32, scipy framing with zero extension, feature-major magnitude/power "
                  "output, no mel/log/affine/clamp, natural frame count");
        return MMFE_EINVAL;
    }
    // tuning knobs, read once per plan: rows per group of the wave-specialised form, its mover waves (-1: phase-serial form)
    if (const char* e = std::getenv("MMFE_STFT_ROWS")) plan.tune_blocks_per_cu = std::atoi(e);
    if (const char* e = std::getenv("MMFE_STFT_MOVERS")) plan.tune_chunk = std::atoi(e);
    if (const char* e = std::getenv("MMFE_STFT_NT")) plan.tune_nt = std::atoi(e);              // 0: cached output stores
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
    const int64_t per_row_bytes = ((int64_t)a.row_lds + 17 * n_frames) * 4;
    const bool packed = src.offsets == nullptr && src.stride == n_samples && reinterpret_cast<uintptr_t>(src.base) % 16 == 0;
    const bool contiguous = packed && n_samples % 4 == 0;      // every row starts 16-byte aligned
    const bool aligned_hop = s.hop % 4 == 0;
    a.inv_frames = (unsigned)(((uint64_t)1 << 32) / (uint64_t)n_frames + 1);
    a.inv_samples = (unsigned)(((uint64_t)1 << 32) / (uint64_t)n_samples + 1);
    int blocks_per_cu = 1, cus = 256;

    // ---- wave-specialised form: two input buffers + two output images in LDS -------------------------------------
    // R: the largest multiple of 4 (16-byte aligned output blocks) whose doubled buffers fit; the compute waves cover a
    // group's R * S frames in as few passes as 16 - movers waves allow, the movers' prefetch registers must hold a group.
    if (packed && aligned_hop && reinterpret_cast<uintptr_t>(out) % 16 == 0 && n_rows < ((int64_t)1 << 30) && plan.tune_chunk >= 0) {
        int r_ws = (int)std::min<int64_t>(64, (150 * 1024) / (2 * per_row_bytes)) & ~3;
        if (plan.tune_blocks_per_cu >= 4 && plan.tune_blocks_per_cu % 4 == 0 && plan.tune_blocks_per_cu < r_ws) r_ws = plan.tune_blocks_per_cu;
        // (the reciprocal divisions need 2 <= S, R * S < 2^16, and the same of the samples)
        if (r_ws >= 4 && n_frames >= 2 && (int64_t)r_ws * n_frames < 65536 && n_samples >= 2 && (int64_t)r_ws * n_samples < 65536) {
            const int pass_waves = (int)(((int64_t)r_ws * n_frames + 63) / 64);       // compute waves for one pass
            int movers = pass_waves <= 12 ? 4 : (pass_waves == 13 ? 3 : (pass_waves == 14 ? 2 : 4));
            if (plan.tune_chunk >= 1 && plan.tune_chunk <= 8) movers = plan.tune_chunk;
            const int compute = std::min(pass_waves, 16 - movers);
            if (((int64_t)r_ws * n_samples + 3) / 4 <= (int64_t)kWsPrefetch * 64 * movers) {
                a.rows_per_block = r_ws;
                a.mover_waves = movers;
                a.vec_in = a.vec_out = 1;
                const int threads = 64 * (compute + movers);
                const size_t lds = (size_t)(2 * r_ws * per_row_bytes);
                const bool nt = plan.tune_nt != 0, mag = s.power == 1, split = n_samples % 4 != 0;
                typedef void (*ws_fn)(const StftArgs);
                const ws_fn kfn = split ? (nt ? (mag ? &stft32_ws_kernel<true, true, true> : &stft32_ws_kernel<true, false, true>)
                                              : (mag ? &stft32_ws_kernel<false, true, true> : &stft32_ws_kernel<false, false, true>))
                                        : (nt ? (mag ? &stft32_ws_kernel<true, true, false> : &stft32_ws_kernel<true, false, false>)
                                              : (mag ? &stft32_ws_kernel<false, true, false> : &stft32_ws_kernel<false, false, false>));
                const int rc = kernel_prepare(reinterpret_cast<const void*>(kfn), threads, lds, &blocks_per_cu, &cus);
                if (rc) return rc;
                const int64_t groups = (n_rows + r_ws - 1) / r_ws;
                const int grid = (int)std::min<int64_t>(groups, (int64_t)cus * blocks_per_cu);
                hipLaunchKernelGGL(kfn, dim3(grid), dim3(threads), lds, stream, a);
                MMFE_HIP_CHECK(hipGetLastError());
                return MMFE_OK;
            }
        }
    }

    // ---- phase-serial form ------------------------------------------------------------------------------------------
    // rows per group R: a multiple of 4 keeps every group's output block 16-byte aligned; the largest group that fits LDS
    // measured fastest (fewest barriers per byte), one block per CU.
    int best_r = 0, best_threads = 0;
    for (int r = 4; r <= 64; r += 4) {
        if (r * per_row_bytes > 144 * 1024) break;
        best_r = r;
        best_threads = (int)std::min<int64_t>(kMaxThreads, ((int64_t)r * n_frames + 63) / 64 * 64);
    }
    if (best_r == 0) {   // very long rows: one row per group, frames looped over the block
        best_r = 1;
        best_threads = 256;
        if (per_row_bytes > 150 * 1024) {
            set_error("stft32: a row of %lld samples does not fit the LDS staging buffers", (long long)n_samples);
            return MMFE_EINVAL;
        }
    }
    a.rows_per_block = best_r;
    a.vec_in = contiguous ? 1 : 0;
    a.vec_out = (((int64_t)best_r * 17 * n_frames) % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 1 : 0;
    const size_t lds = (size_t)best_r * per_row_bytes;
    const int64_t groups = (n_rows + best_r - 1) / best_r;
    const void* fn = aligned_hop ? reinterpret_cast<const void*>(&stft32_kernel<true>) : reinterpret_cast<const void*>(&stft32_kernel<false>);
    // persistent blocks: exactly as many as are resident at once (registers, LDS and wave slots all count)
    const int rc = kernel_prepare(fn, best_threads, lds, &blocks_per_cu, &cus);
    if (rc) return rc;
    const int grid = (int)std::min<int64_t>(groups, (int64_t)cus * blocks_per_cu);
    if (aligned_hop) hipLaunchKernelGGL(stft32_kernel<true>, dim3(grid), dim3(best_threads), lds, stream, a);
    else hipLaunchKernelGGL(stft32_kernel<false>, dim3(grid), dim3(best_threads), lds, stream, a);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace mmfe

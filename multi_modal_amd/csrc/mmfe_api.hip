// C ABI of libmmfe.so: plan management, shape rules, dispatch to the kernel families.
#include "mmfe_internal.h"

#include <map>
#include <mutex>
#include <set>
#include <tuple>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

namespace mmfe {

int kernel_prepare(const void* fn, int threads, size_t lds, int* blocks_per_cu, int* cus) {
    struct Key {
        const void* fn; int dev; int threads; size_t lds;
        bool operator<(const Key& o) const { return std::tie(fn, dev, threads, lds) < std::tie(o.fn, o.dev, o.threads, o.lds); }
    };
    struct Val { int blocks; int cus; };
    static std::mutex mu;
    static std::map<Key, Val> cache;
    static std::set<std::pair<const void*, int>> raised;
    int dev = 0;
    MMFE_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    const Key key{fn, dev, threads, lds};
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (!raised.count({fn, dev})) {
            MMFE_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised.insert({fn, dev});
        }
        Val v{1, 256};
        MMFE_HIP_CHECK(hipDeviceGetAttribute(&v.cus, hipDeviceAttributeMultiprocessorCount, dev));
        MMFE_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v.blocks, fn, threads, lds));
        if (v.blocks < 1) v.blocks = 1;
        it = cache.emplace(key, v).first;
    }
    *blocks_per_cu = it->second.blocks;
    *cus = it->second.cus;
    return MMFE_OK;
}


static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {

int natural_frames(const mmfe_spec& s, int64_t n, int64_t* frames) {
    const int64_t f = natural_frame_count(s.frames_mode, s.win_len, s.hop, s.n_fft, n);
    if (f < 0) {
        set_error("unknown frames_mode %d", s.frames_mode);
        return MMFE_EINVAL;
    }
    *frames = f;
    return MMFE_OK;
}

int validate_spec(const mmfe_spec* s) {
    if (s == nullptr) { set_error("spec is NULL"); return MMFE_EINVAL; }
    if (s->struct_size != (int32_t)sizeof(mmfe_spec)) {
        set_error("mmfe_spec.struct_size is %d, this library expects %d", s->struct_size, (int)sizeof(mmfe_spec));
        return MMFE_EINVAL;
    }
    if (s->kind < MMFE_KIND_FBANK || s->kind > MMFE_KIND_RESAMPLE) { set_error("unknown kind %d", s->kind); return MMFE_EINVAL; }
    if (s->hop < 1) { set_error("hop must be >= 1"); return MMFE_EINVAL; }
    if (s->win_len < 1) { set_error("win_len must be >= 1"); return MMFE_EINVAL; }
    if (s->kind == MMFE_KIND_RESAMPLE) {
        if (s->n_fft < 1) { set_error("RESAMPLE: up (n_fft) must be >= 1"); return MMFE_EINVAL; }
    } else if (s->kind != MMFE_KIND_FRAME) {
        if (s->n_fft < 2) { set_error("n_fft must be >= 2"); return MMFE_EINVAL; }
        if (s->win_off < 0 || s->win_off + s->win_len > s->n_fft) { set_error("window does not fit in n_fft"); return MMFE_EINVAL; }
        if (s->power != 1 && s->power != 2) { set_error("power must be 1 or 2"); return MMFE_EINVAL; }
        if (s->n_mel < 0) { set_error("n_mel must be >= 0"); return MMFE_EINVAL; }
        if (s->log_kind < MMFE_LOG_NONE || s->log_kind > MMFE_LOG_LOG10_MAX) { set_error("unknown log_kind"); return MMFE_EINVAL; }
        if (s->layout != MMFE_LAYOUT_TIME_MAJOR && s->layout != MMFE_LAYOUT_FEAT_MAJOR) { set_error("unknown layout"); return MMFE_EINVAL; }
        if (s->pad_mode < MMFE_PAD_NONE || s->pad_mode > MMFE_PAD_REFLECT) { set_error("unknown pad_mode"); return MMFE_EINVAL; }
    }
    if (s->out_frames < 0) { set_error("out_frames must be >= 0"); return MMFE_EINVAL; }
    if (s->out_dtype < MMFE_OUT_F32 || s->out_dtype > MMFE_OUT_BF16) { set_error("unknown out_dtype"); return MMFE_EINVAL; }
    if (s->in_dtype < MMFE_IN_F32 || s->in_dtype > MMFE_IN_BF16) { set_error("unknown in_dtype"); return MMFE_EINVAL; }
    if (s->in_dtype == MMFE_IN_S16 && s->kind != MMFE_KIND_FBANK && s->kind != MMFE_KIND_LOGSPEC && s->kind != MMFE_KIND_RESAMPLE) {
        set_error("16-bit PCM input is implemented for the FBANK, LOGSPEC and RESAMPLE kinds only");
        return MMFE_EINVAL;
    }
    if (s->in_dtype > MMFE_IN_S16 && s->kind != MMFE_KIND_FRAME) {
        set_error("float64 / float16 / bfloat16 rows are taken by the FRAME kind only");
        return MMFE_EINVAL;
    }
    if (s->kind == MMFE_KIND_FRAME && (s->in_dtype == MMFE_IN_S16 || s->out_dtype != MMFE_OUT_F32)) {
        set_error("FRAME: rows are float32, float64, float16 or bfloat16 and the frames have the rows' type (out_dtype stays 0)");
        return MMFE_EINVAL;
    }
    return MMFE_OK;
}

int out_shape(const mmfe_spec& s, int64_t n, int64_t* natural, int64_t* frames, int64_t* feat) {
    if (s.kind == MMFE_KIND_RESAMPLE) {       // "frames" are output samples, one feature each
        *natural = resample_out_len(s, n);
        *frames = s.out_frames > 0 ? s.out_frames : *natural;
        *feat = 1;
        return MMFE_OK;
    }
    int64_t f = 0;
    int rc = natural_frames(s, n, &f);
    if (rc) return rc;
    if (s.drop_last && f > 0) f -= 1;
    *natural = f;
    *frames = s.out_frames > 0 ? s.out_frames : f;
    if (s.kind == MMFE_KIND_FRAME) *feat = s.win_len;
    else *feat = s.n_mel > 0 ? s.n_mel : s.n_fft / 2 + 1;
    return MMFE_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        target = dev;
    }
    ~DeviceGuard() { if (prev >= 0 && prev != target) (void)hipSetDevice(prev); }
    int target = -1;
};

void free_plan(Plan* p) {
    if (p == nullptr) return;
    (void)hipFree(p->d_window); (void)hipFree(p->d_tw_a); (void)hipFree(p->d_tw_b); (void)hipFree(p->d_tw64); (void)hipFree(p->d_window64);
    (void)hipFree(p->d_mel_start); (void)hipFree(p->d_mel_w); (void)hipFree(p->d_mel_niter); (void)hipFree(p->d_mel_woff);
    (void)hipFree(p->d_mel4_kstart); (void)hipFree(p->d_mel4_w); (void)hipFree(p->d_r400_dst);
    (void)hipFree(p->d_band_start); (void)hipFree(p->d_band_len); (void)hipFree(p->d_band_off); (void)hipFree(p->d_band_w);
    (void)hipFree(p->d_rowmax);
    delete p;
}

// ---- mmfe_run_noalloc: the per-row extremes of clamp_below_max live in a static device array -------------------
constexpr int64_t kNoAllocMaxRows = MMFE_NOALLOC_MAX_ROWS;
constexpr int kNoAllocMaxDevices = 64;
// Ordered-int maxima [0, kNoAllocMaxRows), minima [kNoAllocMaxRows, 2 * kNoAllocMaxRows): 512 KiB, resident on every GPU
// the library runs on until the process ends.  In host code this name is only a handle for hipGetSymbolAddress; the
// address kernels and memsets use is the per-device pointer obtained below.
static __device__ __align__(256) float g_noalloc_scratch[2 * kNoAllocMaxRows];

// Device address of g_noalloc_scratch on the CURRENT device (each GPU holds its own copy at its own address);
// looked up once per device.
int noalloc_scratch_ptr(int device, float** out) {
    static std::mutex mu;
    static float* ptrs[kNoAllocMaxDevices] = {nullptr};
    if (device < 0 || device >= kNoAllocMaxDevices) { set_error("device %d beyond the %d this entry point serves", device, kNoAllocMaxDevices); return MMFE_EINVAL; }
    std::lock_guard<std::mutex> lock(mu);
    if (ptrs[device] == nullptr) {
        void* p = nullptr;
        MMFE_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch)));
        if (p == nullptr) { set_error("hipGetSymbolAddress returned NULL for the static scratch"); return MMFE_EHIP; }
        ptrs[device] = static_cast<float*>(p);
    }
    *out = ptrs[device];
    return MMFE_OK;
}

int run_common(const Plan* plan, const RowSource& src, int64_t n_rows, int64_t n_samples, void* out_any, void* hip_stream,
               bool static_scratch = false) {
    float* out = static_cast<float*>(out_any);      // float32, or the 16-bit type the plan stores (each launcher knows its plan's out_dtype)
    if (plan == nullptr) { set_error("plan is NULL"); return MMFE_EINVAL; }
    if (n_rows < 0 || n_samples < 0) { set_error("negative size"); return MMFE_EINVAL; }
    if (n_rows == 0) return MMFE_OK;
    if (src.base == nullptr || out == nullptr) { set_error("NULL data pointer"); return MMFE_EINVAL; }
    const mmfe_spec& s = plan->spec;
    int64_t natural = 0, frames = 0, feat = 0;
    int rc = out_shape(s, n_samples, &natural, &frames, &feat);
    if (rc) return rc;
    if (frames == 0) return MMFE_OK;
    if (src.row_len != nullptr) {
        if (plan->impl != IMPL_FBANK512 && plan->impl != IMPL_DFT_GENERIC && plan->impl != IMPL_RFFT400) {
            set_error("per-row lengths are implemented for the FBANK and LOGSPEC kinds only");
            return MMFE_EINVAL;
        }
        if (s.clamp_below_max > 0.f) {
            set_error("per-row lengths are not implemented for clamp_below_max plans");
            return MMFE_EINVAL;
        }
    }
    DeviceGuard guard(plan->device);
    if (!guard.ok) { set_error("cannot make device %d current", plan->device); return MMFE_EHIP; }
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    ClampScratch cs{plan->d_rowmax, plan->rowmax_capacity};
    if (static_scratch) {
        cs.capacity = kNoAllocMaxRows;
        if ((rc = noalloc_scratch_ptr(plan->device, &cs.base))) return rc;
        if (s.clamp_below_max > 0.f && n_rows > kNoAllocMaxRows) {
            set_error("mmfe_run_noalloc: %lld rows with clamp_below_max, the static scratch holds %lld (use mmfe_run, or split the batch)",
                      (long long)n_rows, (long long)kNoAllocMaxRows);
            return MMFE_EINVAL;
        }
    }
    switch (plan->impl) {
        case IMPL_FBANK512: rc = fbank512_launch(*plan, src, n_rows, n_samples, natural, frames, out_any, stream); break;
        case IMPL_DFT_GENERIC: rc = dftgen_launch(*plan, src, n_rows, n_samples, natural, frames, out, cs, stream); break;
        case IMPL_RFFT400: rc = rfft400_launch(*plan, src, n_rows, n_samples, natural, frames, out, cs, stream); break;
        case IMPL_STFT32: rc = stft32_launch(*plan, src, n_rows, n_samples, frames, out, stream); break;
        case IMPL_FRAME: rc = frame_launch(*plan, src, n_rows, n_samples, frames, out_any, stream); break;
        case IMPL_RESAMPLE: rc = resample_launch(*plan, src, n_rows, n_samples, natural, frames, out, stream); break;
        default: set_error("plan has no implementation"); rc = MMFE_EINVAL;
    }
    if (rc) return rc;
    const float pad = (s.pad_value + s.out_shift) * s.out_scale;
    if (src.row_len != nullptr) rc = fill_pad_rows_ragged(out_any, s.out_dtype, n_rows, frames, feat, s.layout, pad, s, src.row_len, n_samples, stream);
    else if (natural < frames) rc = fill_pad_rows(out_any, s.out_dtype, n_rows, frames, natural, feat, s.layout, pad, stream);
    return rc;
}

}  // namespace
}  // namespace mmfe

using namespace mmfe;

extern "C" {

int mmfe_abi_version(void) { return MMFE_ABI_VERSION; }

const char* mmfe_last_error(void) { return g_last_error.c_str(); }

int mmfe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int mmfe_spec_out_shape(const mmfe_spec* spec, int64_t n_samples, int64_t* n_frames, int64_t* n_feat) {
    int rc = validate_spec(spec);
    if (rc) return rc;
    if (n_samples < 0 || n_frames == nullptr || n_feat == nullptr) { set_error("bad argument"); return MMFE_EINVAL; }
    int64_t natural = 0;
    return out_shape(*spec, n_samples, &natural, n_frames, n_feat);
}

int mmfe_plan_create(const mmfe_spec* spec, const void* window, const int32_t* band_start, const int32_t* band_len,
                     const float* band_weights, int device, mmfe_plan** out_plan) {
    if (out_plan == nullptr) { set_error("out_plan is NULL"); return MMFE_EINVAL; }
    *out_plan = nullptr;
    int rc = validate_spec(spec);
    if (rc) return rc;
    if (window == nullptr) { set_error("window is NULL"); return MMFE_EINVAL; }
    if (spec->kind != MMFE_KIND_FRAME && spec->n_mel > 0 && (band_start == nullptr || band_len == nullptr || band_weights == nullptr)) {
        set_error("n_mel > 0 needs band_start, band_len and band_weights");
        return MMFE_EINVAL;
    }
    const int ndev = mmfe_device_count();
    if (ndev <= 0) { set_error("no HIP device available: this library has no CPU path"); return MMFE_ENODEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return MMFE_ENODEVICE; }

    Plan* p = new (std::nothrow) Plan();
    if (p == nullptr) { set_error("out of host memory"); return MMFE_ENOMEM; }
    p->spec = *spec;
    p->device = device;
    p->n_bins = spec->n_fft / 2 + 1;
    if (spec->kind == MMFE_KIND_RESAMPLE && spec->window_f64) { set_error("RESAMPLE: the taps are float32"); free_plan(p); return MMFE_EINVAL; }
    p->window.resize(spec->win_len);
    p->window64.resize(spec->win_len);
    for (int i = 0; i < spec->win_len; ++i) {
        const double w = spec->window_f64 ? static_cast<const double*>(window)[i] : (double)static_cast<const float*>(window)[i];
        p->window64[i] = w;
        p->window[i] = (float)w;
    }
    if (spec->kind != MMFE_KIND_FRAME && spec->n_mel > 0) {
        p->band_start.assign(band_start, band_start + spec->n_mel);
        p->band_len.assign(band_len, band_len + spec->n_mel);
        int64_t total = 0;
        for (int m = 0; m < spec->n_mel; ++m) {
            if (band_len[m] < 0) { set_error("band_len[%d] < 0", m); free_plan(p); return MMFE_EINVAL; }
            total += band_len[m];
        }
        p->band_weights.assign(band_weights, band_weights + total);
    }
    DeviceGuard guard(device);
    if (!guard.ok) { set_error("cannot make device %d current", device); free_plan(p); return MMFE_EHIP; }

    switch (spec->kind) {
        case MMFE_KIND_FBANK:
            if (spec->frames_mode != MMFE_FRAMES_SNIP || spec->pad_mode != MMFE_PAD_NONE ||
                spec->n_mel <= 0 || spec->power != 2 || spec->spec_scale != 1.0f || spec->drop_last || spec->clamp_below_max > 0.f) {
                set_error("FBANK: needs snip-edges framing, no padding, mel bands, power spectrum");
                rc = MMFE_EINVAL;
                break;
            }
            if (spec->n_fft == 512 && spec->win_off == 0) { p->impl = IMPL_FBANK512; rc = fbank512_prepare(*p); }
            else if (rfft400_supports(*spec) && rfft400_prepare(*p) == MMFE_OK) p->impl = IMPL_RFFT400;
            else { p->impl = IMPL_DFT_GENERIC; rc = dftgen_prepare(*p); }
            break;
        case MMFE_KIND_LOGSPEC:
            if (spec->frames_mode != MMFE_FRAMES_CENTER || spec->pad_mode == MMFE_PAD_NONE) {
                set_error("LOGSPEC: needs centred framing with zero or reflect padding");
                rc = MMFE_EINVAL;
                break;
            }
            // n_fft = 400 with a mel bank has a radix kernel; everything else (and mel banks whose shape that
            // kernel has no variant for) goes to the any-length kernel
            p->impl = 0;
            if (rfft400_supports(*spec) && rfft400_prepare(*p) == MMFE_OK) p->impl = IMPL_RFFT400;
            if (p->impl == 0) {
                p->impl = IMPL_DFT_GENERIC;
                rc = dftgen_prepare(*p);
            }
            if (rc == MMFE_OK && spec->clamp_below_max > 0.f) {
                p->rowmax_capacity = 1 << 20;
                if (hipMalloc((void**)&p->d_rowmax, 2 * (size_t)p->rowmax_capacity * sizeof(float)) != hipSuccess) {      // maxima, then minima
                    set_error("hipMalloc of the row-max scratch failed");
                    rc = MMFE_ENOMEM;
                }
            }
            break;
        case MMFE_KIND_STFT_MAG:
            // nperseg 32 (the reference's IMU setting) has a kernel of its own; any other segment length runs on the
            // any-length DFT kernel with scipy's framing
            if (spec->n_fft == 32) {
                p->impl = IMPL_STFT32;
                rc = stft32_prepare(*p);
            } else if (spec->frames_mode != MMFE_FRAMES_SCIPY || spec->pad_mode != MMFE_PAD_ZERO || spec->win_off != 0 ||
                       spec->win_len != spec->n_fft || spec->n_mel != 0 || spec->out_frames != 0 || spec->clamp_below_max > 0.f) {
                set_error("STFT_MAG: needs scipy framing with zero padding, win_len == n_fft, no mel bank, natural frame count");
                rc = MMFE_EINVAL;
            } else {
                p->impl = IMPL_DFT_GENERIC;
                rc = dftgen_prepare(*p);
            }
            break;
        case MMFE_KIND_RESAMPLE:
            p->impl = IMPL_RESAMPLE;
            rc = resample_prepare(*p);
            break;
        case MMFE_KIND_FRAME:
            if (spec->frames_mode != MMFE_FRAMES_CEIL) { set_error("FRAME: needs MMFE_FRAMES_CEIL"); rc = MMFE_EINVAL; break; }
            p->impl = IMPL_FRAME;
            rc = spec->in_dtype == MMFE_IN_F64 ? upload(&p->d_window64, p->window64) : upload(&p->d_window, p->window);
            break;
        default:
            rc = MMFE_EINVAL;
    }
    if (rc == MMFE_OK && spec->out_dtype != MMFE_OUT_F32 &&
        !((spec->kind == MMFE_KIND_FBANK || spec->kind == MMFE_KIND_LOGSPEC) &&
          (p->impl == IMPL_FBANK512 || p->impl == IMPL_RFFT400 || (p->impl == IMPL_DFT_GENERIC && !(spec->clamp_below_max > 0.f))))) {
        set_error("16-bit output is implemented for the FBANK and LOGSPEC kinds (n_fft 512, n_fft 400, any-length DFT without clamp_below_max)");
        rc = MMFE_EINVAL;
    }
    if (rc) { free_plan(p); return rc; }
    if (hipDeviceSynchronize() != hipSuccess) { set_error("table upload failed"); free_plan(p); return MMFE_EHIP; }
    *out_plan = reinterpret_cast<mmfe_plan*>(p);
    return MMFE_OK;
}

int mmfe_plan_destroy(mmfe_plan* plan) {
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (p == nullptr) return MMFE_OK;
    DeviceGuard guard(p->device);
    free_plan(p);
    return MMFE_OK;
}

int mmfe_run(const mmfe_plan* plan, const void* in, int64_t in_stride, int64_t n_rows, int64_t n_samples, void* out,
             void* hip_stream) {
    if (in_stride < n_samples && n_rows > 1) { set_error("in_stride (%lld) smaller than n_samples (%lld)", (long long)in_stride, (long long)n_samples); return MMFE_EINVAL; }
    RowSource src{in, in_stride, nullptr, nullptr, nullptr, nullptr};
    return run_common(reinterpret_cast<const Plan*>(plan), src, n_rows, n_samples, out, hip_stream);
}

int mmfe_run_noalloc(const mmfe_plan* plan, const void* in, int64_t in_stride, int64_t n_rows, int64_t n_samples, void* out,
                     void* hip_stream) {
    if (in_stride < n_samples && n_rows > 1) { set_error("in_stride (%lld) smaller than n_samples (%lld)", (long long)in_stride, (long long)n_samples); return MMFE_EINVAL; }
    RowSource src{in, in_stride, nullptr, nullptr, nullptr, nullptr};
    return run_common(reinterpret_cast<const Plan*>(plan), src, n_rows, n_samples, out, hip_stream, true);
}

int mmfe_run_slices(const mmfe_plan* plan, const void* recs, const int64_t* rec_start, const int32_t* rec_index,
                    const int64_t* offsets, int64_t n_rows, int64_t n_samples, void* out, void* hip_stream) {
    if (rec_start == nullptr || rec_index == nullptr || offsets == nullptr) { set_error("NULL slice table"); return MMFE_EINVAL; }
    const Plan* p = reinterpret_cast<const Plan*>(plan);
    if (p != nullptr && p->impl == IMPL_STFT32) { set_error("slices are not supported by the 32-point STFT kernel"); return MMFE_EINVAL; }
    RowSource src{recs, 0, rec_start, rec_index, offsets, nullptr};
    return run_common(p, src, n_rows, n_samples, out, hip_stream);
}

int mmfe_run_ragged(const mmfe_plan* plan, const void* in, int64_t in_stride, const int32_t* row_len, int64_t n_rows,
                    int64_t n_samples, void* out, void* hip_stream) {
    if (row_len == nullptr) { set_error("row_len is NULL (use mmfe_run for rows of one length)"); return MMFE_EINVAL; }
    if (in_stride < n_samples && n_rows > 1) { set_error("in_stride (%lld) smaller than n_samples (%lld)", (long long)in_stride, (long long)n_samples); return MMFE_EINVAL; }
    RowSource src{in, in_stride, nullptr, nullptr, nullptr, row_len};
    return run_common(reinterpret_cast<const Plan*>(plan), src, n_rows, n_samples, out, hip_stream);
}

int mmfe_run_slices_ragged(const mmfe_plan* plan, const void* recs, const int64_t* rec_start, const int32_t* rec_index,
                           const int64_t* offsets, const int32_t* row_len, int64_t n_rows, int64_t n_samples, void* out,
                           void* hip_stream) {
    if (rec_start == nullptr || rec_index == nullptr || offsets == nullptr || row_len == nullptr) { set_error("NULL slice table"); return MMFE_EINVAL; }
    RowSource src{recs, 0, rec_start, rec_index, offsets, row_len};
    return run_common(reinterpret_cast<const Plan*>(plan), src, n_rows, n_samples, out, hip_stream);
}

int mmfe_feature_sums(const float* x, int64_t n, double* sums, int device, void* hip_stream) {
    if (x == nullptr || sums == nullptr || n < 0) { set_error("bad argument"); return MMFE_EINVAL; }
    const int ndev = mmfe_device_count();
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return MMFE_ENODEVICE; }
    DeviceGuard guard(device);
    if (!guard.ok) { set_error("cannot make device %d current", device); return MMFE_EHIP; }
    return feature_sums_launch(x, n, sums, reinterpret_cast<hipStream_t>(hip_stream));
}

const char* mmfe_plan_kernel(const mmfe_plan* plan) {
    const Plan* p = reinterpret_cast<const Plan*>(plan);
    if (p == nullptr) return "none";
    switch (p->impl) {
        case IMPL_FBANK512: return "fbank512";
        case IMPL_RFFT400: return "rfft400";
        case IMPL_DFT_GENERIC: return "dftgen";
        case IMPL_STFT32: return "stft32";
        case IMPL_FRAME: return "frame";
        case IMPL_RESAMPLE: return "resample";
        default: return "none";
    }
}

}  // extern "C"

// gut_api.cpp — host side of libgut_hip.so: handle, grow-only scratch, launch orchestration and the C ABI
// declared in include/gut_hip.h.  Replaces, for the 3DGUT path, the reference's SplatRaster
// (src/splatRaster.cpp:153-364) and GUTRenderer (src/gutRenderer.cu:99-497).
//
// Compiled with -ffp-contract=off: the pose matrices built here feed the bit-exact projection kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>

#include "gut_internal.h"

namespace {

thread_local std::string g_last_error;

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // grow-only (gutRenderer.cu:136-231 resizes the same way); returns hipSuccess or the allocation error
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = bytes + bytes / 4 + 256;
        if (p) {
            hipError_t e = hipFree(p);  // synchronises the device: safe w.r.t. in-flight users
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

// Every entry point runs on the handle's device and leaves the calling thread's current device as it found it
// (a host application with several GPUs per process must not see its device change under it).
struct DeviceGuard {
    int prev = -1;
    hipError_t set(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { prev = -1; return e; }
        if (prev == device) { prev = -1; return hipSuccess; }
        return hipSetDevice(device);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    bool armed = false;
};

// ---- pose math (fp32, GLM-equivalent definitions of the tiny-cuda-nn calls in sensors/sensors.h:44-73) ----
void quat_to_rot(float w, float x, float y, float z, float r[3][3] /* row-major */) {
    const float qxx = x * x, qyy = y * y, qzz = z * z;
    const float qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    // column-major definition m[col][row], written out row-major here
    r[0][0] = 1.0f - 2.0f * (qyy + qzz); r[1][0] = 2.0f * (qxy + qwz); r[2][0] = 2.0f * (qxz - qwy);
    r[0][1] = 2.0f * (qxy - qwz); r[1][1] = 1.0f - 2.0f * (qxx + qzz); r[2][1] = 2.0f * (qyz + qwx);
    r[0][2] = 2.0f * (qxz + qwy); r[1][2] = 2.0f * (qyz - qwx); r[2][2] = 1.0f - 2.0f * (qxx + qyy);
}

void interpolate_pose(const float* a, const float* b, float t, float* out) {
    float qa[4] = {a[6], a[3], a[4], a[5]};
    float qb[4] = {b[6], b[3], b[4], b[5]};
    float cos_t = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
    if (cos_t < 0.0f) {
        for (int i = 0; i < 4; ++i) qb[i] = -qb[i];
        cos_t = -cos_t;
    }
    float q[4];
    if (cos_t > 1.0f - 1.1920929e-07f) {
        for (int i = 0; i < 4; ++i) q[i] = qa[i] * (1.0f - t) + qb[i] * t;
    } else {
        const float ang = acosf(cos_t);
        const float s0 = sinf((1.0f - t) * ang), s1 = sinf(t * ang), sd = sinf(ang);
        for (int i = 0; i < 4; ++i) q[i] = (s0 * qa[i] + s1 * qb[i]) / sd;
    }
    for (int i = 0; i < 3; ++i) out[i] = a[i] * (1.0f - t) + b[i] * t;
    out[3] = q[1]; out[4] = q[2]; out[5] = q[3]; out[6] = q[0];
}

uint32_t bit_width_u32(uint32_t n) {  // == higherMsb of gutRenderer.cu:79-94 for n >= 1
    uint32_t w = 0;
    while (n) { ++w; n >>= 1; }
    return w;
}

}  // namespace

struct gut_context {
    int device = 0;
    GutConfig cfg{};
    gut::RenderConsts consts{};
    std::mutex mu;

    // per-N scratch
    DevBuf tiles_count, tiles_offset, proj_pos, conic_opacity, extent, depth, feat, grad16, scan_temp;
    float* stat_accum = nullptr;    // gut_set_position_gradient_statistics: consumed by the next gut_optimize_after_bwd
    int32_t* stat_denom = nullptr;
    DevBuf cam_pos;   // [3] floats: the sensor position of the cached forward (written by K1)
    DevBuf wave_sums, block_prefix, scan_total;   // two-level scan of the tile counts (K1 wave sums -> k_scan_wave_sums -> K3)
    bool timing_main_stream = true, timing_side_stream = true;   // GUT_OPT_KERNEL_TIMING_SET
    bool dbg_offset_valid = false;                // tiles_offset (debug view only) rebuilt from tiles_count for this frame
    DevBuf sph_widened, sph_grad_wide;   // particle_radiance_sph_degree < 3: the [N,48] rows the kernels read / write
    DevBuf packed12;   // gut_trace_fields: the [N,12] rows packed from the caller's four tensors (kept for its backward)
    bool packed_valid = false;
    bool packed_raw = false;   // packed12 holds rows ACTIVATED by the library from the model's raw tensors (gut_trace_raw_model_fields)
    // per-M scratch
    DevBuf keys_unsorted, keys_sorted, ids_unsorted, ids_sorted, sort_temp;
    // lazy per-tile depth order (unsorted variant): keys_sorted / ids_sorted are grouped by tile only, the forward compositor
    // writes the ids it consumed, in final order, to ids_ordered; the fully sorted lists exist only on debug request
    DevBuf ids_ordered, dbg_keys_sorted, dbg_ids_sorted;
    bool lazy_enabled = true;      // gut_set_option(GUT_OPT_LAZY_TILE_ORDER)
    bool sorted_reference_bwd = true;   // gut_set_option(GUT_OPT_SORTED_REFERENCE_BACKWARD): the reference's form is the default
    bool lazy_order = false;       // this forward used the lazy order
    bool dbg_sorted_valid = false;
    // per-T
    DevBuf ranges, trav_fwd, trav_bwd, tile_order;  // per-tile traversal depths (statistics)
    DevBuf counters;
    uint32_t* host_count = nullptr;  // pinned
    uint32_t* host_count_dev = nullptr;   // the same words as the device addresses them (written by k_scan_wave_sums)
    hipEvent_t count_event = nullptr;  // the count read-back has landed (work queued behind it keeps the GPU busy meanwhile)
    // rows without tiles are optimised early, on a low-priority side stream under the compositing kernels
    // (gut_optimize_rows_without_gradient): ev_projected = projection AND binning of the cached forward have finished (tiles_count is
    // final), ev_early_done = the side stream's pass has finished
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_projected = nullptr, ev_early_done = nullptr;
    // the pass runs in two launches: the first block range under the forward compositor, the second under the backward
    // compositor (queued by gut_trace_bwd_ex), so that the short, latency-bound loss kernels in between are left alone
    struct EarlyArgs {
        float *raw12, *raw_m, *raw_v, *sh48, *sh_m, *sh_v, *act12;
        float lr12[12], lr48[48];
        float beta1, beta2, eps;
        uint32_t step, block_begin, block_end;
        uint32_t extra_end;   // blocks < extra_end: waves with tiles the forward did not walk also go to the second launch
        GutLazyMoments lazy;  // lazy moment decay of this step (d_wave_step == NULL: off)
    } early_args{};
    bool marks_valid = false; // wave_walked holds the walked-wave marks of the cached forward for EVERY wave
    DevBuf wave_walked;       // one byte per 64-row wave (k_mark_walked_waves), valid when early_args.extra_end > 0
    int early_extra_percent = 100;
    bool stats_early = false;   // the last optimiser step used the side stream (for gut_get_stats)
    uint32_t stats_split = 0, stats_extra_end = 0;
    bool early_part2_pending = false;
    hipEvent_t ev_bwd_start = nullptr;
    bool early_ran = false;           // the cached forward's rows without tiles already had their optimiser step
    bool early_wait_pending = false;  // the main stream has not been ordered behind ev_early_done yet
    bool grad16_zero = false;          // every row of grad16 is zero: true after a clear, false once K7 has added to it, true
                                       // again when a per-Gaussian consumer (K8 / the optimiser kernel) has walked the rows
                                       // that have tiles (they zero what they read)
    // binning capacity: the forward is queued against m_capacity list slots before this frame's count is known
    uint32_t m_capacity = 0, m_peak = 0, sort_n = 0;
    uint64_t overflows = 0;
    DevBuf zero_word, tile_ordered, walk_sums;
    bool fwd_longest_first = false;   // forward compositor launched longest-lists-first (decided from the last frames' walked share)
    int fwd_order_mode = -1;          // GUT_OPT_FORWARD_TILE_ORDER: -1 auto, 0 image order, 1 longest first
    bool dbg_ordered_valid = false;

    // cached forward context (gutRenderer.cu:252-254, 413)
    bool have_forward = false;
    hipStream_t fwd_stream = nullptr;
    uint32_t n = 0, m = 0;
    int width = 0, height = 0, tiles = 0, sh_degree = 0, end_bit = 0;
    gut::ViewParams view{};
    bool have_backward = false;

    // timers
    std::deque<EventPair> fwd_timers, bwd_timers;
    float last_fwd_ms = -1.f, last_bwd_ms = -1.f;
    // per-kernel event boundaries: a ring of sets so that bench.py can average over its whole timed region
    static constexpr int kRing = 64;
    struct KevSet {
        hipEvent_t e[16] = {};
        bool fwd = false, bwd = false, opt = false, early = false, early2 = false;
    };
    KevSet ring[kRing];
    int ring_cur = 0;    // set used by the most recent trace()
    int ring_count = 0;  // sets recorded since the last gut_kernel_times_mean
    hipEvent_t* kev = ring[0].e;
    bool kev_fwd_valid = false, kev_bwd_valid = false;
};

namespace {

int build_view(const GutCamera* cam, int W, int H, int shutter_iterations, gut::ViewParams* v) {
    if (cam->model != GUT_CAMERA_OPENCV_PINHOLE && cam->model != GUT_CAMERA_OPENCV_FISHEYE)
        return fail("unsupported camera model %d (only OpenCV pinhole / fisheye exist in the reference)", cam->model);
    if (cam->shutter < GUT_SHUTTER_ROLLING_TOP_TO_BOTTOM || cam->shutter > GUT_SHUTTER_GLOBAL)
        return fail("unknown shutter type %d", cam->shutter);
    memset(v, 0, sizeof(*v));
    v->shutter = cam->shutter;
    for (int i = 0; i < 7; ++i) {
        v->pose_start[i] = cam->pose_start[i];
        v->pose_end[i] = cam->pose_end[i];
    }
    {
        const float* e = cam->pose_end;
        quat_to_rot(e[6], e[3], e[4], e[5], v->w2s_end.r);
        for (int i = 0; i < 3; ++i) v->w2s_end.t[i] = e[i];
    }
    const float* s = cam->pose_start;
    quat_to_rot(s[6], s[3], s[4], s[5], v->w2s_start.r);
    for (int i = 0; i < 3; ++i) v->w2s_start.t[i] = s[i];
    float mid[7];
    interpolate_pose(cam->pose_start, cam->pose_end, 0.5f, mid);
    quat_to_rot(mid[6], mid[3], mid[4], mid[5], v->w2s_mid.r);
    for (int i = 0; i < 3; ++i) v->w2s_mid.t[i] = mid[i];
    // sensor->world = (R^T, -R^T t).  The reference round-trips R^T through a quaternion
    // (sensors.h:44-53); skipping that removes rounding noise and changes no integer buffer.
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) v->s2w.r[r][c] = v->w2s_mid.r[c][r];
    for (int r = 0; r < 3; ++r)
        v->s2w.t[r] = -1.0f * (v->s2w.r[r][0] * mid[0] + v->s2w.r[r][1] * mid[1] + v->s2w.r[r][2] * mid[2]);
    v->model = cam->model;
    v->width = W;
    v->height = H;
    v->grid_x = (W + gut::kTile - 1) / gut::kTile;
    v->grid_y = (H + gut::kTile - 1) / gut::kTile;
    for (int i = 0; i < 2; ++i) {
        v->principal_point[i] = cam->principal_point[i];
        v->focal_length[i] = cam->focal_length[i];
        v->tangential[i] = cam->tangential_coeffs[i];
    }
    for (int i = 0; i < 6; ++i) v->radial[i] = cam->radial_coeffs[i];
    for (int i = 0; i < 4; ++i) v->thin_prism[i] = cam->thin_prism_coeffs[i];
    v->max_angle = cam->max_angle;
    v->shutter_iterations = shutter_iterations;
    return 0;
}

void build_consts(const GutConfig& cfg, gut::RenderConsts* c) {
    c->alpha_threshold = cfg.particle_kernel_min_alpha;
    c->max_alpha = cfg.particle_kernel_max_alpha;
    c->min_response = cfg.particle_kernel_min_response;
    c->min_transmittance = cfg.min_transmittance;
    c->min_sensor_z = 0.2f;   // threedgut.cuh:49
    c->cov_dilation = 0.3f;   // threedgut.cuh:50
    const float D = 3.0f;
    const float lambda = cfg.ut_alpha * cfg.ut_alpha * (D + cfg.ut_kappa) - D;
    // GAUSSIAN_UT_DELTA is made by the reference's build script in double and rounded once (setup_3dgut.py:41-45, threedgut.cuh:70)
    c->ut_delta = (float)sqrt((double)cfg.ut_alpha * (double)cfg.ut_alpha * (3.0 + (double)cfg.ut_kappa));
    c->ut_w0_mean = lambda / (D + lambda);
    c->ut_wi = 1.0f / (2.0f * (D + lambda));
    c->ut_w0_cov = lambda / (D + lambda) + (1.0f - cfg.ut_alpha * cfg.ut_alpha + cfg.ut_beta);
    c->ut_margin = cfg.ut_in_image_margin_factor;
    c->rect_bounding = cfg.rect_bounding;
    c->tight_opacity_bounding = cfg.tight_opacity_bounding;
    c->tile_culling = cfg.tile_based_culling;
    c->global_z_order = cfg.global_z_order;
    // the early-out radius of the compositors; the exact response / alpha tests follow it, so a small margin is free
    c->max_d2 = cfg.particle_kernel_degree == 2 ? -2.0f * logf(cfg.particle_kernel_min_response)
                                                : gut::kernel_cutoff_d2(cfg.particle_kernel_degree, cfg.particle_kernel_min_response) * 1.0001f + 1e-6f;
}

// None of the library's events publishes device memory to the host: they time kernels, order the library's own streams on one
// device, or tell the host that the scan kernel has written the intersection count into COHERENT pinned memory (system-scope stores
// of that kernel).  Without these flags every hipEventRecord makes the next dispatch wait for a system-scope cache writeback and
// invalidation (hip_runtime_api.h: hipEventDisableSystemFence) — about 30 us per event on the 6 M-Gaussian frame, measured where the
// count used to be copied out.
static const bool g_event_system_fence = getenv("GUT_EVENT_SYSTEM_FENCE") != nullptr;   // A/B experiments only
static const unsigned kTimingEvent = g_event_system_fence ? 0u : hipEventDisableSystemFence;
static const unsigned kOrderingEvent = hipEventDisableTiming | (g_event_system_fence ? 0u : hipEventDisableSystemFence);

EventPair* arm_timer(std::deque<EventPair>& q, hipStream_t s) {
    if (q.size() >= 256) {  // keep only the most recent (splatRaster.cpp:212-215)
        EventPair old = q.front();
        q.pop_front();
        if (old.a) (void)hipEventDestroy(old.a);
        if (old.b) (void)hipEventDestroy(old.b);
    }
    EventPair p;
    if (hipEventCreateWithFlags(&p.a, kTimingEvent) != hipSuccess || hipEventCreateWithFlags(&p.b, kTimingEvent) != hipSuccess) return nullptr;
    (void)hipEventRecord(p.a, s);
    q.push_back(p);
    return &q.back();
}

float drain_timers(std::deque<EventPair>& q) {
    if (q.empty()) return -1.f;
    float sum = 0.f;
    int cnt = 0;
    for (auto& p : q) {
        if (p.armed && hipEventSynchronize(p.b) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                sum += ms;
                cnt++;
            }
        }
        if (p.a) (void)hipEventDestroy(p.a);
        if (p.b) (void)hipEventDestroy(p.b);
    }
    q.clear();
    return cnt ? sum / (float)cnt : -1.f;
}

}  // namespace

static int launch_early_part2(gut_context* h, hipStream_t s);
gut::LazyMoments gut_make_lazy(const GutLazyMoments* lazy, uint32_t step);   // gut_train.hip

extern "C" {

const char* gut_last_error(void) { return g_last_error.c_str(); }
int gut_abi_version(void) { return GUT_ABI_VERSION; }

void gut_default_config(GutConfig* cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = GUT_ABI_VERSION;
    cfg->enable_kernel_timings = 0;
    cfg->particle_radiance_sph_degree = 3;
    cfg->particle_kernel_degree = 2;
    cfg->k_buffer_size = 0;
    cfg->global_z_order = 1;
    cfg->n_rolling_shutter_iterations = 5;
    cfg->ut_require_all_sigma_points = 0;
    cfg->rect_bounding = 1;
    cfg->tight_opacity_bounding = 1;
    cfg->tile_based_culling = 1;
    cfg->enable_hitcounts = 1;
    cfg->particle_kernel_min_response = 0.0113f;
    cfg->particle_kernel_min_alpha = 1.0f / 255.0f;
    cfg->particle_kernel_max_alpha = 0.99f;
    cfg->min_transmittance = 0.0001f;
    cfg->ut_alpha = 1.0f;
    cfg->ut_beta = 2.0f;
    cfg->ut_kappa = 0.0f;
    cfg->ut_in_image_margin_factor = 0.1f;
}

int gut_create(const GutConfig* cfg, int device_index, gut_handle* out) {
    if (!cfg || !out) return fail("gut_create: null argument");
    if (cfg->abi_version != GUT_ABI_VERSION) return fail("gut_create: ABI version %d, library is %d", cfg->abi_version, GUT_ABI_VERSION);
    // the reference compiles one kernel variant per render config (setup_3dgut.py:47-70); this library ships
    // the default 3dgut variant and says so instead of silently rendering something else
    switch (cfg->particle_kernel_degree) {   // particleResponse<> has exactly these cases (gaussianParticles.cuh:256-306; anything else
    case 0: case 1: case 2: case 3: case 4: case 5: case 8: break;   // falls to its quadratic default, which would be a silent surprise)
    default: return fail("particle_kernel_degree=%d: the reference's generalised Gaussian kernels are 0, 1, 2, 3, 4, 5 and 8", cfg->particle_kernel_degree);
    }
    if (cfg->k_buffer_size < 0 || cfg->k_buffer_size > 16)
        return fail("k_buffer_size=%d: supported range is 0 (unsorted) .. 16", cfg->k_buffer_size);
    // PARTICLE_RADIANCE_NUM_COEFFS = (degree + 1)^2 (setup_3dgut.py:48): the width of the radiance rows.  The kernels are built for 16
    // coefficients; narrower rows are widened on the way in and the gradient narrowed on the way out (trace_fwd_impl / trace_bwd_impl)
    if (cfg->particle_radiance_sph_degree < 0 || cfg->particle_radiance_sph_degree > 3)
        return fail("particle_radiance_sph_degree=%d outside 0..3", cfg->particle_radiance_sph_degree);
    if (cfg->ut_require_all_sigma_points != 0) return fail("ut_require_all_sigma_points must be false (static_assert in threedgut.cuh:73)");
    if (cfg->n_rolling_shutter_iterations < 0 || cfg->n_rolling_shutter_iterations > 64)
        return fail("n_rolling_shutter_iterations=%d outside 0..64", cfg->n_rolling_shutter_iterations);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_index < 0 || device_index >= ndev) return fail("gut_create: device %d out of range (%d devices)", device_index, ndev);
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(device_index));
    gut_context* h = new (std::nothrow) gut_context();
    if (!h) return fail("gut_create: out of host memory");
    h->device = device_index;
    h->cfg = *cfg;
    build_consts(*cfg, &h->consts);
    // coherent (fine-grained) pinned words: k_scan_wave_sums stores the count here with system scope, no cache writeback needed
    hipError_t e = hipHostMalloc((void**)&h->host_count, 64, hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) memset(h->host_count, 0, 64);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->host_count_dev, h->host_count, 0);
    if (e == hipSuccess) e = h->counters.ensure(sizeof(gut::Counters));
    if (e != hipSuccess) {
        delete h;
        return fail("gut_create: allocation failed: %s", hipGetErrorString(e));
    }
    if (cfg->enable_kernel_timings)
        for (auto& set : h->ring)
            for (auto& ev : set.e)
                if (hipEventCreateWithFlags(&ev, kTimingEvent) != hipSuccess) ev = nullptr;
    *out = h;
    return 0;
}

void gut_destroy(gut_handle h) {
    if (!h) return;
    DeviceGuard dev_guard;
    (void)dev_guard.set(h->device);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = {&h->tiles_count, &h->tiles_offset, &h->proj_pos, &h->conic_opacity, &h->extent, &h->depth, &h->feat,
                      &h->grad16, &h->scan_temp, &h->keys_unsorted, &h->keys_sorted, &h->ids_unsorted, &h->ids_sorted,
                      &h->sort_temp, &h->ranges, &h->trav_fwd, &h->trav_bwd, &h->tile_order, &h->counters, &h->ids_ordered,
                      &h->dbg_keys_sorted, &h->dbg_ids_sorted, &h->zero_word, &h->tile_ordered, &h->wave_walked, &h->packed12, &h->walk_sums,
                      &h->wave_sums, &h->block_prefix, &h->scan_total, &h->sph_widened, &h->sph_grad_wide, &h->cam_pos};
    for (DevBuf* b : bufs) b->release();
    if (h->host_count) (void)hipHostFree(h->host_count);
    if (h->count_event) (void)hipEventDestroy(h->count_event);
    if (h->ev_projected) (void)hipEventDestroy(h->ev_projected);
    if (h->ev_early_done) (void)hipEventDestroy(h->ev_early_done);
    if (h->ev_bwd_start) (void)hipEventDestroy(h->ev_bwd_start);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    (void)drain_timers(h->fwd_timers);
    (void)drain_timers(h->bwd_timers);
    for (auto& set : h->ring)
        for (auto& ev : set.e)
            if (ev) (void)hipEventDestroy(ev);
    delete h;
}

static int trace_fwd_impl(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                          const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                          const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                          float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
                          float* d_particle_visibility, const float* d_features_albedo);

// host_count[2], [3]: sum of the traversal depths / of the list lengths of the last frame that had a backward (copied with the count)
static void update_fwd_order(gut_handle h) {
    if (!h->walk_sums.p) return;
    const uint32_t walked = h->host_count[2], listed = h->host_count[3];
    if (listed) h->fwd_longest_first = (double)walked > 0.25 * (double)listed;
}

int gut_trace(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
              const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
              const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
              float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
              float* d_particle_visibility) {
    return trace_fwd_impl(h, stream_, frame_number, num_active_features, num_particles, d_particle_density, d_particle_radiance, width,
                          height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density, d_ray_hit_distance, d_ray_hit_count,
                          d_particle_visibility, nullptr);
}

// d_features_albedo != NULL: d_particle_radiance is features_specular [N,45] (gut_trace_model_fields)
static int trace_fwd_impl(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                          const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                          const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                          float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
                          float* d_particle_visibility, const float* d_features_albedo) {
    (void)frame_number;
    if (!h) return fail("gut_trace: null handle");
    if (!camera || !d_ray_origin || !d_ray_direction || !d_ray_radiance_density || !d_ray_hit_distance || !d_ray_hit_count)
        return fail("gut_trace: null pointer argument");
    if (width <= 0 || height <= 0) return fail("gut_trace: bad resolution %dx%d", width, height);
    if (num_active_features < 0 || num_active_features > 3) return fail("gut_trace: SH degree %d outside 0..3", num_active_features);
    if (num_particles && (!d_particle_density || !d_particle_radiance || !d_particle_visibility))
        return fail("gut_trace: null particle buffers with %u particles", num_particles);
    if (h->cfg.particle_radiance_sph_degree != 3) {
        if (num_active_features > h->cfg.particle_radiance_sph_degree)
            return fail("gut_trace: %d active SH degrees, the handle was created for particle_radiance_sph_degree=%d", num_active_features,
                        h->cfg.particle_radiance_sph_degree);
        if (d_features_albedo)
            return fail("gut_trace_model_fields: features_specular [N,45] belongs to particle_radiance_sph_degree=3; this handle has %d "
                        "(use gut_trace / gut_trace_fields with the concatenated rows)", h->cfg.particle_radiance_sph_degree);
    }
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    gut::ViewParams v;
    if (build_view(camera, width, height, h->cfg.n_rolling_shutter_iterations, &v)) return 1;
    const uint32_t n = num_particles;
    const int tiles = v.grid_x * v.grid_y;
    if ((uint64_t)tiles >= 0xFFFFFFFFull) return fail("gut_trace: too many tiles");
    if (h->early_ran)
        return fail("gut_trace: gut_optimize_rows_without_gradient was called for the previous forward but gut_optimize_after_bwd "
                    "was not: that optimiser step is half applied");
    if (h->early_wait_pending) {  // (only reachable after an error path; the optimiser call normally orders the streams)
        HIP_TRY(hipStreamWaitEvent(s, h->ev_early_done, 0));
        h->early_wait_pending = false;
    }
    h->have_forward = false;
    h->have_backward = false;
    h->marks_valid = false;
    h->packed_valid = false;   // (gut_trace_fields sets it again once this call has succeeded)

    HIP_TRY(h->tiles_count.ensure(sizeof(uint32_t) * (size_t)n));
    {
        const size_t blocks = ((size_t)n + gut::kBlock - 1) / gut::kBlock;
        HIP_TRY(h->wave_sums.ensure(sizeof(uint32_t) * (gut::kBlock / 64) * (blocks + 1)));
        HIP_TRY(h->block_prefix.ensure(sizeof(uint32_t) * (blocks + 1)));
        HIP_TRY(h->scan_total.ensure(64));
    }
    HIP_TRY(h->proj_pos.ensure(sizeof(float) * 2 * (size_t)n));
    HIP_TRY(h->conic_opacity.ensure(sizeof(float) * 4 * (size_t)n));
    HIP_TRY(h->extent.ensure(sizeof(float) * 2 * (size_t)n));
    HIP_TRY(h->depth.ensure(sizeof(float) * (size_t)n));
    HIP_TRY(h->feat.ensure(sizeof(float) * 3 * (size_t)n));
    HIP_TRY(h->ranges.ensure(sizeof(uint32_t) * 2 * (size_t)tiles));
    HIP_TRY(h->trav_fwd.ensure(sizeof(uint32_t) * (size_t)tiles));
    HIP_TRY(h->trav_bwd.ensure(sizeof(uint32_t) * (size_t)tiles));
    HIP_TRY(h->tile_order.ensure(sizeof(uint32_t) * (size_t)tiles));
    HIP_TRY(h->tile_ordered.ensure(sizeof(uint32_t) * (size_t)tiles));
    if (!h->zero_word.p) {
        HIP_TRY(h->zero_word.ensure(64));
        HIP_TRY(hipMemsetAsync(h->zero_word.p, 0, 64, s));
    }

    const bool timing = h->cfg.enable_kernel_timings != 0 && h->timing_main_stream;
    EventPair* total = timing ? arm_timer(h->fwd_timers, s) : nullptr;
    h->kev_fwd_valid = false;
    if (h->cfg.enable_kernel_timings != 0 && (h->timing_main_stream || h->timing_side_stream)) {   // (a fresh set of events per frame)
        h->ring_cur = (h->ring_cur + 1) % gut_context::kRing;
        h->kev = h->ring[h->ring_cur].e;
        h->ring[h->ring_cur].fwd = false;
        h->ring[h->ring_cur].bwd = false;
        h->ring[h->ring_cur].opt = false;
        h->ring[h->ring_cur].early = false;
        h->ring[h->ring_cur].early2 = false;
    }
    auto mark = [&](int i) {
        if (timing && h->kev[i]) (void)hipEventRecord(h->kev[i], s);
    };

    // per-tile ranges / backward traversal depths and the per-wave "walked" bytes start every frame at zero: K1 clears them on its way
    // (three fill launches on the step's critical chain until round 4); without Gaussians there is no K1
    gut::FrameClears clears;
    clears.ranges = h->ranges.as<uint2>();
    clears.trav_bwd = h->trav_bwd.as<uint32_t>();
    clears.tiles = (uint32_t)tiles;
    if (n) {
        HIP_TRY(h->wave_walked.ensure(((size_t)n + gut::kBlock - 1) / gut::kBlock * (gut::kBlock / 64) + 64));
        clears.wave_walked = h->wave_walked.as<uint8_t>();
        HIP_TRY(h->cam_pos.ensure(64));
        clears.cam_pos = h->cam_pos.as<float>();
    } else {
        HIP_TRY(hipMemsetAsync(h->trav_bwd.p, 0, sizeof(uint32_t) * (size_t)tiles, s));
        HIP_TRY(hipMemsetAsync(h->ranges.p, 0, sizeof(uint32_t) * 2 * (size_t)tiles, s));
    }
    mark(0);
    if (h->cfg.particle_radiance_sph_degree != 3 && n) {
        // rows of 3 (degree + 1)^2 floats (gaussianParticles.cuh:208-216 reads PARTICLE_RADIANCE_NUM_COEFFS coefficients per particle):
        // zero-extended to the 16 coefficients the projection kernel is laid out for; the coefficients above the active degree are
        // never evaluated, so the render is that of the narrower variant
        const uint32_t w = 3u * (uint32_t)((h->cfg.particle_radiance_sph_degree + 1) * (h->cfg.particle_radiance_sph_degree + 1));
        HIP_TRY(h->sph_widened.ensure(sizeof(float) * 48 * (size_t)n));
        gut::launch_resize_sph_rows(s, n, w, 48u, d_particle_radiance, h->sph_widened.as<float>());
        d_particle_radiance = h->sph_widened.as<float>();
    }
    gut::launch_project(s, v, h->consts, n, num_active_features, d_particle_density, d_particle_radiance,
                        h->tiles_count.as<uint32_t>(), h->proj_pos.as<float>(), h->conic_opacity.as<float>(),
                        h->extent.as<float>(), h->depth.as<float>(), h->feat.as<float>(), d_particle_visibility,
                        h->wave_sums.as<uint32_t>(), d_features_albedo, clears);
    mark(1);
    static const bool early_after_project = getenv("GUT_EARLY_AFTER_PROJECT") != nullptr;  // tuning experiments only
    if (early_after_project) {
        if (!h->ev_projected) HIP_TRY(hipEventCreateWithFlags(&h->ev_projected, kOrderingEvent));
        HIP_TRY(hipEventRecord(h->ev_projected, s));
    }
    const int end_bit = 32 + (int)bit_width_u32((uint32_t)tiles);
    h->dbg_sorted_valid = false;
    h->dbg_ordered_valid = false;
    // device-side intersection count: written by the scan of the block sums (a zero word when there are no particles)
    const uint32_t* d_count = n ? h->scan_total.as<uint32_t>() : h->zero_word.as<uint32_t>();
    h->dbg_offset_valid = false;
    uint32_t m = 0;
    bool count_pending = false;
    if (n) {
        // K2 (gutRenderer.cu:300-311: a device-wide cub scan of the [N] counts): two levels here — K1 left one sum per wave, one
        // workgroup scans the per-block sums (and writes the count), K3 finishes inside each wave.  One 5 us launch instead of three
        // launches over 48 MB (0.047 ms at 6 M Gaussians).
        gut::launch_scan_wave_sums(s, n, h->wave_sums.as<uint32_t>(), h->block_prefix.as<uint32_t>(), h->scan_total.as<uint32_t>(),
                                   h->host_count_dev, h->walk_sums.as<uint32_t>());
        // intersection count read-back (gutRenderer.cu:313-321).  The reference blocks on it before it can size the
        // binning buffers; here the copy is queued and the host only waits for it AFTER the rest of the forward has been
        // queued against a capacity taken from the previous frames (m_capacity), so the GPU never idles on the host.
        // The scan kernel writes the count — and with it the walked share of the last frame that had a backward (k_tile_order's
        // walk_sums; see launch_render below) — straight into the pinned host words; the event below orders the host's read.
        if (!h->count_event) HIP_TRY(hipEventCreateWithFlags(&h->count_event, kOrderingEvent));
        HIP_TRY(hipEventRecord(h->count_event, s));
        count_pending = true;
    }
    mark(2);
    // One binning + compositing pass over `sort_n` list slots (>= the real count, the tail is padding).  Returns through
    // the lambda so that the rare overflow can run it a second time.
    auto bin_and_render = [&](uint32_t sort_n, bool lazy) -> int {
        h->lazy_order = lazy;
        if (sort_n) {
            HIP_TRY(h->keys_unsorted.ensure(sizeof(uint64_t) * (size_t)sort_n));
            HIP_TRY(h->keys_sorted.ensure(sizeof(uint64_t) * (size_t)sort_n));
            HIP_TRY(h->ids_unsorted.ensure(sizeof(uint32_t) * (size_t)sort_n));
            HIP_TRY(h->ids_sorted.ensure(sizeof(uint32_t) * (size_t)sort_n));
            gut::launch_expand(s, v, h->consts, n, h->tiles_count.as<uint32_t>(), h->wave_sums.as<uint32_t>(),
                               h->block_prefix.as<uint32_t>(), d_count, h->proj_pos.as<float>(),
                               h->conic_opacity.as<float>(), h->extent.as<float>(), h->depth.as<float>(),
                               h->keys_unsorted.as<uint64_t>(), h->ids_unsorted.as<uint32_t>(), sort_n);
            mark(3);
            if (lazy) {
                HIP_TRY(h->sort_temp.ensure(gut::sort_tiles_temp_bytes(sort_n, end_bit)));
                HIP_TRY(h->ids_ordered.ensure(sizeof(uint32_t) * (size_t)sort_n));
                HIP_TRY(gut::run_sort_tiles(s, h->sort_temp.p, h->sort_temp.cap, h->keys_unsorted.as<uint64_t>(),
                                            h->keys_sorted.as<uint64_t>(), h->ids_unsorted.as<uint32_t>(), h->ids_sorted.as<uint32_t>(),
                                            sort_n, end_bit));
            } else {
                HIP_TRY(h->sort_temp.ensure(gut::sort_temp_bytes(sort_n, end_bit)));
                HIP_TRY(gut::run_sort(s, h->sort_temp.p, h->sort_temp.cap, h->keys_unsorted.as<uint64_t>(), h->keys_sorted.as<uint64_t>(),
                                      h->ids_unsorted.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), sort_n, end_bit));
            }
            mark(4);
            gut::launch_tile_ranges(s, sort_n, h->keys_sorted.as<uint64_t>(), h->ranges.as<uint32_t>());
        } else {
            mark(3);
            mark(4);
        }
        mark(5);
        // the side-stream optimiser pass may start here: tiles_count is final and the HBM-bound part of the forward
        // (projection, scan, expansion, sort) is over — what follows on this stream is VALU-bound compositing
        if (!early_after_project) {
            if (!h->ev_projected) HIP_TRY(hipEventCreateWithFlags(&h->ev_projected, kOrderingEvent));
            HIP_TRY(hipEventRecord(h->ev_projected, s));
        }
        // with zero intersections the reference returns its freshly initialised outputs (gutRenderer.cu:323-325);
        // running the compositor over empty ranges writes exactly those values
        if (h->cfg.k_buffer_size > 0) {
            HIP_TRY(hipMemsetAsync(h->trav_fwd.p, 0, sizeof(uint32_t) * (size_t)tiles, s));
            gut::launch_render_sorted(s, v, h->consts, h->cfg.k_buffer_size, d_particle_density, h->feat.as<float>(), d_ray_origin,
                                      d_ray_direction, h->ranges.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), d_count,
                                      d_ray_radiance_density, d_ray_hit_distance, d_ray_hit_count, h->cfg.particle_kernel_degree);
        } else {
            // Longest lists first (as the backward launches its deepest tiles first), when the frames before this one walked a good part
            // of their lists — then a tile's list length says how long it will run, and the few long ones must not form the tail of
            // the grid: surface-like stand-in (E/M = 0.46) K6 1.36 -> 1.15 ms; where whole-tile termination leaves most of every list
            // untouched (headline stand-in, E/M = 0.12) the length says little and the natural order keeps neighbouring tiles, which
            // share Gaussians, together in time (0.52 -> 0.53 ms with the order, plus its 8 us launch).  The walked share comes from
            // the backward's own ordering kernel (walk_sums), read back with the intersection count: no extra synchronisation.
            const bool fwd_order = h->fwd_order_mode >= 0 ? h->fwd_order_mode != 0 : h->fwd_longest_first;   // GUT_OPT_FORWARD_TILE_ORDER
            if (fwd_order && sort_n)
                gut::launch_tile_order(s, (uint32_t)tiles, nullptr, h->tile_order.as<uint32_t>(), h->ranges.as<uint32_t>(), true);
            gut::launch_render(s, v, h->consts, d_particle_density, h->feat.as<float>(), d_ray_origin, d_ray_direction,
                               h->ranges.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), d_count, d_ray_radiance_density,
                               d_ray_hit_distance, d_ray_hit_count, h->trav_fwd.as<uint32_t>(),
                               lazy ? h->keys_sorted.as<uint64_t>() : nullptr, lazy ? h->ids_ordered.as<uint32_t>() : nullptr,
                               lazy ? h->tile_ordered.as<uint32_t>() : nullptr,
                               (fwd_order && sort_n) ? h->tile_order.as<uint32_t>() : nullptr, h->cfg.particle_kernel_degree);
        }
        // render.enable_hitcounts = false: the reference compiles the counter out (rayPayload.cuh:44-46,69-73,126-128) and its
        // output tensor keeps the zeros it was created with (splatRaster.cpp:198).  Not a configuration worth a kernel variant.
        if (!h->cfg.enable_hitcounts) HIP_TRY(hipMemsetAsync(d_ray_hit_count, 0, sizeof(float) * (size_t)width * (size_t)height, s));
        return 0;
    };
    // the k-buffer variant walks whole lists: it keeps the full sort
    // ... and frames whose tile lists are so long on average that re-scanning them per 512 ordered entries cannot pay
    auto want_lazy = [&](uint64_t count) {
        return h->lazy_enabled && h->cfg.k_buffer_size == 0 && count != 0 && count < (uint64_t)tiles * 16384ull;
    };
    if (n && h->m_capacity == 0) {
        // first frame on this handle (or after a reset): nothing to size from, wait for the count like the reference
        HIP_TRY(hipEventSynchronize(h->count_event));
        count_pending = false;
        m = *h->host_count;
        update_fwd_order(h);
        if (bin_and_render(m, want_lazy(m))) return 1;
    } else {
        const uint32_t sort_n = n ? h->m_capacity : 0u;
        if (bin_and_render(sort_n, want_lazy(sort_n))) return 1;
        if (count_pending) {
            HIP_TRY(hipEventSynchronize(h->count_event));  // everything is queued: the GPU keeps working while the host waits here
            count_pending = false;
            m = *h->host_count;
            update_fwd_order(h);
        update_fwd_order(h);
        }
        if (m > sort_n) {
            // the frame has more intersections than the capacity assumed: entries beyond it were dropped by the expansion.
            // Redo binning and compositing with the real count (same stream: the second pass simply overwrites the first).
            HIP_TRY(hipMemsetAsync(h->ranges.p, 0, sizeof(uint32_t) * 2 * (size_t)tiles, s));
            h->overflows++;
            if (bin_and_render(m, want_lazy(m))) return 1;
            h->sort_n = m;
        } else {
            h->sort_n = sort_n;
        }
    }
    if (n && h->m_capacity == 0) h->sort_n = m;
    // capacity for the next frame: a little above the largest recent count (slowly forgetting old peaks), multiple of 4096
    {
        const double decayed = (double)h->m_peak * 0.999;
        h->m_peak = (uint32_t)((double)m > decayed ? (double)m : decayed);
        const uint64_t want = (uint64_t)((double)h->m_peak * 1.03) + 4096ull;
        h->m_capacity = m || h->m_peak ? (uint32_t)(((want + 4095ull) / 4096ull) * 4096ull) : 0u;
    }
    mark(6);
    HIP_TRY(hipGetLastError());
    if (total) {
        (void)hipEventRecord(total->b, s);
        total->armed = true;
    }
    h->kev_fwd_valid = timing;
    if (timing) h->ring[h->ring_cur].fwd = true;
    if (h->cfg.enable_kernel_timings != 0 && (h->timing_main_stream || h->timing_side_stream) && h->ring_count < gut_context::kRing) h->ring_count++;
    h->have_forward = true;
    h->fwd_stream = s;
    h->n = n;
    h->m = m;
    h->width = width;
    h->height = height;
    h->tiles = tiles;
    h->sh_degree = num_active_features;
    h->end_bit = end_bit;
    h->view = v;
    return 0;
}

int gut_trace_bwd_ex(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                     const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                     const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                     const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                     const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_particle_density_grad,
                     float* d_particle_radiance_grad, uint32_t flags);

int gut_trace_bwd(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                  const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                  const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                  const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                  const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_particle_density_grad,
                  float* d_particle_radiance_grad) {
    return gut_trace_bwd_ex(h, stream_, frame_number, num_active_features, num_particles, d_particle_density, d_particle_radiance,
                            width, height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density,
                            d_ray_radiance_density_grad, d_ray_hit_distance, d_ray_hit_distance_grad, d_particle_density_grad,
                            d_particle_radiance_grad, 0u);
}

static int trace_bwd_impl(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                          const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                          const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                          const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                          const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_particle_density_grad,
                          float* d_particle_radiance_grad, uint32_t flags, const gut::GradFields& fields);

int gut_trace_bwd_ex(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                     const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                     const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                     const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                     const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_particle_density_grad,
                     float* d_particle_radiance_grad, uint32_t flags) {
    return trace_bwd_impl(h, stream_, frame_number, num_active_features, num_particles, d_particle_density, d_particle_radiance, width,
                          height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density, d_ray_radiance_density_grad,
                          d_ray_hit_distance, d_ray_hit_distance_grad, d_particle_density_grad, d_particle_radiance_grad, flags,
                          gut::GradFields());
}

// SplatRaster::trace with the four activated tensors the reference's Tracer hands to _Autograd (tracer.py:317-327) instead of
// their concatenation: the [N,12] rows are packed into handle scratch by one coalesced kernel (the torch.cat of tracer.py:176-178
// runs at a quarter of that rate) and stay there for gut_trace_bwd_fields.
int gut_trace_fields(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                     const float* d_positions, const float* d_density, const float* d_rotation, const float* d_scale,
                     const float* d_particle_radiance, int32_t width, int32_t height, const float* d_ray_origin,
                     const float* d_ray_direction, const GutCamera* camera, float* d_ray_radiance_density,
                     float* d_ray_hit_distance, float* d_ray_hit_count, float* d_particle_visibility) {
    if (!h) return fail("gut_trace_fields: null handle");
    if (num_particles && (!d_positions || !d_density || !d_rotation || !d_scale))
        return fail("gut_trace_fields: null particle buffers with %u particles", num_particles);
    if (((uintptr_t)d_rotation & 15u) != 0) return fail("gut_trace_fields: the rotation tensor must be 16-byte aligned");
    {
        std::lock_guard<std::mutex> lock(h->mu);
        DeviceGuard dev_guard;
        HIP_TRY(dev_guard.set(h->device));
        h->packed_valid = false;
        h->packed_raw = false;
        HIP_TRY(h->packed12.ensure(sizeof(float) * 12 * (size_t)num_particles + 64));
        gut::launch_pack_fields(static_cast<hipStream_t>(stream_), num_particles, d_positions, d_density, d_rotation, d_scale,
                                h->packed12.as<float>());
        HIP_TRY(hipGetLastError());
    }
    const int rc = gut_trace(h, stream_, frame_number, num_active_features, num_particles, h->packed12.as<float>(), d_particle_radiance,
                             width, height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density, d_ray_hit_distance,
                             d_ray_hit_count, d_particle_visibility);
    if (rc == 0) h->packed_valid = true;
    return rc;
}

// SplatRaster::traceBwd for the forward above; the density gradient is written as the four tensors of _Autograd.backward's
// return value (tracer.py:268-286: no [N,12] tensor, no split, no four .contiguous() copies)
int gut_trace_bwd_fields(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                         const float* d_particle_radiance, int32_t width, int32_t height, const float* d_ray_origin,
                         const float* d_ray_direction, const GutCamera* camera, const float* d_ray_radiance_density,
                         const float* d_ray_radiance_density_grad, const float* d_ray_hit_distance,
                         const float* d_ray_hit_distance_grad, float* d_positions_grad, float* d_density_grad, float* d_rotation_grad,
                         float* d_scale_grad, float* d_particle_radiance_grad) {
    if (!h) return fail("gut_trace_bwd_fields: null handle");
    if (!h->packed_valid) return fail("gut_trace_bwd_fields: no gut_trace_fields forward on this handle");
    if (num_particles && (!d_positions_grad || !d_density_grad || !d_rotation_grad || !d_scale_grad || !d_particle_radiance_grad))
        return fail("gut_trace_bwd_fields: null gradient buffers");
    if (((uintptr_t)d_rotation_grad & 15u) != 0) return fail("gut_trace_bwd_fields: the rotation gradient must be 16-byte aligned");
    gut::GradFields f;
    f.pos = d_positions_grad; f.dns = d_density_grad; f.rot = d_rotation_grad; f.scl = d_scale_grad;
    return trace_bwd_impl(h, stream_, frame_number, num_active_features, num_particles, h->packed12.as<float>(), d_particle_radiance, width,
                          height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density, d_ray_radiance_density_grad,
                          d_ray_hit_distance, d_ray_hit_distance_grad, d_positions_grad /* non-null marker */, d_particle_radiance_grad, 0u, f);
}

// gut_trace_fields / gut_trace_bwd_fields with the SH coefficients as the model's two tensors too (features_albedo [N,3],
// features_specular [N,45]; model.py:68-75) instead of get_features()'s torch.cat: K1 reads the wave's [64,45] block directly
// (only the rows of Gaussians that survived culling), K8 writes the two gradient tensors (LDS-transposed, coalesced).
int gut_trace_model_fields(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                           const float* d_positions, const float* d_density, const float* d_rotation, const float* d_scale,
                           const float* d_features_albedo, const float* d_features_specular, int32_t width, int32_t height,
                           const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                           float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
                           float* d_particle_visibility) {
    if (!h) return fail("gut_trace_model_fields: null handle");
    if (num_particles && (!d_positions || !d_density || !d_rotation || !d_scale || !d_features_albedo || !d_features_specular))
        return fail("gut_trace_model_fields: null particle buffers with %u particles", num_particles);
    if ((((uintptr_t)d_rotation | (uintptr_t)d_features_specular) & 15u) != 0)
        return fail("gut_trace_model_fields: the rotation and features_specular tensors must be 16-byte aligned");
    {
        std::lock_guard<std::mutex> lock(h->mu);
        DeviceGuard dev_guard;
        HIP_TRY(dev_guard.set(h->device));
        h->packed_valid = false;
        h->packed_raw = false;
        HIP_TRY(h->packed12.ensure(sizeof(float) * 12 * (size_t)num_particles + 64));
        gut::launch_pack_fields(static_cast<hipStream_t>(stream_), num_particles, d_positions, d_density, d_rotation, d_scale,
                                h->packed12.as<float>());
        HIP_TRY(hipGetLastError());
    }
    const int rc = trace_fwd_impl(h, stream_, frame_number, num_active_features, num_particles, h->packed12.as<float>(),
                                  d_features_specular, width, height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density,
                                  d_ray_hit_distance, d_ray_hit_count, d_particle_visibility,
                                  num_particles ? d_features_albedo : nullptr);
    if (rc == 0) h->packed_valid = true;
    return rc;
}

// gut_trace_model_fields on the model's PRE-ACTIVATION tensors (model.py:74-93: density logit, un-normalised quaternion, log-scale):
// sigmoid / normalize / exp and the packing run as one kernel (k_pack_activate_fields), and the backward that follows
// (gut_trace_bwd_model_fields on this handle) returns the gradients w.r.t. those raw tensors (K8 <raw, split>): the model's three
// activation kernels, their three backward kernels and autograd's bookkeeping for them drop out of the reference trainer's step.
int gut_trace_raw_model_fields(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                               const float* d_positions, const float* d_density_logit, const float* d_rotation_raw, const float* d_log_scale,
                               const float* d_features_albedo, const float* d_features_specular, int32_t width, int32_t height,
                               const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                               float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
                               float* d_particle_visibility) {
    if (!h) return fail("gut_trace_raw_model_fields: null handle");
    if (num_particles && (!d_positions || !d_density_logit || !d_rotation_raw || !d_log_scale || !d_features_albedo || !d_features_specular))
        return fail("gut_trace_raw_model_fields: null particle buffers with %u particles", num_particles);
    if ((((uintptr_t)d_rotation_raw | (uintptr_t)d_features_specular) & 15u) != 0)
        return fail("gut_trace_raw_model_fields: the rotation and features_specular tensors must be 16-byte aligned");
    {
        std::lock_guard<std::mutex> lock(h->mu);
        DeviceGuard dev_guard;
        HIP_TRY(dev_guard.set(h->device));
        h->packed_valid = false;
        h->packed_raw = false;
        HIP_TRY(h->packed12.ensure(sizeof(float) * 12 * (size_t)num_particles + 64));
        gut::launch_pack_activate_fields(static_cast<hipStream_t>(stream_), num_particles, d_positions, d_density_logit, d_rotation_raw,
                                         d_log_scale, h->packed12.as<float>());
        HIP_TRY(hipGetLastError());
    }
    const int rc = trace_fwd_impl(h, stream_, frame_number, num_active_features, num_particles, h->packed12.as<float>(),
                                  d_features_specular, width, height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density,
                                  d_ray_hit_distance, d_ray_hit_count, d_particle_visibility,
                                  num_particles ? d_features_albedo : nullptr);
    if (rc == 0) { h->packed_valid = true; h->packed_raw = true; }
    return rc;
}

int gut_trace_bwd_model_fields(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                               int32_t width, int32_t height, const float* d_ray_origin, const float* d_ray_direction,
                               const GutCamera* camera, const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                               const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_positions_grad,
                               float* d_density_grad, float* d_rotation_grad, float* d_scale_grad, float* d_features_albedo_grad,
                               float* d_features_specular_grad) {
    if (!h) return fail("gut_trace_bwd_model_fields: null handle");
    if (!h->packed_valid) return fail("gut_trace_bwd_model_fields: no gut_trace_fields / gut_trace_model_fields forward on this handle");
    if (num_particles && (!d_positions_grad || !d_density_grad || !d_rotation_grad || !d_scale_grad || !d_features_albedo_grad ||
                          !d_features_specular_grad))
        return fail("gut_trace_bwd_model_fields: null gradient buffers");
    if ((((uintptr_t)d_rotation_grad | (uintptr_t)d_features_specular_grad) & 15u) != 0)
        return fail("gut_trace_bwd_model_fields: the rotation and features_specular gradients must be 16-byte aligned");
    gut::GradFields f;
    f.pos = d_positions_grad; f.dns = d_density_grad; f.rot = d_rotation_grad; f.scl = d_scale_grad;
    f.alb = d_features_albedo_grad; f.spec = d_features_specular_grad;
    // (after gut_trace_raw_model_fields the rows carry |quat| in the pad column and the gradients are chained to the raw tensors)
    return trace_bwd_impl(h, stream_, frame_number, num_active_features, num_particles, h->packed12.as<float>(), nullptr, width,
                          height, d_ray_origin, d_ray_direction, camera, d_ray_radiance_density, d_ray_radiance_density_grad,
                          d_ray_hit_distance, d_ray_hit_distance_grad, d_positions_grad /* non-null markers */, d_features_specular_grad,
                          h->packed_raw ? GUT_BWD_RAW_PARAMETER_GRADS : 0u, f);
}

static int trace_bwd_impl(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                          const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                          const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                          const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                          const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_particle_density_grad,
                          float* d_particle_radiance_grad, uint32_t flags, const gut::GradFields& fields) {
    (void)frame_number;
    (void)d_particle_radiance;
    if (!h) return fail("gut_trace_bwd: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    // same contract as the reference: backward needs the forward's cached context on the same stream
    // (gutRenderer.cu:413-417)
    if (!h->have_forward || h->fwd_stream != s)
        return fail("gut_trace_bwd: no forward context on this stream (call gut_trace first, same stream)");
    if (num_particles != h->n || width != h->width || height != h->height || num_active_features != h->sh_degree)
        return fail("gut_trace_bwd: arguments differ from the cached forward (N %u vs %u, %dx%d vs %dx%d)", num_particles, h->n,
                    width, height, h->width, h->height);
    // d_ray_hit_distance_grad may be NULL (= all zeros: no loss on pred_dist, the default of the reference's trainer);
    // the hit-distance gradient terms are then compiled out of the backward kernel
    if (!camera || !d_ray_origin || !d_ray_direction || !d_ray_radiance_density || !d_ray_radiance_density_grad)
        return fail("gut_trace_bwd: null pointer argument");
    if (num_particles && (!d_particle_density || (!(flags & GUT_BWD_SKIP_EPILOGUE) && (!d_particle_density_grad || !d_particle_radiance_grad))))
        return fail("gut_trace_bwd: null particle buffers");
    if (h->early_ran && !(flags & GUT_BWD_SKIP_EPILOGUE))
        return fail("gut_trace_bwd: gut_optimize_rows_without_gradient was called for this forward: the backward must be "
                    "gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE) followed by gut_optimize_after_bwd");
    float* narrow_radiance_grad = nullptr;   // particle_radiance_sph_degree < 3: the caller's [N, 3 (degree + 1)^2] gradient
    if (h->cfg.particle_radiance_sph_degree != 3 && !(flags & GUT_BWD_SKIP_EPILOGUE)) {
        if ((flags & GUT_BWD_COMPACT_RADIANCE_GRADS) || fields.alb)
            return fail("gut_trace_bwd: particle_radiance_sph_degree=%d supports the packed radiance gradient only", h->cfg.particle_radiance_sph_degree);
        narrow_radiance_grad = d_particle_radiance_grad;
    }
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    gut::ViewParams v;
    if (build_view(camera, width, height, h->cfg.n_rolling_shutter_iterations, &v)) return 1;
    if (memcmp(&v, &h->view, sizeof(v)) != 0) return fail("gut_trace_bwd: camera differs from the cached forward");
    const uint32_t n = h->n;
    if (n == 0) return 0;
    {
        const void* before = h->grad16.p;
        HIP_TRY(h->grad16.ensure(sizeof(float) * 16 * (size_t)n));
        if (h->grad16.p != before) h->grad16_zero = false;  // fresh allocation
    }
    if (narrow_radiance_grad) {
        HIP_TRY(h->sph_grad_wide.ensure(sizeof(float) * 48 * (size_t)n));
        d_particle_radiance_grad = h->sph_grad_wide.as<float>();
    }

    const bool timing = h->cfg.enable_kernel_timings != 0 && h->timing_main_stream;
    EventPair* total = timing ? arm_timer(h->bwd_timers, s) : nullptr;
    h->kev_bwd_valid = false;
    auto mark = [&](int i) {
        if (timing && h->kev[i]) (void)hipEventRecord(h->kev[i], s);
    };
    mark(8);
    // the gradient rows are zero already unless this is the first backward or the previous one was never consumed (every
    // per-Gaussian consumer below zeroes the rows it reads, so there is no 64 N-byte clear per step)
    if (!h->grad16_zero) HIP_TRY(hipMemsetAsync(h->grad16.p, 0, h->grad16.cap, s));
    h->grad16_zero = false;
    if (launch_early_part2(h, s)) return 1;
    mark(9);
    if (h->m && h->cfg.k_buffer_size > 0) {
        if (!d_ray_hit_distance) return fail("gut_trace_bwd: the sorted variant needs d_ray_hit_distance");
        HIP_TRY(hipMemsetAsync(h->trav_bwd.p, 0, sizeof(uint32_t) * (size_t)h->tiles, s));
        gut::launch_render_sorted_bwd(s, v, h->consts, h->cfg.k_buffer_size, d_particle_density, h->feat.as<float>(), d_ray_origin,
                                      d_ray_direction, h->ranges.as<uint32_t>(), h->ids_sorted.as<uint32_t>(),
                                      d_ray_radiance_density, d_ray_hit_distance, d_ray_radiance_density_grad,
                                      d_ray_hit_distance_grad, h->grad16.as<float>(), h->sorted_reference_bwd, h->cfg.particle_kernel_degree);
    } else if (h->m) {
        HIP_TRY(h->walk_sums.ensure(2 * sizeof(uint32_t)));
        gut::launch_tile_order(s, (uint32_t)h->tiles, h->trav_fwd.as<uint32_t>(), h->tile_order.as<uint32_t>(), h->ranges.as<uint32_t>(), false,
                               h->walk_sums.as<uint32_t>());
        gut::launch_render_bwd(s, v, h->consts, d_particle_density, h->feat.as<float>(), d_ray_origin, d_ray_direction,
                               h->ranges.as<uint32_t>(), (h->lazy_order ? h->ids_ordered : h->ids_sorted).as<uint32_t>(),
                               d_ray_radiance_density,
                               d_ray_radiance_density_grad, d_ray_hit_distance_grad, h->grad16.as<float>(),
                               h->trav_bwd.as<uint32_t>(), h->tile_order.as<uint32_t>(), h->trav_fwd.as<uint32_t>(), h->cfg.particle_kernel_degree);
    }
    mark(10);
    if (flags & GUT_BWD_SKIP_EPILOGUE) {
        // the caller folds the per-Gaussian epilogue into its optimiser step (gut_optimize_after_bwd)
    } else {
        if (flags & GUT_BWD_COMPACT_RADIANCE_GRADS)
            gut::launch_project_bwd_compact(s, n, d_particle_density, h->tiles_count.as<uint32_t>(), h->feat.as<float>(),
                                            h->grad16.as<float>(), d_particle_density_grad, d_particle_radiance_grad);
        else
            gut::launch_project_bwd(s, v, n, h->sh_degree, d_particle_density, h->tiles_count.as<uint32_t>(), h->feat.as<float>(),
                                    h->grad16.as<float>(), d_particle_density_grad, d_particle_radiance_grad,
                                    (flags & GUT_BWD_RAW_PARAMETER_GRADS) != 0, fields);
        if (narrow_radiance_grad) {
            const uint32_t w = 3u * (uint32_t)((h->cfg.particle_radiance_sph_degree + 1) * (h->cfg.particle_radiance_sph_degree + 1));
            gut::launch_resize_sph_rows(s, n, 48u, w, h->sph_grad_wide.as<float>(), narrow_radiance_grad);
        }
        h->grad16_zero = true;  // the epilogue zeroed every row K7 could have touched (the rows with tiles)
    }
    mark(11);
    HIP_TRY(hipGetLastError());
    if (total) {
        (void)hipEventRecord(total->b, s);
        total->armed = true;
    }
    h->kev_bwd_valid = timing;
    if (timing) h->ring[h->ring_cur].bwd = true;
    h->have_backward = true;
    return 0;
}

int gut_optimize_after_bwd(gut_handle h, void* stream_, int32_t num_active_features, const float* d_camera_position,
                           float* d_raw12, float* d_raw_m, float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v,
                           const float* lr12, const float* lr48, float beta1, float beta2, float eps, uint32_t step,
                           const float* d_visibility, float* d_act12_out, const GutLazyMoments* lazy) {
    if (!h) return fail("gut_optimize_after_bwd: null handle");
    if (h->cfg.particle_radiance_sph_degree != 3)
        return fail("gut_optimize_after_bwd: the fused optimiser is built for particle_radiance_sph_degree=3 (59 parameters per Gaussian), this handle has %d", h->cfg.particle_radiance_sph_degree);
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (!h->have_backward || h->fwd_stream != s)
        return fail("gut_optimize_after_bwd: no backward context on this stream (call gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE) first)");
    if (num_active_features != h->sh_degree) return fail("gut_optimize_after_bwd: sh degree differs from the cached forward");
    if (h->n == 0) return 0;
    if (!d_raw12 || !d_raw_m || !d_raw_v || !d_sh48 || !d_sh_m || !d_sh_v || !lr12 || !lr48)
        return fail("gut_optimize_after_bwd: null pointer argument");
    if (!d_camera_position) d_camera_position = h->cam_pos.as<float>();   // the cached forward's own sensor position (K1 left it there)
    if (h->early_ran && d_visibility)
        return fail("gut_optimize_after_bwd: a visibility mask cannot follow gut_optimize_rows_without_gradient");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    if (lazy && d_visibility) return fail("gut_optimize_after_bwd: lazy moment decay cannot be combined with a visibility mask");
    if (h->early_ran && ((lazy && lazy->d_wave_step) != (h->early_args.lazy.d_wave_step != nullptr)))
        return fail("gut_optimize_after_bwd: lazy moment decay must be the same as in gut_optimize_rows_without_gradient");
    if (launch_early_part2(h, s)) return 1;  // (normally queued by gut_trace_bwd_ex already)
    const gut::LazyMoments lz = gut_make_lazy(lazy, step);
    // lazy moment decay: the waves that cannot receive a gradient are the same whichever kernel walks them — no tile, or no
    // Gaussian of the wave among the entries the forward walked (unsorted variant).  The marks exist already when the side
    // stream's second launch built them; otherwise (one-pass step) they are built here.
    if (lz.wave_step && h->cfg.k_buffer_size == 0 && h->m && !h->marks_valid) {
        // (the bytes are zero: K1 cleared them at the start of this frame and nothing has marked them since — marks_valid)
        gut::launch_mark_walked_waves(s, h->n, (uint32_t)h->tiles, h->ranges.as<uint32_t>(), h->trav_fwd.as<uint32_t>(),
                                      (h->lazy_order ? h->ids_ordered : h->ids_sorted).as<uint32_t>(), h->wave_walked.as<uint8_t>());
        h->marks_valid = true;
    }
    const bool timing = h->cfg.enable_kernel_timings != 0 && h->timing_main_stream && h->kev[12] && h->kev[13];
    if (timing) (void)hipEventRecord(h->kev[12], s);
    gut::launch_sh_adam_from_scratch(s, h->n, h->sh_degree, d_camera_position, h->grad16.as<float>(), h->tiles_count.as<uint32_t>(),
                                     h->feat.as<float>(), d_raw12, d_raw_m, d_raw_v, d_sh48, d_sh_m, d_sh_v, lr12, lr48, beta1, beta2,
                                     eps, step, d_visibility, d_act12_out, h->early_ran,
                                     (h->early_ran && h->early_args.extra_end) ? h->wave_walked.as<uint8_t>() : nullptr,
                                     h->early_args.block_begin, h->early_ran ? h->early_args.extra_end : 0u, lz,
                                     (lz.wave_step && h->marks_valid) ? h->wave_walked.as<uint8_t>() : nullptr,
                                     h->stat_accum, h->stat_denom);
    h->stat_accum = nullptr;   // one optimiser call only (the caller may reallocate its buffers any time)
    h->stat_denom = nullptr;
    HIP_TRY(hipGetLastError());
    if (timing) {
        (void)hipEventRecord(h->kev[13], s);
        h->ring[h->ring_cur].opt = true;
    }
    if (h->early_wait_pending) {  // whatever follows on this stream (the next forward) also sees the side stream's rows
        HIP_TRY(hipStreamWaitEvent(s, h->ev_early_done, 0));
        h->early_wait_pending = false;
    }
    h->stats_early = h->early_ran;
    h->stats_split = h->early_args.block_begin;
    h->stats_extra_end = h->early_ran ? h->early_args.extra_end : 0u;
    h->early_ran = false;
    h->have_backward = false;  // the gradient rows are consumed ...
    h->grad16_zero = true;     // ... and left zero by the kernel
    return 0;
}

int gut_set_position_gradient_statistics(gut_handle h, float* d_norm_accum, int32_t* d_norm_denom) {
    if (!h) return fail("gut_set_position_gradient_statistics: null handle");
    if ((d_norm_accum == nullptr) != (d_norm_denom == nullptr))
        return fail("gut_set_position_gradient_statistics: both buffers or neither");
    std::lock_guard<std::mutex> lock(h->mu);
    h->stat_accum = d_norm_accum;
    h->stat_denom = d_norm_denom;
    return 0;
}

int gut_optimize_finish_without_gradient(gut_handle h, void* stream_) {
    if (!h) return fail("gut_optimize_finish_without_gradient: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->early_ran) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (h->fwd_stream != s) return fail("gut_optimize_finish_without_gradient: not the stream of the forward");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    const gut_context::EarlyArgs ea = h->early_args;
    h->stat_accum = nullptr;   // (a step finished without its gradient has no statistics either)
    h->stat_denom = nullptr;
    // discard whatever the backward compositor may have accumulated: every remaining wave sees an exactly-zero gradient
    HIP_TRY(h->grad16.ensure(sizeof(float) * 16 * (size_t)h->n));
    HIP_TRY(hipMemsetAsync(h->grad16.p, 0, h->grad16.cap, s));
    if (launch_early_part2(h, s)) return 1;
    gut::launch_sh_adam_from_scratch(s, h->n, h->sh_degree, h->zero_word.as<float>() /* never read: no row has a colour gradient */,
                                     h->grad16.as<float>(), h->tiles_count.as<uint32_t>(), h->feat.as<float>(), ea.raw12, ea.raw_m,
                                     ea.raw_v, ea.sh48, ea.sh_m, ea.sh_v, ea.lr12, ea.lr48, ea.beta1, ea.beta2, ea.eps, ea.step, nullptr,
                                     ea.act12, true, ea.extra_end ? h->wave_walked.as<uint8_t>() : nullptr, ea.block_begin, ea.extra_end,
                                     gut_make_lazy(&ea.lazy, ea.step), h->marks_valid ? h->wave_walked.as<uint8_t>() : nullptr);
    HIP_TRY(hipGetLastError());
    if (h->early_wait_pending) {
        HIP_TRY(hipStreamWaitEvent(s, h->ev_early_done, 0));
        h->early_wait_pending = false;
    }
    h->stats_early = true;
    h->stats_split = ea.block_begin;
    h->stats_extra_end = ea.extra_end;
    h->early_ran = false;
    h->have_backward = false;
    h->grad16_zero = true;
    return 0;
}

int gut_mark_walked_waves(gut_handle h, void* stream_, uint8_t* d_wave_flags) {
    if (!h) return fail("gut_mark_walked_waves: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (!h->have_forward || h->fwd_stream != s)
        return fail("gut_mark_walked_waves: no forward context on this stream (call gut_trace first, same stream)");
    if (h->n == 0) return 0;
    if (!d_wave_flags) return fail("gut_mark_walked_waves: null pointer argument");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    HIP_TRY(hipMemsetAsync(d_wave_flags, 0, ((size_t)h->n + 63) / 64, s));
    if (h->m == 0) return 0;
    if (h->cfg.k_buffer_size > 0)   // sorted variant: its backward is not bounded by the forward's depth -> every wave with a tile
        gut::launch_mark_waves_with_tiles(s, h->n, h->tiles_count.as<uint32_t>(), d_wave_flags);
    else
        gut::launch_mark_walked_waves(s, h->n, (uint32_t)h->tiles, h->ranges.as<uint32_t>(), h->trav_fwd.as<uint32_t>(),
                                      (h->lazy_order ? h->ids_ordered : h->ids_sorted).as<uint32_t>(), d_wave_flags);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gut_compact_gradient_rows(gut_handle h, void* stream_, const float* d_particle_density, float* d_records, uint32_t capacity,
                              uint32_t* d_count) {
    if (!h) return fail("gut_compact_gradient_rows: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (!h->have_backward || h->fwd_stream != s)
        return fail("gut_compact_gradient_rows: no backward context on this stream (call gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE) first)");
    if (h->early_ran) return fail("gut_compact_gradient_rows: cannot follow gut_optimize_rows_without_gradient");
    if (!d_count || (h->n && (!d_particle_density || !d_records))) return fail("gut_compact_gradient_rows: null pointer argument");
    if (capacity < h->n) return fail("gut_compact_gradient_rows: the record buffer must hold one record per particle");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
    gut::launch_compact_gradient_rows(s, h->n, d_particle_density, h->tiles_count.as<uint32_t>(), h->feat.as<float>(),
                                      h->grad16.as<float>(), d_records, capacity, d_count);
    HIP_TRY(hipGetLastError());
    h->have_backward = false;  // the gradient rows are consumed ...
    h->grad16_zero = true;     // ... and left zero by the kernel
    return 0;
}

int gut_optimize_rows_without_gradient(gut_handle h, void* stream_, float* d_raw12, float* d_raw_m, float* d_raw_v, float* d_sh48,
                                       float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48, float beta1, float beta2,
                                       float eps, uint32_t step, float* d_act12_out, const GutLazyMoments* lazy) {
    if (!h) return fail("gut_optimize_rows_without_gradient: null handle");
    if (h->cfg.particle_radiance_sph_degree != 3)
        return fail("gut_optimize_rows_without_gradient: the fused optimiser is built for particle_radiance_sph_degree=3 (59 parameters per Gaussian), this handle has %d", h->cfg.particle_radiance_sph_degree);
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (!h->have_forward || h->fwd_stream != s)
        return fail("gut_optimize_rows_without_gradient: no forward context on this stream (call gut_trace first, same stream)");
    if (h->early_ran) return fail("gut_optimize_rows_without_gradient: already called for this forward");
    if (h->have_backward) return fail("gut_optimize_rows_without_gradient: call it between gut_trace and gut_trace_bwd_ex");
    if (h->n == 0) return 0;
    if (!d_raw12 || !d_raw_m || !d_raw_v || !d_sh48 || !d_sh_m || !d_sh_v || !lr12 || !lr48)
        return fail("gut_optimize_rows_without_gradient: null pointer argument");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    if (!h->side_stream) {
        // default priority (GUT_SIDE_STREAM_PRIORITY=low|high for experiments): the kernel's fixed footprint is what keeps it
        // out of the way, not the queue priority — and with a lowest-priority queue the forward compositor was, in about one
        // process out of ten, 2-3x slower for the whole process (hardware queue arbitration; cause inferred)
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* pe = getenv("GUT_SIDE_STREAM_PRIORITY");
        int prio = 0;
        if (pe && pe[0] == 'l') prio = least;
        if (pe && pe[0] == 'h') prio = greatest;
        HIP_TRY(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, prio));
    }
    if (!h->ev_early_done) HIP_TRY(hipEventCreateWithFlags(&h->ev_early_done, kOrderingEvent));
    HIP_TRY(hipStreamWaitEvent(h->side_stream, h->ev_projected, 0));
    const bool timing = h->cfg.enable_kernel_timings != 0 && h->timing_side_stream && h->kev[14] && h->kev[15];
    if (timing) (void)hipEventRecord(h->kev[14], h->side_stream);
    static int split_percent = -1;
    if (split_percent < 0) {
        const char* e = getenv("GUT_EARLY_SPLIT");  // tuning experiments only
        split_percent = e ? atoi(e) : -1;
        if (split_percent > 100) split_percent = -1;
    }
    const uint32_t nblocks = (h->n + gut::kBlock - 1) / gut::kBlock;
    // share of the row blocks whose tile-less waves go to the first launch, beside the forward compositor: 60 % in the
    // 32-register form that launch has with lazy moments (about what it streams while K6 and the loss kernels run on the bench
    // frame: 50 / 75 / 100 % measured 2.30 / 2.29 / 2.40 ms per step), 25 % in the wide form (which takes K6's fifth wave)
    const uint32_t percent = split_percent >= 0 ? (uint32_t)split_percent : ((lazy && lazy->d_wave_step) ? 60u : 25u);
    const uint32_t first = (uint32_t)((uint64_t)nblocks * percent / 100u);
    gut::launch_adam_rows_without_gradient(h->side_stream, h->n, h->tiles_count.as<uint32_t>(), d_raw12, d_raw_m, d_raw_v, d_sh48,
                                           d_sh_m, d_sh_v, lr12, lr48, beta1, beta2, eps, step, d_act12_out, 0, first, nullptr, first, 0,
                                           false, gut_make_lazy(lazy, step));
    HIP_TRY(hipGetLastError());
    gut_context::EarlyArgs& ea = h->early_args;
    ea.raw12 = d_raw12; ea.raw_m = d_raw_m; ea.raw_v = d_raw_v; ea.sh48 = d_sh48; ea.sh_m = d_sh_m; ea.sh_v = d_sh_v;
    ea.act12 = d_act12_out;
    memcpy(ea.lr12, lr12, sizeof(ea.lr12));
    memcpy(ea.lr48, lr48, sizeof(ea.lr48));
    ea.beta1 = beta1; ea.beta2 = beta2; ea.eps = eps; ea.step = step; ea.block_begin = first; ea.block_end = nblocks;
    ea.lazy = GutLazyMoments{};
    if (lazy) ea.lazy = *lazy;
    // unsorted variant with something to walk: the second launch also takes, in the first early_extra_percent of the blocks,
    // the waves with tiles in which the forward walked no Gaussian (see launch_early_part2)
    ea.extra_end = (h->cfg.k_buffer_size == 0 && h->m) ? (uint32_t)((uint64_t)nblocks * (uint32_t)h->early_extra_percent / 100u) : 0u;
    // (wave_walked was sized and cleared by the forward)
    h->early_part2_pending = first < nblocks || ea.extra_end > 0;
    if (timing) {
        (void)hipEventRecord(h->kev[15], h->side_stream);  // re-recorded behind the second launch
        h->ring[h->ring_cur].early = true;
    }
    HIP_TRY(hipEventRecord(h->ev_early_done, h->side_stream));
    h->early_ran = true;
    h->early_wait_pending = true;
    return 0;
}

// second launch of the early optimiser pass, ordered behind the point of the main stream where the backward compositor starts
static int launch_early_part2(gut_context* h, hipStream_t s) {
    if (!h->early_part2_pending) return 0;
    h->early_part2_pending = false;
    if (!h->ev_bwd_start) HIP_TRY(hipEventCreateWithFlags(&h->ev_bwd_start, kOrderingEvent));
    const gut_context::EarlyArgs& ea = h->early_args;
    if (ea.extra_end) {
        // The forward compositor is done by now: mark the waves that hold a Gaussian it walked.  The backward compositor walks
        // no further than the forward did (it is bounded by the forward's per-tile depth), so every other wave is gradient-free.
        // (zero since K1 of this frame)
        gut::launch_mark_walked_waves(s, h->n, (uint32_t)h->tiles, h->ranges.as<uint32_t>(), h->trav_fwd.as<uint32_t>(),
                                      (h->lazy_order ? h->ids_ordered : h->ids_sorted).as<uint32_t>(), h->wave_walked.as<uint8_t>());
        h->marks_valid = true;
    }
    HIP_TRY(hipEventRecord(h->ev_bwd_start, s));
    HIP_TRY(hipStreamWaitEvent(h->side_stream, h->ev_bwd_start, 0));
    if (h->cfg.enable_kernel_timings != 0 && h->timing_side_stream && h->kev[7]) (void)hipEventRecord(h->kev[7], h->side_stream);  // start of the second launch
    gut::launch_adam_rows_without_gradient(h->side_stream, h->n, h->tiles_count.as<uint32_t>(), ea.raw12, ea.raw_m, ea.raw_v, ea.sh48,
                                           ea.sh_m, ea.sh_v, ea.lr12, ea.lr48, ea.beta1, ea.beta2, ea.eps, ea.step, ea.act12,
                                           ea.extra_end ? 0u : ea.block_begin, ea.block_end,
                                           ea.extra_end ? h->wave_walked.as<uint8_t>() : nullptr, ea.block_begin, ea.extra_end, true,
                                           gut_make_lazy(&ea.lazy, ea.step));
    HIP_TRY(hipGetLastError());
    const bool timing = h->cfg.enable_kernel_timings != 0 && h->timing_side_stream && h->kev[14] && h->kev[15];
    if (timing) {
        (void)hipEventRecord(h->kev[15], h->side_stream);
        h->ring[h->ring_cur].early2 = h->kev[7] != nullptr;
    }
    HIP_TRY(hipEventRecord(h->ev_early_done, h->side_stream));
    return 0;
}

int gut_set_option(gut_handle h, int32_t option, int32_t value) {
    if (!h) return fail("gut_set_option: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    switch (option) {
    case GUT_OPT_LAZY_TILE_ORDER: h->lazy_enabled = value != 0; return 0;
    case GUT_OPT_SORTED_REFERENCE_BACKWARD: h->sorted_reference_bwd = value != 0; return 0;
    case GUT_OPT_KERNEL_TIMING_SET:
        if (value < 0 || value > 2) return fail("gut_set_option: GUT_OPT_KERNEL_TIMING_SET takes 0, 1 or 2");
        h->timing_main_stream = value == 0;
        h->timing_side_stream = value != 2;
        return 0;
    case GUT_OPT_FORWARD_TILE_ORDER:
        if (value < -1 || value > 1) return fail("gut_set_option: GUT_OPT_FORWARD_TILE_ORDER takes -1, 0 or 1");
        h->fwd_order_mode = value;
        return 0;
    case GUT_OPT_EARLY_EXTRA_PERCENT:
        if (value < 0 || value > 100) return fail("gut_set_option: GUT_OPT_EARLY_EXTRA_PERCENT takes 0..100");
        h->early_extra_percent = value;
        return 0;
    case GUT_OPT_DEBUG_REPLACE_SCRATCH: {
        // developer probe (tools/scratch_placement.py): move ONE scratch buffer to a fresh allocation, contents kept
        DevBuf* bufs[] = {&h->tiles_count, &h->tiles_offset, &h->proj_pos, &h->conic_opacity, &h->extent, &h->depth, &h->feat,
                          &h->grad16, &h->scan_temp, &h->keys_unsorted, &h->keys_sorted, &h->ids_unsorted, &h->ids_sorted,
                          &h->sort_temp, &h->ids_ordered, &h->ranges, &h->trav_fwd, &h->trav_bwd, &h->tile_order, &h->tile_ordered};
        const int count = (int)(sizeof(bufs) / sizeof(bufs[0]));
        if (value < 0 || value >= count) return fail("gut_set_option: GUT_OPT_DEBUG_REPLACE_SCRATCH takes 0..%d", count - 1);
        DevBuf& b = *bufs[value];
        if (!b.p) return 0;
        DeviceGuard dev_guard;
        HIP_TRY(dev_guard.set(h->device));
        HIP_TRY(hipDeviceSynchronize());
        void* fresh = nullptr;
        HIP_TRY(hipMalloc(&fresh, b.cap));          // the old allocation is still held: this one is somewhere else
        HIP_TRY(hipMemcpy(fresh, b.p, b.cap, hipMemcpyDeviceToDevice));
        HIP_TRY(hipFree(b.p));
        b.p = fresh;
        return 0;
    }
    default: return fail("gut_set_option: unknown option %d", option);
    }
}

int gut_collect_times(gut_handle h, float* forward_render_ms, float* backward_render_ms) {
    if (!h) return fail("gut_collect_times: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    const float f = drain_timers(h->fwd_timers);
    const float b = drain_timers(h->bwd_timers);
    if (f >= 0.f) h->last_fwd_ms = f;  // m_timings persists between calls (splatRaster.cpp:357-363)
    if (b >= 0.f) h->last_bwd_ms = b;
    if (forward_render_ms) *forward_render_ms = h->last_fwd_ms;
    if (backward_render_ms) *backward_render_ms = h->last_bwd_ms;
    return 0;
}

int gut_kernel_times(gut_handle h, float* ms8) {
    if (!h || !ms8) return fail("gut_kernel_times: null argument");
    std::lock_guard<std::mutex> lock(h->mu);
    for (int i = 0; i < GUT_NUM_KERNEL_TIMERS; ++i) ms8[i] = -1.f;
    if (!h->cfg.enable_kernel_timings) return fail("gut_kernel_times: enable_kernel_timings is off");
    auto span = [&](int a, int b) -> float {
        float ms = -1.f;
        if (h->kev[a] && h->kev[b] && hipEventSynchronize(h->kev[b]) == hipSuccess) (void)hipEventElapsedTime(&ms, h->kev[a], h->kev[b]);
        return ms;
    };
    if (h->kev_fwd_valid) {
        ms8[0] = span(0, 1);  // project
        ms8[1] = span(1, 2);  // scan (+ count readback)
        ms8[2] = span(2, 3);  // expand
        ms8[3] = span(3, 4);  // sort
        ms8[4] = span(4, 5);  // ranges
        ms8[5] = span(5, 6);  // render
    }
    if (h->kev_bwd_valid) {
        ms8[6] = span(9, 10);   // render backward
        ms8[7] = span(10, 11);  // project backward
    }
    if (h->ring[h->ring_cur].opt) ms8[8] = span(12, 13);  // one-pass optimiser (gut_optimize_after_bwd)
    if (h->ring[h->ring_cur].early) ms8[9] = span(14, 15);  // side-stream pass, start of its first to end of its second launch
    if (h->ring[h->ring_cur].early2) ms8[10] = span(7, 15);  // its second launch alone
    return 0;
}

int gut_kernel_times_mean(gut_handle h, float* ms8, int32_t* count) {
    if (!h || !ms8) return fail("gut_kernel_times_mean: null argument");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->cfg.enable_kernel_timings) return fail("gut_kernel_times_mean: enable_kernel_timings is off");
    double sum[GUT_NUM_KERNEL_TIMERS] = {};
    int cnt[GUT_NUM_KERNEL_TIMERS] = {};
    static const int kA[GUT_NUM_KERNEL_TIMERS] = {0, 1, 2, 3, 4, 5, 9, 10, 12, 14, 7};
    static const int kB[GUT_NUM_KERNEL_TIMERS] = {1, 2, 3, 4, 5, 6, 10, 11, 13, 15, 15};
    for (int k = 0; k < h->ring_count; ++k) {
        const auto& set = h->ring[(h->ring_cur - k + 2 * gut_context::kRing) % gut_context::kRing];
        for (int i = 0; i < GUT_NUM_KERNEL_TIMERS; ++i) {
            const bool ok = i < 6 ? set.fwd : (i < 8 ? set.bwd : (i == 8 ? set.opt : (i == 9 ? set.early : set.early2)));
            if (!ok || !set.e[kA[i]] || !set.e[kB[i]]) continue;
            float ms = 0.f;
            if (hipEventSynchronize(set.e[kB[i]]) == hipSuccess && hipEventElapsedTime(&ms, set.e[kA[i]], set.e[kB[i]]) == hipSuccess) {
                sum[i] += ms;
                cnt[i]++;
            }
        }
    }
    for (int i = 0; i < GUT_NUM_KERNEL_TIMERS; ++i) ms8[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : -1.f;
    if (count) *count = h->ring_count;
    h->ring_count = 0;
    return 0;
}

int gut_get_stats(gut_handle h, GutStats* out) {
    if (!h || !out) return fail("gut_get_stats: null argument");
    std::lock_guard<std::mutex> lock(h->mu);
    memset(out, 0, sizeof(*out));
    if (!h->have_forward) return fail("gut_get_stats: no forward yet");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    HIP_TRY(hipMemsetAsync(h->counters.p, 0, sizeof(gut::Counters), h->fwd_stream));
    gut::launch_stats_reduce(h->fwd_stream, h->n, h->tiles_count.as<uint32_t>(), (uint32_t)h->tiles, h->trav_fwd.as<uint32_t>(),
                             h->trav_bwd.as<uint32_t>(), h->counters.as<gut::Counters>());
    if (h->stats_early)
        gut::launch_count_side_stream_rows(h->fwd_stream, h->n, h->tiles_count.as<uint32_t>(),
                                           h->stats_extra_end ? h->wave_walked.as<uint8_t>() : nullptr, h->stats_split,
                                           h->stats_extra_end, h->counters.as<gut::Counters>());
    gut::Counters c;
    HIP_TRY(hipMemcpyAsync(&c, h->counters.p, sizeof(c), hipMemcpyDeviceToHost, h->fwd_stream));
    HIP_TRY(hipStreamSynchronize(h->fwd_stream));
    out->num_particles = h->n;
    out->num_visible = c.visible;
    out->num_intersections = h->m;
    out->num_tiles = (uint64_t)h->tiles;
    out->num_pixels = (uint64_t)h->width * (uint64_t)h->height;
    out->traversed_fwd = c.traversed_fwd;
    out->traversed_bwd = c.traversed_bwd;
    out->sort_end_bit = (uint32_t)h->end_bit;
    out->binning_overflows = (uint32_t)h->overflows;
    out->side_stream_rows = c.side_stream_rows;
    out->side_stream_rows_first_launch = c.side_stream_rows_first;
    return 0;
}

int gut_debug_buffer(gut_handle h, int32_t which, void** d_ptr, size_t* bytes) {
    if (!h || !d_ptr || !bytes) return fail("gut_debug_buffer: null argument");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->have_forward) return fail("gut_debug_buffer: no forward yet");
    const size_t n = h->n, m = h->m, t = (size_t)h->tiles;
    switch (which) {
    case GUT_BUF_TILES_COUNT: *d_ptr = h->tiles_count.p; *bytes = 4 * n; break;
    case GUT_BUF_TILES_OFFSET:
        // debug view: the product path never materialises the [N] inclusive scan (see K2 in trace_fwd_impl); built here on request
        if (!h->dbg_offset_valid && n) {
            DeviceGuard dev_guard;
            HIP_TRY(dev_guard.set(h->device));
            HIP_TRY(h->tiles_offset.ensure(sizeof(uint32_t) * (size_t)n));
            HIP_TRY(h->scan_temp.ensure(gut::scan_temp_bytes(n)));
            HIP_TRY(gut::run_scan(h->fwd_stream, h->scan_temp.p, h->scan_temp.cap, h->tiles_count.as<uint32_t>(), h->tiles_offset.as<uint32_t>(), n));
            HIP_TRY(hipStreamSynchronize(h->fwd_stream));
            h->dbg_offset_valid = true;
        }
        *d_ptr = h->tiles_offset.p; *bytes = 4 * n; break;
    case GUT_BUF_PROJ_POSITION: *d_ptr = h->proj_pos.p; *bytes = 8 * n; break;
    case GUT_BUF_CONIC_OPACITY: *d_ptr = h->conic_opacity.p; *bytes = 16 * n; break;
    case GUT_BUF_PROJ_EXTENT: *d_ptr = h->extent.p; *bytes = 8 * n; break;
    case GUT_BUF_GLOBAL_DEPTH: *d_ptr = h->depth.p; *bytes = 4 * n; break;
    case GUT_BUF_FEATURES: *d_ptr = h->feat.p; *bytes = 12 * n; break;
    case GUT_BUF_UNSORTED_KEYS: *d_ptr = h->keys_unsorted.p; *bytes = 8 * m; break;
    case GUT_BUF_UNSORTED_IDS: *d_ptr = h->ids_unsorted.p; *bytes = 4 * m; break;
    case GUT_BUF_SORTED_KEYS:
    case GUT_BUF_SORTED_IDS:
        if (h->lazy_order) {  // the product path never needs the fully sorted lists: build them for the caller
            if (!h->dbg_sorted_valid && m) {
                DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
                HIP_TRY(h->dbg_keys_sorted.ensure(8 * m));
                HIP_TRY(h->dbg_ids_sorted.ensure(4 * m));
                DevBuf tmp;
                HIP_TRY(tmp.ensure(gut::sort_temp_bytes((uint32_t)m, h->end_bit)));
                hipError_t e = gut::run_sort(h->fwd_stream, tmp.p, tmp.cap, h->keys_unsorted.as<uint64_t>(),
                                             h->dbg_keys_sorted.as<uint64_t>(), h->ids_unsorted.as<uint32_t>(),
                                             h->dbg_ids_sorted.as<uint32_t>(), (uint32_t)m, h->end_bit);
                if (e == hipSuccess) e = hipStreamSynchronize(h->fwd_stream);
                tmp.release();
                HIP_TRY(e);
                h->dbg_sorted_valid = true;
            }
            if (which == GUT_BUF_SORTED_KEYS) { *d_ptr = h->dbg_keys_sorted.p; *bytes = 8 * m; }
            else { *d_ptr = h->dbg_ids_sorted.p; *bytes = 4 * m; }
        } else if (which == GUT_BUF_SORTED_KEYS) { *d_ptr = h->keys_sorted.p; *bytes = 8 * m; }
        else { *d_ptr = h->ids_sorted.p; *bytes = 4 * m; }
        break;
    case GUT_BUF_ORDERED_IDS:
        if (h->lazy_order && !h->dbg_ordered_valid && m) {
            // the product path tracks the walked prefix of every tile in tile_ordered; for the caller the positions behind
            // it are filled with padding ids (what the backward treats them as)
            DeviceGuard dev_guard;
            HIP_TRY(dev_guard.set(h->device));
            gut::launch_mask_unordered(h->fwd_stream, (uint32_t)h->tiles, h->ranges.as<uint32_t>(), h->tile_ordered.as<uint32_t>(),
                                       h->ids_ordered.as<uint32_t>());
            HIP_TRY(hipStreamSynchronize(h->fwd_stream));
            h->dbg_ordered_valid = true;
        }
        *d_ptr = h->lazy_order ? h->ids_ordered.p : h->ids_sorted.p; *bytes = 4 * m; break;
    case GUT_BUF_PACKED_ROWS:
        if (!h->packed_valid) return fail("gut_debug_buffer: the last forward was not a field-wise one (no packed rows)");
        *d_ptr = h->packed12.p; *bytes = 48 * n; break;
    case GUT_BUF_TILE_RANGES: *d_ptr = h->ranges.p; *bytes = 8 * t; break;
    case GUT_BUF_GRAD_SCRATCH:
        if (!h->have_backward) return fail("gut_debug_buffer: no backward yet");
        *d_ptr = h->grad16.p; *bytes = 64 * n; break;
    case GUT_BUF_TILE_TRAVERSED_FWD: *d_ptr = h->trav_fwd.p; *bytes = 4 * t; break;
    case GUT_BUF_TILE_TRAVERSED_BWD:
        if (!h->have_backward) return fail("gut_debug_buffer: no backward yet");
        *d_ptr = h->trav_bwd.p; *bytes = 4 * t; break;
    default: return fail("gut_debug_buffer: unknown buffer %d", which);
    }
    return 0;
}

int gut_debug_copy(gut_handle h, int32_t which, void* d_dst, size_t bytes) {
    void* src = nullptr;
    size_t have = 0;
    if (gut_debug_buffer(h, which, &src, &have)) return 1;
    if (bytes < have) return fail("gut_debug_copy: destination holds %zu bytes, buffer has %zu", bytes, have);
    if (have == 0) return 0;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    HIP_TRY(hipMemcpyAsync(d_dst, src, have, hipMemcpyDeviceToDevice, h->fwd_stream));
    HIP_TRY(hipStreamSynchronize(h->fwd_stream));
    return 0;
}

}  // extern "C"

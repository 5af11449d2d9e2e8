// gut_api.cpp — host side of libgut_hip.so: handle, grow-only scratch, launch orchestration and the C ABI
// declared in include/gut_hip.h.  Replaces, for the 3DGUT path, the reference's SplatRaster
// (src/splatRaster.cpp:153-364) and GUTRenderer (src/gutRenderer.cu:99-497).
//
// Compiled with -ffp-contract=off: the pose matrices built here feed the bit-exact projection kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
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
    // per-M scratch
    DevBuf keys_unsorted, keys_sorted, ids_unsorted, ids_sorted, sort_temp;
    // lazy per-tile depth order (unsorted variant): keys_sorted / ids_sorted are grouped by tile only, the forward compositor
    // writes the ids it consumed, in final order, to ids_ordered; the fully sorted lists exist only on debug request
    DevBuf ids_ordered, dbg_keys_sorted, dbg_ids_sorted;
    bool lazy_enabled = true;      // gut_set_option(GUT_OPT_LAZY_TILE_ORDER)
    bool lazy_order = false;       // this forward used the lazy order
    bool dbg_sorted_valid = false;
    // per-T
    DevBuf ranges, trav_fwd, trav_bwd, tile_order;  // per-tile traversal depths (statistics)
    DevBuf counters;
    uint32_t* host_count = nullptr;  // pinned
    hipEvent_t count_event = nullptr;  // the count read-back has landed (work queued behind it keeps the GPU busy meanwhile)
    bool trains = false;               // a backward has run on this handle: forwards pre-clear the gradient rows
    bool grad16_clean = false;         // ... and this says the rows are already zero when the backward starts

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
        hipEvent_t e[14] = {};
        bool fwd = false, bwd = false, opt = false;
    };
    KevSet ring[kRing];
    int ring_cur = 0;    // set used by the most recent trace()
    int ring_count = 0;  // sets recorded since the last gut_kernel_times_mean
    hipEvent_t* kev = ring[0].e;
    bool kev_fwd_valid = false, kev_bwd_valid = false;
};

namespace {

int build_view(const GutCamera* cam, int W, int H, gut::ViewParams* v) {
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
    c->ut_delta = sqrtf(cfg.ut_alpha * cfg.ut_alpha * (D + cfg.ut_kappa));
    c->ut_w0_mean = lambda / (D + lambda);
    c->ut_wi = 1.0f / (2.0f * (D + lambda));
    c->ut_w0_cov = lambda / (D + lambda) + (1.0f - cfg.ut_alpha * cfg.ut_alpha + cfg.ut_beta);
    c->ut_margin = cfg.ut_in_image_margin_factor;
    c->rect_bounding = cfg.rect_bounding;
    c->tight_opacity_bounding = cfg.tight_opacity_bounding;
    c->tile_culling = cfg.tile_based_culling;
    c->global_z_order = cfg.global_z_order;
    c->max_d2 = -2.0f * logf(cfg.particle_kernel_min_response);
}

EventPair* arm_timer(std::deque<EventPair>& q, hipStream_t s) {
    if (q.size() >= 256) {  // keep only the most recent (splatRaster.cpp:212-215)
        EventPair old = q.front();
        q.pop_front();
        if (old.a) (void)hipEventDestroy(old.a);
        if (old.b) (void)hipEventDestroy(old.b);
    }
    EventPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
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
    if (cfg->particle_kernel_degree != 2) return fail("particle_kernel_degree=%d: only the quadratic (2) kernel of render/3dgut.yaml is built", cfg->particle_kernel_degree);
    if (cfg->k_buffer_size < 0 || cfg->k_buffer_size > 16)
        return fail("k_buffer_size=%d: supported range is 0 (unsorted) .. 16", cfg->k_buffer_size);
    if (cfg->particle_radiance_sph_degree != 3) return fail("particle_radiance_sph_degree=%d: only degree 3 (16 coefficients) is built", cfg->particle_radiance_sph_degree);
    if (cfg->ut_require_all_sigma_points != 0) return fail("ut_require_all_sigma_points must be false (static_assert in threedgut.cuh:73)");
    if (!cfg->enable_hitcounts) return fail("enable_hitcounts=false is not built");
    if (cfg->n_rolling_shutter_iterations != 5) return fail("n_rolling_shutter_iterations=%d: only 5 (render/3dgut.yaml) is built", cfg->n_rolling_shutter_iterations);
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
    hipError_t e = hipHostMalloc((void**)&h->host_count, 64, hipHostMallocDefault);
    if (e == hipSuccess) e = h->counters.ensure(sizeof(gut::Counters));
    if (e != hipSuccess) {
        delete h;
        return fail("gut_create: allocation failed: %s", hipGetErrorString(e));
    }
    if (cfg->enable_kernel_timings)
        for (auto& set : h->ring)
            for (auto& ev : set.e)
                if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
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
                      &h->dbg_keys_sorted, &h->dbg_ids_sorted};
    for (DevBuf* b : bufs) b->release();
    if (h->host_count) (void)hipHostFree(h->host_count);
    if (h->count_event) (void)hipEventDestroy(h->count_event);
    (void)drain_timers(h->fwd_timers);
    (void)drain_timers(h->bwd_timers);
    for (auto& set : h->ring)
        for (auto& ev : set.e)
            if (ev) (void)hipEventDestroy(ev);
    delete h;
}

int gut_trace(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
              const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
              const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
              float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
              float* d_particle_visibility) {
    (void)frame_number;
    if (!h) return fail("gut_trace: null handle");
    if (!camera || !d_ray_origin || !d_ray_direction || !d_ray_radiance_density || !d_ray_hit_distance || !d_ray_hit_count)
        return fail("gut_trace: null pointer argument");
    if (width <= 0 || height <= 0) return fail("gut_trace: bad resolution %dx%d", width, height);
    if (num_active_features < 0 || num_active_features > 3) return fail("gut_trace: SH degree %d outside 0..3", num_active_features);
    if (num_particles && (!d_particle_density || !d_particle_radiance || !d_particle_visibility))
        return fail("gut_trace: null particle buffers with %u particles", num_particles);
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    gut::ViewParams v;
    if (build_view(camera, width, height, &v)) return 1;
    const uint32_t n = num_particles;
    const int tiles = v.grid_x * v.grid_y;
    if ((uint64_t)tiles >= 0xFFFFFFFFull) return fail("gut_trace: too many tiles");
    h->have_forward = false;
    h->have_backward = false;

    HIP_TRY(h->tiles_count.ensure(sizeof(uint32_t) * (size_t)n));
    HIP_TRY(h->tiles_offset.ensure(sizeof(uint32_t) * (size_t)n));
    HIP_TRY(h->proj_pos.ensure(sizeof(float) * 2 * (size_t)n));
    HIP_TRY(h->conic_opacity.ensure(sizeof(float) * 4 * (size_t)n));
    HIP_TRY(h->extent.ensure(sizeof(float) * 2 * (size_t)n));
    HIP_TRY(h->depth.ensure(sizeof(float) * (size_t)n));
    HIP_TRY(h->feat.ensure(sizeof(float) * 3 * (size_t)n));
    HIP_TRY(h->ranges.ensure(sizeof(uint32_t) * 2 * (size_t)tiles));
    HIP_TRY(h->trav_fwd.ensure(sizeof(uint32_t) * (size_t)tiles));
    HIP_TRY(h->trav_bwd.ensure(sizeof(uint32_t) * (size_t)tiles));
    HIP_TRY(h->tile_order.ensure(sizeof(uint32_t) * (size_t)tiles));
    if (n) HIP_TRY(h->scan_temp.ensure(gut::scan_temp_bytes(n)));

    const bool timing = h->cfg.enable_kernel_timings != 0;
    EventPair* total = timing ? arm_timer(h->fwd_timers, s) : nullptr;
    h->kev_fwd_valid = false;
    if (timing) {
        h->ring_cur = (h->ring_cur + 1) % gut_context::kRing;
        h->kev = h->ring[h->ring_cur].e;
        h->ring[h->ring_cur].fwd = false;
        h->ring[h->ring_cur].bwd = false;
        h->ring[h->ring_cur].opt = false;
    }
    auto mark = [&](int i) {
        if (timing && h->kev[i]) (void)hipEventRecord(h->kev[i], s);
    };

    HIP_TRY(hipMemsetAsync(h->trav_bwd.p, 0, sizeof(uint32_t) * (size_t)tiles, s));
    mark(0);
    gut::launch_project(s, v, h->consts, n, num_active_features, d_particle_density, d_particle_radiance,
                        h->tiles_count.as<uint32_t>(), h->proj_pos.as<float>(), h->conic_opacity.as<float>(),
                        h->extent.as<float>(), h->depth.as<float>(), h->feat.as<float>(), d_particle_visibility,
                        h->counters.as<gut::Counters>());
    mark(1);
    uint32_t m = 0;
    if (n) {
        HIP_TRY(gut::run_scan(s, h->scan_temp.p, h->scan_temp.cap, h->tiles_count.as<uint32_t>(), h->tiles_offset.as<uint32_t>(), n));
        // intersection count readback: the one host sync of the path (gutRenderer.cu:313-321)
        HIP_TRY(hipMemcpyAsync(h->host_count, h->tiles_offset.as<uint32_t>() + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        if (!h->count_event) HIP_TRY(hipEventCreateWithFlags(&h->count_event, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->count_event, s));
        // work that does not depend on the count is queued BEHIND the read-back and runs while the host waits for it:
        // the tile-range clear and, on a handle that trains, the clear of the 64-byte gradient rows of the coming backward
        HIP_TRY(hipMemsetAsync(h->ranges.p, 0, sizeof(uint32_t) * 2 * (size_t)tiles, s));
        h->grad16_clean = false;
        if (h->trains) {
            HIP_TRY(h->grad16.ensure(sizeof(float) * 16 * (size_t)n));
            HIP_TRY(hipMemsetAsync(h->grad16.p, 0, sizeof(float) * 16 * (size_t)n, s));
            h->grad16_clean = true;
        }
        HIP_TRY(hipEventSynchronize(h->count_event));
        m = *h->host_count;
    } else {
        HIP_TRY(hipMemsetAsync(h->ranges.p, 0, sizeof(uint32_t) * 2 * (size_t)tiles, s));
    }
    mark(2);
    const int end_bit = 32 + (int)bit_width_u32((uint32_t)tiles);
    // the k-buffer variant walks whole lists: it keeps the full sort
    // ... and frames whose tile lists are so long on average that re-scanning them per 512 ordered entries cannot pay
    const bool lazy = h->lazy_enabled && h->cfg.k_buffer_size == 0 && m != 0 && (uint64_t)m < (uint64_t)tiles * 16384ull;
    h->lazy_order = lazy;
    h->dbg_sorted_valid = false;
    if (m) {
        HIP_TRY(h->keys_unsorted.ensure(sizeof(uint64_t) * (size_t)m));
        HIP_TRY(h->keys_sorted.ensure(sizeof(uint64_t) * (size_t)m));
        HIP_TRY(h->ids_unsorted.ensure(sizeof(uint32_t) * (size_t)m));
        HIP_TRY(h->ids_sorted.ensure(sizeof(uint32_t) * (size_t)m));
        gut::launch_expand(s, v, h->consts, n, h->tiles_offset.as<uint32_t>(), h->proj_pos.as<float>(),
                           h->conic_opacity.as<float>(), h->extent.as<float>(), h->depth.as<float>(),
                           h->keys_unsorted.as<uint64_t>(), h->ids_unsorted.as<uint32_t>());
        mark(3);
        if (lazy) {
            HIP_TRY(h->sort_temp.ensure(gut::sort_tiles_temp_bytes(m, end_bit)));
            HIP_TRY(h->ids_ordered.ensure(sizeof(uint32_t) * (size_t)m));
            // unwritten positions read as padding ids (end of list) in the backward
            HIP_TRY(hipMemsetAsync(h->ids_ordered.p, 0xFF, sizeof(uint32_t) * (size_t)m, s));
            HIP_TRY(gut::run_sort_tiles(s, h->sort_temp.p, h->sort_temp.cap, h->keys_unsorted.as<uint64_t>(),
                                        h->keys_sorted.as<uint64_t>(), h->ids_unsorted.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), m,
                                        end_bit));
        } else {
            HIP_TRY(h->sort_temp.ensure(gut::sort_temp_bytes(m, end_bit)));
            HIP_TRY(gut::run_sort(s, h->sort_temp.p, h->sort_temp.cap, h->keys_unsorted.as<uint64_t>(), h->keys_sorted.as<uint64_t>(),
                                  h->ids_unsorted.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), m, end_bit));
        }
        mark(4);
        gut::launch_tile_ranges(s, m, h->keys_sorted.as<uint64_t>(), h->ranges.as<uint32_t>());
    } else {
        mark(3);
        mark(4);
    }
    mark(5);
    // with zero intersections the reference returns its freshly initialised outputs (gutRenderer.cu:323-325);
    // running the compositor over empty ranges writes exactly those values
    if (h->cfg.k_buffer_size > 0) {
        HIP_TRY(hipMemsetAsync(h->trav_fwd.p, 0, sizeof(uint32_t) * (size_t)tiles, s));
        gut::launch_render_sorted(s, v, h->consts, h->cfg.k_buffer_size, d_particle_density, h->feat.as<float>(), d_ray_origin,
                                  d_ray_direction, h->ranges.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), m,
                                  d_ray_radiance_density, d_ray_hit_distance, d_ray_hit_count);
    } else {
        gut::launch_render(s, v, h->consts, d_particle_density, h->feat.as<float>(), d_ray_origin, d_ray_direction,
                           h->ranges.as<uint32_t>(), h->ids_sorted.as<uint32_t>(), m, d_ray_radiance_density, d_ray_hit_distance,
                           d_ray_hit_count, h->trav_fwd.as<uint32_t>(), lazy ? h->keys_sorted.as<uint64_t>() : nullptr,
                           lazy ? h->ids_ordered.as<uint32_t>() : nullptr);
    }
    mark(6);
    HIP_TRY(hipGetLastError());
    if (total) {
        (void)hipEventRecord(total->b, s);
        total->armed = true;
    }
    h->kev_fwd_valid = timing;
    if (timing) {
        h->ring[h->ring_cur].fwd = true;
        if (h->ring_count < gut_context::kRing) h->ring_count++;
    }
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

int gut_trace_bwd_ex(gut_handle h, void* stream_, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                     const float* d_particle_density, const float* d_particle_radiance, int32_t width, int32_t height,
                     const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                     const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                     const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_particle_density_grad,
                     float* d_particle_radiance_grad, uint32_t flags) {
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
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    gut::ViewParams v;
    if (build_view(camera, width, height, &v)) return 1;
    if (memcmp(&v, &h->view, sizeof(v)) != 0) return fail("gut_trace_bwd: camera differs from the cached forward");
    const uint32_t n = h->n;
    if (n == 0) return 0;
    HIP_TRY(h->grad16.ensure(sizeof(float) * 16 * (size_t)n));

    const bool timing = h->cfg.enable_kernel_timings != 0;
    EventPair* total = timing ? arm_timer(h->bwd_timers, s) : nullptr;
    h->kev_bwd_valid = false;
    auto mark = [&](int i) {
        if (timing && h->kev[i]) (void)hipEventRecord(h->kev[i], s);
    };
    mark(8);
    if (!h->grad16_clean) HIP_TRY(hipMemsetAsync(h->grad16.p, 0, sizeof(float) * 16 * (size_t)n, s));
    h->grad16_clean = false;
    h->trains = true;
    mark(9);
    if (h->m && h->cfg.k_buffer_size > 0) {
        if (!d_ray_hit_distance) return fail("gut_trace_bwd: the sorted variant needs d_ray_hit_distance");
        HIP_TRY(hipMemsetAsync(h->trav_bwd.p, 0, sizeof(uint32_t) * (size_t)h->tiles, s));
        gut::launch_render_sorted_bwd(s, v, h->consts, h->cfg.k_buffer_size, d_particle_density, h->feat.as<float>(), d_ray_origin,
                                      d_ray_direction, h->ranges.as<uint32_t>(), h->ids_sorted.as<uint32_t>(),
                                      d_ray_radiance_density, d_ray_hit_distance, d_ray_radiance_density_grad,
                                      d_ray_hit_distance_grad, h->grad16.as<float>());
    } else if (h->m) {
        gut::launch_tile_order(s, (uint32_t)h->tiles, h->trav_fwd.as<uint32_t>(), h->tile_order.as<uint32_t>());
        gut::launch_render_bwd(s, v, h->consts, d_particle_density, h->feat.as<float>(), d_ray_origin, d_ray_direction,
                               h->ranges.as<uint32_t>(), (h->lazy_order ? h->ids_ordered : h->ids_sorted).as<uint32_t>(),
                               d_ray_radiance_density,
                               d_ray_radiance_density_grad, d_ray_hit_distance_grad, h->grad16.as<float>(),
                               h->trav_bwd.as<uint32_t>(), h->tile_order.as<uint32_t>());
    }
    mark(10);
    if (flags & GUT_BWD_SKIP_EPILOGUE) {
        // the caller folds the per-Gaussian epilogue into its optimiser step (gut_optimize_after_bwd)
    } else if (flags & GUT_BWD_COMPACT_RADIANCE_GRADS)
        gut::launch_project_bwd_compact(s, n, d_particle_density, h->tiles_count.as<uint32_t>(), h->feat.as<float>(),
                                        h->grad16.as<float>(), d_particle_density_grad, d_particle_radiance_grad);
    else
        gut::launch_project_bwd(s, v, n, h->sh_degree, d_particle_density, h->tiles_count.as<uint32_t>(), h->feat.as<float>(),
                                h->grad16.as<float>(), d_particle_density_grad, d_particle_radiance_grad,
                                (flags & GUT_BWD_RAW_PARAMETER_GRADS) != 0);
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
                           const float* d_visibility, float* d_act12_out) {
    if (!h) return fail("gut_optimize_after_bwd: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (!h->have_backward || h->fwd_stream != s)
        return fail("gut_optimize_after_bwd: no backward context on this stream (call gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE) first)");
    if (num_active_features != h->sh_degree) return fail("gut_optimize_after_bwd: sh degree differs from the cached forward");
    if (h->n == 0) return 0;
    if (!d_camera_position || !d_raw12 || !d_raw_m || !d_raw_v || !d_sh48 || !d_sh_m || !d_sh_v || !lr12 || !lr48)
        return fail("gut_optimize_after_bwd: null pointer argument");
    DeviceGuard dev_guard;
    HIP_TRY(dev_guard.set(h->device));
    const bool timing = h->cfg.enable_kernel_timings != 0 && h->kev[12] && h->kev[13];
    if (timing) (void)hipEventRecord(h->kev[12], s);
    gut::launch_sh_adam_from_scratch(s, h->n, h->sh_degree, d_camera_position, h->grad16.as<float>(), h->tiles_count.as<uint32_t>(),
                                     h->feat.as<float>(), d_raw12, d_raw_m, d_raw_v, d_sh48, d_sh_m, d_sh_v, lr12, lr48, beta1, beta2,
                                     eps, step, d_visibility, d_act12_out);
    HIP_TRY(hipGetLastError());
    if (timing) {
        (void)hipEventRecord(h->kev[13], s);
        h->ring[h->ring_cur].opt = true;
    }
    h->have_backward = false;  // the gradient rows are consumed
    return 0;
}

int gut_set_option(gut_handle h, int32_t option, int32_t value) {
    if (!h) return fail("gut_set_option: null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    switch (option) {
    case GUT_OPT_LAZY_TILE_ORDER: h->lazy_enabled = value != 0; return 0;
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
    return 0;
}

int gut_kernel_times_mean(gut_handle h, float* ms8, int32_t* count) {
    if (!h || !ms8) return fail("gut_kernel_times_mean: null argument");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->cfg.enable_kernel_timings) return fail("gut_kernel_times_mean: enable_kernel_timings is off");
    double sum[GUT_NUM_KERNEL_TIMERS] = {};
    int cnt[GUT_NUM_KERNEL_TIMERS] = {};
    static const int kA[GUT_NUM_KERNEL_TIMERS] = {0, 1, 2, 3, 4, 5, 9, 10, 12};
    static const int kB[GUT_NUM_KERNEL_TIMERS] = {1, 2, 3, 4, 5, 6, 10, 11, 13};
    for (int k = 0; k < h->ring_count; ++k) {
        const auto& set = h->ring[(h->ring_cur - k + 2 * gut_context::kRing) % gut_context::kRing];
        for (int i = 0; i < GUT_NUM_KERNEL_TIMERS; ++i) {
            const bool ok = i < 6 ? set.fwd : (i < 8 ? set.bwd : set.opt);
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
    return 0;
}

int gut_debug_buffer(gut_handle h, int32_t which, void** d_ptr, size_t* bytes) {
    if (!h || !d_ptr || !bytes) return fail("gut_debug_buffer: null argument");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->have_forward) return fail("gut_debug_buffer: no forward yet");
    const size_t n = h->n, m = h->m, t = (size_t)h->tiles;
    switch (which) {
    case GUT_BUF_TILES_COUNT: *d_ptr = h->tiles_count.p; *bytes = 4 * n; break;
    case GUT_BUF_TILES_OFFSET: *d_ptr = h->tiles_offset.p; *bytes = 4 * n; break;
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
        *d_ptr = h->lazy_order ? h->ids_ordered.p : h->ids_sorted.p; *bytes = 4 * m; break;
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

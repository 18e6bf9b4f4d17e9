// api.cpp -- the extern "C" surface of libnerfhip.so (include/nerfhip.h).
// Argument checking happens here, on the host, before any kernel is launched: a
// launch only goes out once every shape the kernel and its grid assume has been
// verified.  No C++ exception leaves this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/nerfhip.h"
#include "kernels.hpp"
#include "packing.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(NRF_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define NRF_HIP(call)                                    \
    do {                                                 \
        const hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) return;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

bool copy_linears(const nrf_linear* in, int n, std::vector<nrf::HostLinear>& out, std::string& err) {
    out.resize(n);
    for (int i = 0; i < n; ++i) {
        if (!in[i].weight || !in[i].bias || in[i].out_f <= 0 || in[i].in_f <= 0) {
            err = "linear " + std::to_string(i) + ": null pointer or non-positive shape";
            return false;
        }
        out[i].out_f = in[i].out_f;
        out[i].in_f = in[i].in_f;
        out[i].w.assign(in[i].weight, in[i].weight + (size_t)in[i].out_f * in[i].in_f);
        out[i].b.assign(in[i].bias, in[i].bias + in[i].out_f);
    }
    return true;
}

nrf::Camera make_camera(int H, int W, float focal, const float c2w[12]) {
    nrf::Camera c;
    c.H = H; c.W = W; c.focal = focal;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.r[i][j] = c2w[4 * i + j];
        c.t[i] = c2w[4 * i + 3];
    }
    return c;
}

bool make_dino(const nrf_dino* in, int want_c, nrf::DinoDev& d, std::string& err) {
    if (!in || !in->features) { err = "dino side channel missing (features == NULL)"; return false; }
    if (in->Hp < 1 || in->Wp < 1 || in->C < 1 || in->H < 1 || in->W < 1) { err = "dino: bad map / image size"; return false; }
    if (want_c > 0 && in->C != want_c) { err = "dino: channel count differs from the model's dino_dim"; return false; }
    d.features = in->features; d.Hp = in->Hp; d.Wp = in->Wp; d.C = in->C;
    std::memcpy(d.inv_pose, in->inv_pose, sizeof(float) * 12);
    d.focal = in->focal; d.H = in->H; d.W = in->W;
    return true;
}

}  // namespace

struct nrf_model {
    nrf_arch arch{};
    int device = 0;
    nrf::NetPlan plan;
    std::vector<nrf::HostLinear> lin;
    nrf::PackedStream h_stream[nrf::kModes];
    std::vector<float> h_bias;
    void* d_stream[nrf::kModes] = {nullptr, nullptr, nullptr, nullptr};
    float* d_bias = nullptr;
    unsigned long long* d_queues = nullptr;
    nrf::DeviceNet net{};
    // training path (built on first use: ensure_train)
    bool train_ready = false;
    bool lin_stale = false;                 // parameters were last set from a device vector: the host copy `lin` is old
    bool bfresh[3] = {false, false, false}; // backward stream of the mode matches the current parameters
    nrf::ParamLayout layout;
    nrf::NetPlan bplan;
    nrf::TrainPlan tplan;
    nrf::TrainDev train{};
    void* d_bstream[3] = {nullptr, nullptr, nullptr};
    int32_t* d_maps = nullptr;
    int32_t* d_src[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // [forward|backward][packing.hpp stream kind] element sources
    int64_t n_src[2][3] = {{0, 0, 0}, {0, 0, 0}};                                        // (no backward stream in the split mode)
    int32_t* d_bias_src = nullptr;
};

namespace {

int upload(nrf_model* m, hipStream_t s, bool allocate) {
    for (int mode = 0; mode < nrf::kModes; ++mode) {
        m->h_stream[mode] = nrf::pack_stream(m->plan, m->lin, mode);
        const size_t bytes = m->h_stream[mode].bytes.size();
        if (allocate) NRF_HIP(hipMalloc(&m->d_stream[mode], bytes));
        NRF_HIP(hipMemcpyAsync(m->d_stream[mode], m->h_stream[mode].bytes.data(), bytes, hipMemcpyHostToDevice, s));
        m->net.stream[mode] = m->d_stream[mode];
        m->net.n_chunks[mode] = m->h_stream[mode].n_chunks;
    }
    m->lin_stale = false;
    if (m->train_ready) {
        for (int mode = 0; mode < 3; ++mode) {
            const nrf::PackedStream ps = nrf::pack_stream(m->bplan, m->lin, mode);
            NRF_HIP(hipMemcpyAsync(m->d_bstream[mode], ps.bytes.data(), ps.bytes.size(), hipMemcpyHostToDevice, s));
            NRF_HIP(hipStreamSynchronize(s));
            m->bfresh[mode] = true;
        }
    }
    m->h_bias = nrf::pack_bias(m->plan, m->lin);
    if (allocate) NRF_HIP(hipMalloc((void**)&m->d_bias, m->h_bias.size() * sizeof(float)));
    if (allocate) {
        NRF_HIP(hipMalloc((void**)&m->d_queues, nrf::kQueueSlots * sizeof(unsigned long long)));
        m->net.queues = m->d_queues;
    }
    NRF_HIP(hipMemcpyAsync(m->d_bias, m->h_bias.data(), m->h_bias.size() * sizeof(float), hipMemcpyHostToDevice, s));
    NRF_HIP(hipStreamSynchronize(s));     // pageable staging: the host vectors may be re-packed right after
    m->net.bias = m->d_bias;
    m->net.n_bias = m->plan.n_bias;
    m->net.flops_per_sample = m->plan.flops_per_sample;
    return NRF_OK;
}

// device-side element sources of the forward streams + bias table (needed by nrf_model_update_device)
int ensure_sources(nrf_model* m) {
    if (m->d_src[0][0]) return NRF_OK;
    m->layout = nrf::param_layout(m->lin);
    for (int kind = 0; kind < 3; ++kind) {
        const std::vector<int32_t> src = nrf::stream_sources(m->plan, m->layout, kind);
        NRF_HIP(hipMalloc((void**)&m->d_src[0][kind], src.size() * sizeof(int32_t)));
        NRF_HIP(hipMemcpy(m->d_src[0][kind], src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        m->n_src[0][kind] = (int64_t)src.size();
    }
    const std::vector<int32_t> bsrc = nrf::bias_sources(m->plan, m->layout);
    NRF_HIP(hipMalloc((void**)&m->d_bias_src, bsrc.size() * sizeof(int32_t)));
    NRF_HIP(hipMemcpy(m->d_bias_src, bsrc.data(), bsrc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return NRF_OK;
}

// backward streams, weight-gradient maps: everything the training kernels need beyond the forward model
int ensure_train(nrf_model* m) {
    if (m->train_ready) return NRF_OK;
    int rc = ensure_sources(m);
    if (rc != NRF_OK) return rc;
    std::string err;
    if (!nrf::make_backward_plan(m->arch, m->lin, m->bplan, err) || !nrf::make_train_plan(m->arch, m->plan, m->layout, m->tplan, err))
        return fail(NRF_EUNSUPPORTED, err);
    if ((int)m->tplan.slot_tiles.size() > nrf::kMaxSlots || (int)m->tplan.jobs.size() > nrf::kMaxJobs || m->tplan.n_mask_slots > nrf::kMaxMaskSlots)
        return fail(NRF_EUNSUPPORTED, "network too deep for the training path");
    for (int mode = 0; mode < 3; ++mode) {
        const nrf::PackedStream ps = nrf::pack_stream(m->bplan, m->lin, mode);
        NRF_HIP(hipMalloc(&m->d_bstream[mode], ps.bytes.size()));
        NRF_HIP(hipMemcpy(m->d_bstream[mode], ps.bytes.data(), ps.bytes.size(), hipMemcpyHostToDevice));
        m->train.bstream[mode] = m->d_bstream[mode];
        m->train.n_bchunks[mode] = ps.n_chunks;
        m->bfresh[mode] = !m->lin_stale;
    }
    for (int f32 = 0; f32 < 2; ++f32) {
        const std::vector<int32_t> src = nrf::stream_sources(m->bplan, m->layout, f32);
        NRF_HIP(hipMalloc((void**)&m->d_src[1][f32], src.size() * sizeof(int32_t)));
        NRF_HIP(hipMemcpy(m->d_src[1][f32], src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        m->n_src[1][f32] = (int64_t)src.size();
    }
    const int nj = (int)m->tplan.jobs.size();
    std::vector<int32_t> maps((size_t)nj * nrf::kMapStride, -1);
    for (int j = 0; j < nj; ++j) {
        const nrf::GradJobPlan& J = m->tplan.jobs[j];
        if (J.row_w.size() > 320 || J.col.size() > 320) return fail(NRF_EUNSUPPORTED, "layer too wide for the weight-gradient maps");
        std::copy(J.row_w.begin(), J.row_w.end(), maps.begin() + (size_t)j * nrf::kMapStride);
        std::copy(J.row_b.begin(), J.row_b.end(), maps.begin() + (size_t)j * nrf::kMapStride + 320);
        std::copy(J.col.begin(), J.col.end(), maps.begin() + (size_t)j * nrf::kMapStride + 640);
        m->train.job_x_slot[j] = J.x_slot; m->train.job_dz_slot[j] = J.dz_slot; m->train.job_KT[j] = J.KT; m->train.job_MT[j] = J.MT; m->train.job_x_first[j] = J.x_first;
    }
    NRF_HIP(hipMalloc((void**)&m->d_maps, maps.size() * sizeof(int32_t)));
    NRF_HIP(hipMemcpy(m->d_maps, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    m->train.maps = m->d_maps;
    m->train.n_jobs = nj;
    m->train.n_slots = (int)m->tplan.slot_tiles.size();
    for (int i = 0; i < m->train.n_slots; ++i) m->train.slot_tiles[i] = m->tplan.slot_tiles[i];
    m->train.n_mask_slots = m->tplan.n_mask_slots;
    m->train.aux_floats = m->tplan.aux_floats;
    m->train.cu_count = m->net.cu_count;
    m->train.n_params = m->layout.total;
    m->train_ready = true;
    return NRF_OK;
}

int check_opts(const nrf_render_opts* o) {
    if (!o) return fail(NRF_EINVAL, "opts is NULL");
    if (o->n_samples < 1 || o->n_samples > 4096) return fail(NRF_EINVAL, "n_samples must be in 1..4096");
    if (!(o->near > 0.0f) || !(o->far > o->near) || !std::isfinite(o->far)) return fail(NRF_EINVAL, "need 0 < near < far");
    if (o->mma_mode < 0 || o->mma_mode >= nrf::kModes) return fail(NRF_EINVAL, "unknown mma_mode");
    if (!(o->ert_eps >= 0.0f) || o->ert_eps >= 1.0f) return fail(NRF_EINVAL, "ert_eps must be in [0,1)");
    return NRF_OK;
}

// out_rgbd rows are written with one 16-byte store per ray: refuse a misaligned base before anything else is looked at (a C caller
// passing a float-aligned sub-view would otherwise get a GPU memory fault instead of an error code)
bool rgbd_misaligned(const nrf_render_opts* o, const float* rgb) {
    return o && o->out_rgbd && rgb && (reinterpret_cast<uintptr_t>(rgb) & 15u) != 0;
}
const char* const kRgbdAlign = "out_rgbd: rgb must be 16-byte aligned ((R,4) rows written with one 16-byte store per ray)";

void fill_common(nrf::RenderArgs& a, const nrf_render_opts* o, float* rgb, float* depth, float* weights, float* z_vals) {
    a.near = o->near; a.far = o->far; a.n_samples = o->n_samples; a.lindisp = o->lindisp; a.perturb = o->perturb;
    a.t_rand = o->perturb ? o->t_rand : nullptr; a.z_ladder = o->z_ladder; a.z_in = o->z_in; a.seed = o->rng_seed;
    a.ert_eps = o->ert_eps; a.white_bkgd = o->white_bkgd;
    a.interleaved = o->out_rgbd ? 1 : 0;
    a.rgb = rgb; a.depth = depth; a.weights = weights; a.z_vals = z_vals;
}

}  // namespace

extern "C" {

int nrf_abi_version(void) { return NRF_ABI_VERSION; }

int nrf_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(nrf_arch);
        case 1: return (int)sizeof(nrf_linear);
        case 2: return (int)sizeof(nrf_dino);
        case 3: return (int)sizeof(nrf_render_opts);
        default: return -1;
    }
}

const char* nrf_last_error(void) { return g_err.c_str(); }

int nrf_model_create(nrf_model** out, int device, const nrf_arch* arch, const nrf_linear* linears, int n_linear) {
    if (!out || !arch || !linears || n_linear <= 0) return fail(NRF_EINVAL, "nrf_model_create: null argument");
    *out = nullptr;
    nrf_model* m = new (std::nothrow) nrf_model();
    if (!m) return fail(NRF_ENOMEM, "out of host memory");
    m->arch = *arch;
    m->device = device;
    std::string err;
    if (!copy_linears(linears, n_linear, m->lin, err) || !nrf::make_plan(*arch, m->lin, m->plan, err)) {
        delete m;
        return fail(NRF_EINVAL, err);
    }
    DeviceGuard guard(device);
    if (!guard.ok) { delete m; return fail(NRF_EHIP, "cannot select device " + std::to_string(device)); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete m; return fail(NRF_EHIP, "hipGetDeviceProperties failed"); }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        delete m;
        return fail(NRF_EUNSUPPORTED, std::string("libnerfhip is built for gfx950 only, device is ") + prop.gcnArchName);
    }
    m->net.arch = *arch;
    m->net.device = device;
    m->net.cu_count = prop.multiProcessorCount;
    const int rc = upload(m, nullptr, true);
    if (rc != NRF_OK) { nrf_model_destroy(m); return rc; }
    *out = m;
    return NRF_OK;
}

int nrf_model_update(nrf_model* m, const nrf_linear* linears, int n_linear, void* stream) {
    if (!m || !linears) return fail(NRF_EINVAL, "nrf_model_update: null argument");
    std::string err;
    std::vector<nrf::HostLinear> lin;
    nrf::NetPlan plan;
    if (!copy_linears(linears, n_linear, lin, err) || !nrf::make_plan(m->arch, lin, plan, err)) return fail(NRF_EINVAL, err);
    m->lin.swap(lin);
    m->plan = plan;
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    return upload(m, (hipStream_t)stream, false);
}

void nrf_model_destroy(nrf_model* m) {
    if (!m) return;
    DeviceGuard guard(m->device);
    for (int i = 0; i < nrf::kModes; ++i)
        if (m->d_stream[i]) (void)hipFree(m->d_stream[i]);
    if (m->d_bias) (void)hipFree(m->d_bias);
    if (m->d_queues) (void)hipFree(m->d_queues);
    for (int i = 0; i < 3; ++i)
        if (m->d_bstream[i]) (void)hipFree(m->d_bstream[i]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            if (m->d_src[i][j]) (void)hipFree(m->d_src[i][j]);
    if (m->d_maps) (void)hipFree(m->d_maps);
    if (m->d_bias_src) (void)hipFree(m->d_bias_src);
    delete m;
}

int64_t nrf_model_flops_per_sample(const nrf_model* m) { return m ? m->plan.flops_per_sample : 0; }

int nrf_render_rays(const nrf_model* m, const float* rays_o, const float* rays_d, int64_t n_rays, const nrf_render_opts* opts,
                    float* rgb, float* depth, float* weights, float* z_vals, void* stream) {
    if (rgbd_misaligned(opts, rgb)) return fail(NRF_EINVAL, kRgbdAlign);
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n_rays < 0) return fail(NRF_EINVAL, "n_rays < 0");
    if (n_rays == 0) return NRF_OK;
    const int rc = check_opts(opts);
    if (rc != NRF_OK) return rc;
    if (!rays_o || !rays_d || !rgb || (!depth && !opts->out_rgbd)) return fail(NRF_EINVAL, "nrf_render_rays: null ray or output pointer");
    if ((int64_t)opts->n_samples * n_rays > (int64_t)1 << 40) return fail(NRF_EINVAL, "ray-sample count too large");
    nrf::RenderArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.camera_mode = 0; a.ray_begin = 0; a.n_rays = n_rays;
    a.n_cams = 1; a.rays_per_cam = n_rays; a.tile_rays = n_rays; a.tile_stride = 0;
    fill_common(a, opts, rgb, depth, weights, z_vals);
    std::string err;
    if (m->arch.net == NRF_NET_V3 && !make_dino(opts->dino, m->arch.dino_dim, a.dino, err)) return fail(NRF_EINVAL, err);
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int r = nrf::launch_render(m->net, opts->mma_mode, a, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_render_camera(const nrf_model* m, int H, int W, float focal, const float c2w[12], int64_t ray_begin, int64_t ray_end,
                      const nrf_render_opts* opts, float* rgb, float* depth, float* weights, float* z_vals, void* stream) {
    if (rgbd_misaligned(opts, rgb)) return fail(NRF_EINVAL, kRgbdAlign);
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (H < 1 || W < 1 || !(focal > 0.0f) || !c2w) return fail(NRF_EINVAL, "bad camera");
    if (ray_begin < 0 || ray_end < ray_begin || ray_end > (int64_t)H * W) return fail(NRF_EINVAL, "ray range outside the image");
    if (ray_end == ray_begin) return NRF_OK;
    const int rc = check_opts(opts);
    if (rc != NRF_OK) return rc;
    if (!rgb || (!depth && !opts->out_rgbd)) return fail(NRF_EINVAL, "nrf_render_camera: null output pointer");
    nrf::RenderArgs a{};
    a.camera_mode = 1; a.n_cams = 1; a.cams[0] = make_camera(H, W, focal, c2w); a.ray_begin = ray_begin; a.n_rays = ray_end - ray_begin;
    a.rays_per_cam = a.n_rays; a.tile_rays = a.n_rays; a.tile_stride = 0;
    fill_common(a, opts, rgb, depth, weights, z_vals);
    std::string err;
    if (m->arch.net == NRF_NET_V3 && !make_dino(opts->dino, m->arch.dino_dim, a.dino, err)) return fail(NRF_EINVAL, err);
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int r = nrf::launch_render(m->net, opts->mma_mode, a, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_render_cameras_tiles(const nrf_model* m, int H, int W, float focal, const float* c2w, int n_cams, int64_t tile_rays,
                             int64_t first_tile, int64_t tile_step, int64_t n_tiles, const nrf_render_opts* opts, float* rgb, float* depth,
                             float* weights, float* z_vals, void* stream) {
    if (rgbd_misaligned(opts, rgb)) return fail(NRF_EINVAL, kRgbdAlign);
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (H < 1 || W < 1 || !(focal > 0.0f) || !c2w) return fail(NRF_EINVAL, "bad camera");
    if (n_cams < 1 || n_cams > nrf::kMaxCams) return fail(NRF_EINVAL, "n_cams must be in 1..8 per call");
    if (tile_rays < 1 || first_tile < 0 || tile_step < 1 || n_tiles < 0) return fail(NRF_EINVAL, "bad tile description");
    if (n_tiles == 0) return NRF_OK;
    if (first_tile * tile_rays >= (int64_t)H * W) return fail(NRF_EINVAL, "first tile lies outside the image");
    const int rc = check_opts(opts);
    if (rc != NRF_OK) return rc;
    if (!rgb || (!depth && !opts->out_rgbd)) return fail(NRF_EINVAL, "nrf_render_cameras_tiles: null output pointer");
    if (opts->t_rand || opts->z_in) return fail(NRF_EINVAL, "tile rendering takes no per-ray inputs (t_rand and z_in must be NULL)");
    nrf::RenderArgs a{};
    a.camera_mode = 1; a.n_cams = n_cams;
    for (int c = 0; c < n_cams; ++c) a.cams[c] = make_camera(H, W, focal, c2w + 12 * c);
    a.ray_begin = first_tile * tile_rays; a.rays_per_cam = n_tiles * tile_rays; a.n_rays = a.rays_per_cam * n_cams;
    a.tile_rays = tile_rays; a.tile_stride = tile_step * tile_rays;
    fill_common(a, opts, rgb, depth, weights, z_vals);
    std::string err;
    if (m->arch.net == NRF_NET_V3 && !make_dino(opts->dino, m->arch.dino_dim, a.dino, err)) return fail(NRF_EINVAL, err);
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int r = nrf::launch_render(m->net, opts->mma_mode, a, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_get_rays(int H, int W, float focal, const float c2w[12], int64_t ray_begin, int64_t ray_end, float* rays_o, float* rays_d,
                 void* stream) {
    if (H < 1 || W < 1 || !(focal > 0.0f) || !c2w) return fail(NRF_EINVAL, "bad camera");
    if (ray_begin < 0 || ray_end < ray_begin || ray_end > (int64_t)H * W) return fail(NRF_EINVAL, "ray range outside the image");
    if (ray_end > ray_begin && (!rays_o || !rays_d)) return fail(NRF_EINVAL, "null output");
    const int r = nrf::launch_get_rays(make_camera(H, W, focal, c2w), ray_begin, ray_end - ray_begin, rays_o, rays_d, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "get_rays launch failed");
}

int nrf_sample_along_rays(const float* rays_o, const float* rays_d, int64_t n_rays, float near, float far, int n_samples, int lindisp,
                          int perturb, const float* t_rand, const float* z_ladder, uint64_t rng_seed, float* pts, float* z_vals,
                          void* stream) {
    if (n_rays < 0 || n_samples < 1) return fail(NRF_EINVAL, "bad sizes");
    if (n_rays == 0) return NRF_OK;
    if (!rays_o || !rays_d || (!pts && !z_vals)) return fail(NRF_EINVAL, "null pointer");
    const int r = nrf::launch_sample(rays_o, rays_d, n_rays, near, far, n_samples, lindisp, perturb, perturb ? t_rand : nullptr, z_ladder, rng_seed,
                                     pts, z_vals, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "sample launch failed");
}

int nrf_encode(const float* x, int64_t n, int dim, int num_freqs, int include_input, const float* freq_bands, float* out, void* stream) {
    if (n < 0 || dim < 1 || num_freqs < 0 || (num_freqs > 31 && !freq_bands)) return fail(NRF_EINVAL, "bad sizes");
    if (n == 0) return NRF_OK;
    if (!x || !out) return fail(NRF_EINVAL, "null pointer");
    const int r = nrf::launch_encode(x, n, dim, num_freqs, include_input, freq_bands, out, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "encode launch failed");
}

int nrf_mlp_forward_v1(const nrf_model* m, int mma_mode, const float* x_enc, int64_t n, float* out4, void* stream) {
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n < 0) return fail(NRF_EINVAL, "n < 0");
    if (n == 0) return NRF_OK;
    if (!x_enc || !out4) return fail(NRF_EINVAL, "null pointer");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    std::string err;
    const int r = nrf::launch_forward_v1(m->net, mma_mode, x_enc, n, out4, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_mlp_forward(const nrf_model* m, int mma_mode, const float* positions, const float* directions, const float* dino, int64_t n,
                    float* rgb, float* density, void* stream) {
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n < 0) return fail(NRF_EINVAL, "n < 0");
    if (n == 0) return NRF_OK;
    if (!positions || !directions || !rgb || !density) return fail(NRF_EINVAL, "null pointer");
    if (m->arch.net == NRF_NET_V3 && !dino) return fail(NRF_EINVAL, "V3 needs dino features");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    std::string err;
    const int r = nrf::launch_forward(m->net, mma_mode, positions, directions, dino, n, rgb, density, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z_vals, const float* rays_d,
                  int64_t n_rays, int n_samples, int white_bkgd, float* out_rgb, float* out_depth, float* out_weights, void* stream) {
    if (n_rays < 0 || n_samples < 1) return fail(NRF_EINVAL, "bad sizes");
    if (rgb_stride < 3 || sigma_stride < 1) return fail(NRF_EINVAL, "bad strides");
    if (n_rays == 0) return NRF_OK;
    if (!rgb || !sigma || !z_vals || !rays_d || !out_rgb) return fail(NRF_EINVAL, "null pointer");
    const int r = nrf::launch_composite(rgb, rgb_stride, sigma, sigma_stride, z_vals, rays_d, n_rays, n_samples, white_bkgd, out_rgb,
                                        out_depth, out_weights, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "composite launch failed");
}

int64_t nrf_param_count(const nrf_model* m) {
    if (!m) return 0;
    int64_t n = 0;
    for (const auto& l : m->lin) n += (int64_t)l.out_f * l.in_f + l.out_f;
    return n;
}

int nrf_model_update_device(nrf_model* m, const float* flat_params, int mode_mask, void* stream) {
    if (!m || !flat_params) return fail(NRF_EINVAL, "nrf_model_update_device: null argument");
    if (mode_mask <= 0 || mode_mask >= (1 << nrf::kModes)) return fail(NRF_EINVAL, "mode_mask must select at least one of the NRF_MMA_* modes");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    int rc = ensure_sources(m);
    if (rc != NRF_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((mode_mask & (mode_mask - 1)) == 0) {
        // the per-step case: one mode -> forward stream, backward stream (once training is set up; the split mode has none)
        // and bias table in one launch
        int mode = 0;
        while (!(mode_mask & (1 << mode))) ++mode;
        const int kind = nrf::stream_kind(mode);
        const bool bwd = m->train_ready && mode < 3;
        const int32_t* src[3] = {m->d_src[0][kind], bwd ? m->d_src[1][kind] : nullptr, m->d_bias_src};
        const int64_t n[3] = {m->n_src[0][kind], bwd ? m->n_src[1][kind] : 0, m->plan.n_bias};
        const int modes[3] = {mode, mode, NRF_MMA_F32};
        void* out[3] = {m->d_stream[mode], bwd ? m->d_bstream[mode] : nullptr, m->d_bias};
        rc = nrf::launch_repack3(flat_params, src, n, modes, out, s);
        if (rc != NRF_OK) return fail(rc, "repack launch failed");
        m->lin_stale = true;
        for (int k = 0; k < 3; ++k) m->bfresh[k] = (k == mode) && m->train_ready;
        return NRF_OK;
    }
    for (int mode = 0; mode < nrf::kModes; ++mode) {
        if (!(mode_mask & (1 << mode))) continue;
        const int kind = nrf::stream_kind(mode);
        rc = nrf::launch_repack(flat_params, m->d_src[0][kind], m->n_src[0][kind], mode, m->d_stream[mode], s);
        if (rc != NRF_OK) return fail(rc, "repack launch failed");
        if (m->train_ready && mode < 3) {
            rc = nrf::launch_repack(flat_params, m->d_src[1][kind], m->n_src[1][kind], mode, m->d_bstream[mode], s);
            if (rc != NRF_OK) return fail(rc, "repack launch failed");
            m->bfresh[mode] = true;
        }
    }
    m->lin_stale = true;
    for (int mode = 0; mode < 3; ++mode)
        if (!(mode_mask & (1 << mode))) m->bfresh[mode] = false;
    rc = nrf::launch_repack(flat_params, m->d_bias_src, m->plan.n_bias, NRF_MMA_F32, m->d_bias, s);
    return rc == NRF_OK ? NRF_OK : fail(rc, "repack launch failed");
}

int64_t nrf_train_context_bytes(nrf_model* m, int mma_mode, int64_t n) {
    if (!m || n < 0 || mma_mode < 0 || mma_mode > 2) { (void)fail(NRF_EINVAL, "nrf_train_context_bytes: bad argument"); return -1; }
    DeviceGuard guard(m->device);
    if (!guard.ok) { (void)fail(NRF_EHIP, "cannot select the model's device"); return -1; }
    if (ensure_train(m) != NRF_OK) return -1;
    return nrf::train_ctx_bytes(m->train, mma_mode, n);
}

int nrf_mlp_forward_train_v1(nrf_model* m, int mma_mode, const float* x_enc, int64_t n, float* out4, void* ctx, int64_t ctx_bytes,
                             void* stream) {
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n < 0 || mma_mode < 0 || mma_mode > 2) return fail(NRF_EINVAL, "bad n / mma_mode (the training path is built for bf16, f16 and f32)");
    if (n == 0) return NRF_OK;
    if (!x_enc || !out4 || !ctx) return fail(NRF_EINVAL, "null pointer");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int rc = ensure_train(m);
    if (rc != NRF_OK) return rc;
    if (ctx_bytes < nrf::train_ctx_bytes(m->train, mma_mode, n)) return fail(NRF_EINVAL, "context buffer smaller than nrf_train_context_bytes");
    std::string err;
    const int r = nrf::launch_train_forward(m->net, m->train, mma_mode, x_enc, n, out4, ctx, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_mlp_backward_v1(nrf_model* m, int mma_mode, const float* out4, const float* g_out4, int64_t n, void* ctx, int64_t ctx_bytes,
                        float* flat_grad, void* stream) {
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n < 0 || mma_mode < 0 || mma_mode > 2) return fail(NRF_EINVAL, "bad n / mma_mode (the training path is built for bf16, f16 and f32)");
    if (n == 0) return NRF_OK;
    if (!out4 || !g_out4 || !ctx || !flat_grad) return fail(NRF_EINVAL, "null pointer");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int rc = ensure_train(m);
    if (rc != NRF_OK) return rc;
    if (ctx_bytes < nrf::train_ctx_bytes(m->train, mma_mode, n)) return fail(NRF_EINVAL, "context buffer smaller than nrf_train_context_bytes");
    if (!m->bfresh[mma_mode])
        return fail(NRF_EINVAL, "backward weights of this mode are older than the parameters: call nrf_model_update_device (with this mode) first");
    std::string err;
    const int r = nrf::launch_train_backward(m->net, m->train, mma_mode, out4, g_out4, n, ctx, flat_grad, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_mlp_forward_train(nrf_model* m, int mma_mode, const float* positions, const float* directions, const float* dino, int64_t n, float* rgb,
                          float* density, void* ctx, int64_t ctx_bytes, void* stream) {
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n < 0 || mma_mode < 0 || mma_mode > 2) return fail(NRF_EINVAL, "bad n / mma_mode (the training path is built for bf16, f16 and f32)");
    if (n == 0) return NRF_OK;
    if (!positions || !directions || !rgb || !density || !ctx) return fail(NRF_EINVAL, "null pointer");
    if (m->arch.net == NRF_NET_V1) return fail(NRF_EINVAL, "V1 models take encoded inputs: nrf_mlp_forward_train_v1");
    if (m->arch.net == NRF_NET_V3 && !dino) return fail(NRF_EINVAL, "V3 needs per-sample dino features");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int rc = ensure_train(m);
    if (rc != NRF_OK) return rc;
    if (ctx_bytes < nrf::train_ctx_bytes(m->train, mma_mode, n)) return fail(NRF_EINVAL, "context buffer smaller than nrf_train_context_bytes");
    std::string err;
    const int r = m->arch.net == NRF_NET_V3
        ? nrf::launch_train_forward_v3(m->net, m->train, mma_mode, positions, directions, dino, n, rgb, density, ctx, (hipStream_t)stream, err)
        : nrf::launch_train_forward_v2(m->net, m->train, mma_mode, positions, directions, n, rgb, density, ctx, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_mlp_backward(nrf_model* m, int mma_mode, const float* rgb, const float* density, const float* g_rgb, const float* g_density, int64_t n,
                     void* ctx, int64_t ctx_bytes, float* flat_grad, void* stream) {
    if (!m) return fail(NRF_EINVAL, "model is NULL");
    if (n < 0 || mma_mode < 0 || mma_mode > 2) return fail(NRF_EINVAL, "bad n / mma_mode (the training path is built for bf16, f16 and f32)");
    if (n == 0) return NRF_OK;
    if (!rgb || !density || !g_rgb || !g_density || !ctx || !flat_grad) return fail(NRF_EINVAL, "null pointer");
    if (m->arch.net == NRF_NET_V1) return fail(NRF_EINVAL, "V1 models: nrf_mlp_backward_v1");
    DeviceGuard guard(m->device);
    if (!guard.ok) return fail(NRF_EHIP, "cannot select the model's device");
    const int rc = ensure_train(m);
    if (rc != NRF_OK) return rc;
    if (ctx_bytes < nrf::train_ctx_bytes(m->train, mma_mode, n)) return fail(NRF_EINVAL, "context buffer smaller than nrf_train_context_bytes");
    if (!m->bfresh[mma_mode])
        return fail(NRF_EINVAL, "backward weights of this mode are older than the parameters: call nrf_model_update_device (with this mode) first");
    std::string err;
    const int r = m->arch.net == NRF_NET_V3
        ? nrf::launch_train_backward_v3(m->net, m->train, mma_mode, rgb, density, g_rgb, g_density, n, ctx, flat_grad, (hipStream_t)stream, err)
        : nrf::launch_train_backward_v2(m->net, m->train, mma_mode, rgb, density, g_rgb, g_density, n, ctx, flat_grad, (hipStream_t)stream, err);
    return r == NRF_OK ? NRF_OK : fail(r, err);
}

int nrf_composite_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z_vals, const float* rays_d,
                           int64_t n_rays, int n_samples, int white_bkgd, const float* g_rgb, const float* g_depth, const float* g_weights,
                           float* d_rgb, int d_rgb_stride, float* d_sigma, int d_sigma_stride, void* stream) {
    if (n_rays < 0 || n_samples < 1 || n_samples > 4096) return fail(NRF_EINVAL, "bad sizes");
    if (rgb_stride < 3 || sigma_stride < 1 || d_rgb_stride < 3 || d_sigma_stride < 1) return fail(NRF_EINVAL, "bad strides");
    if (n_rays == 0) return NRF_OK;
    if (!rgb || !sigma || !z_vals || !rays_d || !d_rgb || !d_sigma) return fail(NRF_EINVAL, "null pointer");
    if (!g_rgb && !g_depth && !g_weights) return fail(NRF_EINVAL, "no incoming gradient");
    const int r = nrf::launch_composite_backward(rgb, rgb_stride, sigma, sigma_stride, z_vals, rays_d, n_rays, n_samples, white_bkgd, g_rgb,
                                                 g_depth, g_weights, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "composite backward launch failed");
}

int nrf_mse_grad(const float* pred, const float* target, int64_t n, float weight, float* g_pred, float* loss, void* stream) {
    if (n <= 0 || n > ((int64_t)1 << 22)) return fail(NRF_EINVAL, "nrf_mse_grad: n must be in 1 .. 2^22");
    if (!pred || !target || !g_pred || !loss) return fail(NRF_EINVAL, "null pointer");
    const int r = nrf::launch_mse_grad(pred, target, n, weight, g_pred, loss, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "mse launch failed");
}

int nrf_composite_mse_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z_vals, const float* rays_d,
                               int64_t n_rays, int n_samples, int white_bkgd, const float* target, float weight, float* pred, float* d_rgb,
                               int d_rgb_stride, float* d_sigma, int d_sigma_stride, float* ray_loss, float* zero_buf, int64_t zero_n,
                               void* stream) {
    if (n_rays <= 0 || n_rays > ((int64_t)1 << 30) || n_samples < 1 || n_samples > 4096) return fail(NRF_EINVAL, "nrf_composite_mse_backward: bad sizes");
    if (rgb_stride < 3 || sigma_stride < 1 || d_rgb_stride < 3 || d_sigma_stride < 1) return fail(NRF_EINVAL, "bad strides");
    if (!rgb || !sigma || !z_vals || !rays_d || !target || !d_rgb || !d_sigma || !ray_loss) return fail(NRF_EINVAL, "null pointer");
    if (zero_n < 0 || (zero_n > 0 && !zero_buf)) return fail(NRF_EINVAL, "zero_buf is NULL");
    const int r = nrf::launch_composite_mse_backward(rgb, rgb_stride, sigma, sigma_stride, z_vals, rays_d, n_rays, n_samples, white_bkgd, target,
                                                     weight, pred, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, ray_loss, zero_buf, zero_n,
                                                     (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "composite + mse + backward launch failed");
}

int nrf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, void* stream) {
    if (n < 0 || step < 1) return fail(NRF_EINVAL, "bad n / step");
    if (n == 0) return NRF_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(NRF_EINVAL, "null pointer");
    if (!(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps >= 0.0f)) return fail(NRF_EINVAL, "bad Adam constants");
    const int r = nrf::launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, 0, 0.0f, nullptr,
                                   (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "adam launch failed");
}

int nrf_adam_step_loss(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, const float* ray_loss, int64_t n_rays, float loss_weight, float* loss,
                       void* stream) {
    if (n <= 0 || step < 1) return fail(NRF_EINVAL, "bad n / step");
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(NRF_EINVAL, "null pointer");
    if (!(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps >= 0.0f)) return fail(NRF_EINVAL, "bad Adam constants");
    if (!ray_loss || !loss || n_rays <= 0) return fail(NRF_EINVAL, "nrf_adam_step_loss: ray_loss / loss missing");
    const int r = nrf::launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, ray_loss, n_rays, loss_weight,
                                   loss, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "adam launch failed");
}

int nrf_sample_pdf(const float* z_vals, const float* weights, int64_t n_rays, int n_samples, int n_importance, const float* u, int64_t u_ray_stride,
                   float* samples, float* z_union, void* stream) {
    if (n_rays < 0 || n_samples < 2 || n_importance < 1) return fail(NRF_EINVAL, "bad sizes");
    if (n_rays == 0) return NRF_OK;
    if (!z_vals || !weights || (!samples && !z_union)) return fail(NRF_EINVAL, "null pointer");
    if (u && u_ray_stride != 0 && u_ray_stride < n_importance) return fail(NRF_EINVAL, "u_ray_stride must be 0 (one shared row) or >= n_importance");
    const int r = nrf::launch_sample_pdf(z_vals, weights, n_rays, n_samples, n_importance, u, u_ray_stride, samples, z_union, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, r == NRF_EINVAL ? "n_samples + n_importance too large for one LDS row" : "sample_pdf launch failed");
}

int nrf_debug_pack(const nrf_arch* arch, const nrf_linear* linears, int n_linear, int mma_mode, uint8_t* stream_out, int64_t stream_cap,
                   int64_t* stream_bytes, float* bias_out, int64_t bias_cap, int64_t* n_bias) {
    if (!arch || !linears || n_linear <= 0) return fail(NRF_EINVAL, "nrf_debug_pack: null argument");
    if (mma_mode < 0 || mma_mode >= nrf::kModes) return fail(NRF_EINVAL, "unknown mma_mode");
    std::string err;
    std::vector<nrf::HostLinear> lin;
    nrf::NetPlan plan;
    if (!copy_linears(linears, n_linear, lin, err) || !nrf::make_plan(*arch, lin, plan, err)) return fail(NRF_EINVAL, err);
    const nrf::PackedStream ps = nrf::pack_stream(plan, lin, mma_mode);
    const std::vector<float> b = nrf::pack_bias(plan, lin);
    if (stream_bytes) *stream_bytes = (int64_t)ps.bytes.size();
    if (n_bias) *n_bias = (int64_t)b.size();
    if (stream_out) {
        if (stream_cap < (int64_t)ps.bytes.size()) return fail(NRF_EINVAL, "stream_out too small");
        std::memcpy(stream_out, ps.bytes.data(), ps.bytes.size());
    }
    if (bias_out) {
        if (bias_cap < (int64_t)b.size()) return fail(NRF_EINVAL, "bias_out too small");
        std::memcpy(bias_out, b.data(), b.size() * sizeof(float));
    }
    return NRF_OK;
}

int nrf_sample_features(const float* features, int Hp, int Wp, int C, const float* points_2d, int64_t n, float* feats, void* stream) {
    if (n < 0 || Hp < 1 || Wp < 1 || C < 1) return fail(NRF_EINVAL, "bad sizes");
    if (n == 0) return NRF_OK;
    if (!features || !points_2d || !feats) return fail(NRF_EINVAL, "null pointer");
    const int r = nrf::launch_sample_features(features, Hp, Wp, C, points_2d, n, feats, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "sample_features launch failed");
}

int nrf_debug_pack_backward(const nrf_arch* arch, const nrf_linear* linears, int n_linear, int mma_mode, uint8_t* stream_out, int64_t stream_cap,
                            int64_t* stream_bytes) {
    if (!arch || !linears || n_linear <= 0) return fail(NRF_EINVAL, "nrf_debug_pack_backward: null argument");
    if (mma_mode < 0 || mma_mode > 2) return fail(NRF_EINVAL, "unknown mma_mode");
    std::string err;
    std::vector<nrf::HostLinear> lin;
    nrf::NetPlan plan, bplan;
    if (!copy_linears(linears, n_linear, lin, err) || !nrf::make_plan(*arch, lin, plan, err)) return fail(NRF_EINVAL, err);
    if (!nrf::make_backward_plan(*arch, lin, bplan, err)) return fail(NRF_EUNSUPPORTED, err);
    const nrf::PackedStream ps = nrf::pack_stream(bplan, lin, mma_mode);
    if (stream_bytes) *stream_bytes = (int64_t)ps.bytes.size();
    if (stream_out) {
        if (stream_cap < (int64_t)ps.bytes.size()) return fail(NRF_EINVAL, "stream_out too small");
        std::memcpy(stream_out, ps.bytes.data(), ps.bytes.size());
    }
    return NRF_OK;
}

int nrf_debug_train_plan(const nrf_arch* arch, const nrf_linear* linears, int n_linear, int32_t* out, int64_t cap, int64_t* n_ints) {
    if (!arch || !linears || n_linear <= 0) return fail(NRF_EINVAL, "nrf_debug_train_plan: null argument");
    std::string err;
    std::vector<nrf::HostLinear> lin;
    nrf::NetPlan plan;
    nrf::TrainPlan tp;
    if (!copy_linears(linears, n_linear, lin, err) || !nrf::make_plan(*arch, lin, plan, err)) return fail(NRF_EINVAL, err);
    if (!nrf::make_train_plan(*arch, plan, nrf::param_layout(lin), tp, err)) return fail(NRF_EUNSUPPORTED, err);
    std::vector<int32_t> v;
    v.push_back((int32_t)tp.slot_tiles.size());
    v.insert(v.end(), tp.slot_tiles.begin(), tp.slot_tiles.end());
    v.push_back((int32_t)tp.jobs.size());
    for (const auto& J : tp.jobs) {
        v.push_back(J.x_slot); v.push_back(J.dz_slot); v.push_back(J.KT); v.push_back(J.MT); v.push_back(J.x_first);
        v.insert(v.end(), J.row_w.begin(), J.row_w.end());
        v.insert(v.end(), J.row_b.begin(), J.row_b.end());
        v.insert(v.end(), J.col.begin(), J.col.end());
    }
    v.push_back(tp.n_mask_slots);
    if (n_ints) *n_ints = (int64_t)v.size();
    if (out) {
        if (cap < (int64_t)v.size()) return fail(NRF_EINVAL, "out too small");
        std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
    }
    return NRF_OK;
}

int nrf_project_fetch(const nrf_dino* dino, const float* points, int64_t n, float* feats, float* xy, void* stream) {
    if (n < 0) return fail(NRF_EINVAL, "n < 0");
    if (n == 0) return NRF_OK;
    if (!points || !feats) return fail(NRF_EINVAL, "null pointer");
    nrf::DinoDev d{};
    std::string err;
    if (!make_dino(dino, 0, d, err)) return fail(NRF_EINVAL, err);
    const int r = nrf::launch_project_fetch(d, points, n, feats, xy, (hipStream_t)stream);
    return r == NRF_OK ? NRF_OK : fail(r, "project_fetch launch failed");
}

}  // extern "C"

// train_v1.hip -- host launchers of the training path (V1, pos_freq 10): train_impl.hpp
#include "train_impl.hpp"

namespace nrf {

namespace {

constexpr int kWgSamples = 256;          // the context is laid out for whole 256-sample groups, whatever the geometry

int64_t tiles32(int64_t n) { return (n + kWgSamples - 1) / kWgSamples * (kWgSamples / 32); }

int tile_bytes_of(int mode) { return mode == NRF_MMA_F32 ? tile_bytes<ModeF32>() : tile_bytes<ModeBF16>(); }

bool fill_slots(const TrainDev& t, int mode, int64_t n, TrainKArgs& k, std::string& err) {
    if (t.n_slots < 1 || t.n_slots > kMaxSlots) { err = "training plan missing"; return false; }
    const int64_t nt = tiles32(n);
    int64_t off = 0;
    for (int i = 0; i < t.n_slots; ++i) {
        k.slot_off[i] = off;
        k.slot_tiles[i] = t.slot_tiles[i];
        off += nt * t.slot_tiles[i] * tile_bytes_of(mode);
    }
    return true;
}

template <class Mode, int WAVES>
int run_train_forward(const DeviceNet& net, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_forward_kernel<Mode, WAVES, 10>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err);
    if (prepared != NRF_OK) return prepared;
    k.net = net_args(net, mode);
    k.net.ablate = 0;
    k.n_tiles = tiles32(k.n) / WAVES;
    const int64_t grid = k.n_tiles < net.cu_count ? k.n_tiles : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytes, s, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("train forward launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

template <class Mode, int WAVES>
int run_train_backward(const DeviceNet& net, const TrainDev& t, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_backward_kernel<Mode, WAVES, 10>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err);
    if (prepared != NRF_OK) return prepared;
    k.net = net_args(net, mode);
    k.net.ablate = 0;
    k.net.stream = t.bstream[mode];
    k.net.n_chunks = t.n_bchunks[mode];
    k.n_tiles = tiles32(k.n) / WAVES;
    const int64_t grid = k.n_tiles < net.cu_count ? k.n_tiles : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytes, s, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("train backward launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

template <class Mode, int ST>
int run_weight_grad(const DeviceNet& net, const TrainDev& t, const TrainKArgs& k, float* grad, hipStream_t s, std::string& err) {
    auto kernel = weight_grad_kernel<Mode, ST>;
    constexpr int kLds = 2 * ST * 16 * tile_bytes<Mode>();
    static_assert(kLds <= 160 * 1024, "weight-gradient staging exceeds the LDS");
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err, kLds);
    if (prepared != NRF_OK) return prepared;
    GradKArgs g{};
    g.ctx = k.ctx; g.grad = grad; g.maps = t.maps; g.n_jobs = t.n_jobs; g.n_tiles32 = tiles32(k.n);
    for (int j = 0; j < t.n_jobs; ++j) {
        g.jobs[j].x_off = k.slot_off[t.job_x_slot[j]];
        g.jobs[j].dz_off = k.slot_off[t.job_dz_slot[j]];
        g.jobs[j].KT = t.job_KT[j]; g.jobs[j].MT = t.job_MT[j];
        g.jobs[j].map_off = j * kMapStride;
    }
    // one workgroup per CU (128 KiB of LDS): about two rounds of workgroups, each with at least a few stages
    int64_t splits = (2 * (int64_t)net.cu_count + t.n_jobs - 1) / t.n_jobs;
    const int64_t max_splits = (g.n_tiles32 + 4 * ST - 1) / (4 * ST);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    g.splits = (int)splits;
    const unsigned grid = (unsigned)(t.n_jobs * splits);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), kLds, s, g);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("weight gradient launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

bool check_train(const DeviceNet& net, const TrainDev& t, int mode, std::string& err) {
    if (mode < 0 || mode > 2) { err = "unknown mma_mode"; return false; }
    if (net.arch.net != NRF_NET_V1 || net.arch.pos_freq != 10) { err = "the training path is built for V1 with pos_freq 10"; return false; }
    if (!t.bstream[mode] || !t.maps) { err = "model not prepared for training"; return false; }
    if (net.n_bias > kBiasMaxFloats) { err = "bias table exceeds the LDS carve-out"; return false; }
    return true;
}

}  // namespace

int64_t train_ctx_bytes(const TrainDev& t, int mode, int64_t n) {
    int64_t tiles = 0;
    for (int i = 0; i < t.n_slots; ++i) tiles += t.slot_tiles[i];
    return tiles32(n) * tiles * tile_bytes_of(mode);
}

int launch_train_forward(const DeviceNet& net, const TrainDev& t, int mode, const float* x_enc, int64_t n, float* out4, void* ctx,
                         hipStream_t s, std::string& err) {
    if (!check_train(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.x_enc = x_enc; k.n = n; k.out4 = out4; k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    switch (mode) {
        case NRF_MMA_BF16: return run_train_forward<ModeBF16, 8>(net, mode, k, s, err);
        case NRF_MMA_F16:  return run_train_forward<ModeF16, 8>(net, mode, k, s, err);
        default:           return run_train_forward<ModeF32, 4>(net, mode, k, s, err);
    }
}

int launch_train_backward(const DeviceNet& net, const TrainDev& t, int mode, const float* out4, const float* g_out4, int64_t n,
                          void* ctx, float* grad, hipStream_t s, std::string& err) {
    if (!check_train(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.n = n; k.out4 = const_cast<float*>(out4); k.g_out4 = g_out4; k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    int r;
    switch (mode) {
        case NRF_MMA_BF16: r = run_train_backward<ModeBF16, 8>(net, t, mode, k, s, err); break;
        case NRF_MMA_F16:  r = run_train_backward<ModeF16, 8>(net, t, mode, k, s, err); break;
        default:           r = run_train_backward<ModeF32, 4>(net, t, mode, k, s, err); break;
    }
    if (r != NRF_OK) return r;
    switch (mode) {
        case NRF_MMA_BF16: return run_weight_grad<ModeBF16, 2>(net, t, k, grad, s, err);
        case NRF_MMA_F16:  return run_weight_grad<ModeF16, 2>(net, t, k, grad, s, err);
        default:           return run_weight_grad<ModeF32, 1>(net, t, k, grad, s, err);
    }
}

int launch_repack(const float* flat, const int32_t* src, int64_t n_elems, int mode, void* out, hipStream_t s) {
    if (n_elems <= 0) return NRF_OK;
    if (mode == NRF_MMA_F32) {
        hipLaunchKernelGGL(repack32_kernel, dim3((unsigned)((n_elems + 255) / 256 < 4096 ? (n_elems + 255) / 256 : 4096)), dim3(256), 0, s, flat, src,
                           n_elems, (float*)out);
    } else {
        const int64_t pairs = n_elems / 2;
        hipLaunchKernelGGL(repack16_kernel, dim3((unsigned)((pairs + 255) / 256 < 4096 ? (pairs + 255) / 256 : 4096)), dim3(256), 0, s, flat, src, pairs,
                           mode == NRF_MMA_BF16 ? 1 : 0, (uint32_t*)out);
    }
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step,
                hipStream_t s) {
    if (n <= 0) return NRF_OK;
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                       wd, bc1, bc2_sqrt);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

}  // namespace nrf

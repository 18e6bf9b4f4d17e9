// train_v2.hip -- host launchers of the V2 training kernels (pos_freq 10, dir_freq 4): train_v2_impl.hpp
#include "train_v2_impl.hpp"

namespace nrf {

namespace {

template <class Mode, int WAVES>
int run_forward(const DeviceNet& net, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_forward_v2_kernel<Mode, WAVES, 10, 4>;
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
int run_backward(const DeviceNet& net, const TrainDev& t, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_backward_v2_kernel<Mode, WAVES, 10>;
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

bool check(const DeviceNet& net, const TrainDev& t, int mode, std::string& err) {
    if (!check_train_common(net, t, mode, err)) return false;
    if (net.arch.net != NRF_NET_V2 || net.arch.dir_freq != 4) { err = "nrf_mlp_forward_train / nrf_mlp_backward need a V2 model with dir_freq 4"; return false; }
    return true;
}

}  // namespace

int launch_train_forward_v2(const DeviceNet& net, const TrainDev& t, int mode, const float* pos, const float* dir, int64_t n, float* rgb,
                            float* density, void* ctx, hipStream_t s, std::string& err) {
    if (!check(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.pos = pos; k.dir = dir; k.n = n; k.rgb = rgb; k.density = density; k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    const bool small = small_batch(net, n);
    switch (mode) {
        case NRF_MMA_BF16: return small ? run_forward<ModeBF16, 4>(net, mode, k, s, err) : run_forward<ModeBF16, 8>(net, mode, k, s, err);
        case NRF_MMA_F16:  return small ? run_forward<ModeF16, 4>(net, mode, k, s, err) : run_forward<ModeF16, 8>(net, mode, k, s, err);
        default:           return run_forward<ModeF32, 4>(net, mode, k, s, err);
    }
}

int launch_train_backward_v2(const DeviceNet& net, const TrainDev& t, int mode, const float* rgb, const float* density,
                             const float* g_rgb, const float* g_density, int64_t n, void* ctx, float* grad, hipStream_t s, std::string& err) {
    if (!check(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.n = n; k.rgb = const_cast<float*>(rgb); k.density = const_cast<float*>(density); k.g_rgb = g_rgb; k.g_density = g_density;
    k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    int r;
    const bool small = small_batch(net, n);
    switch (mode) {
        case NRF_MMA_BF16: r = small ? run_backward<ModeBF16, 4>(net, t, mode, k, s, err) : run_backward<ModeBF16, 8>(net, t, mode, k, s, err); break;
        case NRF_MMA_F16:  r = small ? run_backward<ModeF16, 4>(net, t, mode, k, s, err) : run_backward<ModeF16, 8>(net, t, mode, k, s, err); break;
        default:           r = run_backward<ModeF32, 4>(net, t, mode, k, s, err); break;
    }
    if (r != NRF_OK) return r;
    return launch_weight_grad(net, t, mode, k, grad, s, err);
}

}  // namespace nrf

// train_v3.hip -- host launchers of the V3 (NeRFWithDINO) training kernels (pos_freq 12, dir_freq 4, dino_dim 64 / 128)
#include "train_v3_impl.hpp"

namespace nrf {

namespace {

template <class Mode, int WAVES, int DT>
int run_forward(const DeviceNet& net, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_forward_v3_kernel<Mode, WAVES, 12, 4, DT>;
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

template <class Mode, int WAVES, int DT>
int run_backward(const DeviceNet& net, const TrainDev& t, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_backward_v3_kernel<Mode, WAVES, 12, DT>;
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
    if (net.arch.net != NRF_NET_V3 || net.arch.dir_freq != 4 || (net.arch.dino_dim != 64 && net.arch.dino_dim != 128)) {
        err = "V3 training needs dir_freq 4 and dino_dim 64 or 128";
        return false;
    }
    if (net.arch.n_layers > 8) { err = "V3 training: at most 8 trunk layers (saved-tensor slot table)"; return false; }
    return true;
}

#define NRF_V3_DISPATCH(FN, ...)                                                          \
    const bool small = small_batch(net, k.n);                                             \
    if (net.arch.dino_dim == 64) {                                                        \
        switch (mode) {                                                                   \
            case NRF_MMA_BF16: return small ? FN<ModeBF16, 4, 2>(__VA_ARGS__) : FN<ModeBF16, 8, 2>(__VA_ARGS__);   \
            case NRF_MMA_F16:  return small ? FN<ModeF16, 4, 2>(__VA_ARGS__) : FN<ModeF16, 8, 2>(__VA_ARGS__);     \
            default:           return FN<ModeF32, 4, 2>(__VA_ARGS__);                     \
        }                                                                                 \
    }                                                                                     \
    switch (mode) {                                                                       \
        case NRF_MMA_BF16: return small ? FN<ModeBF16, 4, 4>(__VA_ARGS__) : FN<ModeBF16, 8, 4>(__VA_ARGS__);       \
        case NRF_MMA_F16:  return small ? FN<ModeF16, 4, 4>(__VA_ARGS__) : FN<ModeF16, 8, 4>(__VA_ARGS__);         \
        default:           return FN<ModeF32, 4, 4>(__VA_ARGS__);                         \
    }

int forward_any(const DeviceNet& net, int mode, const TrainKArgs& k, hipStream_t s, std::string& err) { NRF_V3_DISPATCH(run_forward, net, mode, k, s, err) }
int backward_any(const DeviceNet& net, const TrainDev& t, int mode, const TrainKArgs& k, hipStream_t s, std::string& err) { NRF_V3_DISPATCH(run_backward, net, t, mode, k, s, err) }

}  // namespace

int launch_train_forward_v3(const DeviceNet& net, const TrainDev& t, int mode, const float* pos, const float* dir, const float* dino, int64_t n,
                            float* rgb, float* density, void* ctx, hipStream_t s, std::string& err) {
    if (!check(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.pos = pos; k.dir = dir; k.dino = dino; k.n = n; k.rgb = rgb; k.density = density; k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    return forward_any(net, mode, k, s, err);
}

int launch_train_backward_v3(const DeviceNet& net, const TrainDev& t, int mode, const float* rgb, const float* density,
                             const float* g_rgb, const float* g_density, int64_t n, void* ctx, float* grad, hipStream_t s, std::string& err) {
    if (!check(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.n = n; k.rgb = const_cast<float*>(rgb); k.density = const_cast<float*>(density); k.g_rgb = g_rgb; k.g_density = g_density;
    k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    const int r = backward_any(net, t, mode, k, s, err);
    if (r != NRF_OK) return r;
    return launch_weight_grad(net, t, mode, k, grad, s, err);
}

}  // namespace nrf

// fused_kernels.hip -- host-side routing of the fused renderer / staged MLP forward to the per-family translation units
#include "fused_impl.hpp"

namespace nrf {

int launch_render(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) {
    if (!check_net(net, mode, err)) return NRF_EINVAL;
    if (a.n_rays <= 0) return NRF_OK;
    if (net.arch.net == NRF_NET_V1 && net.arch.pos_freq == 10) return render_v1(net, mode, a, s, err);
    if (net.arch.net == NRF_NET_V2 && net.arch.pos_freq == 10) return render_v2(net, mode, a, s, err);
    if (net.arch.net == NRF_NET_V3 && net.arch.pos_freq == 12 && net.arch.dino_dim == 64) return render_v3(net, mode, a, s, err);
    if (net.arch.net == NRF_NET_V3 && net.arch.pos_freq == 12 && net.arch.dino_dim == 128) return render_v3w(net, mode, a, s, err);
    err = "no fused renderer built for this (net, pos_freq): V1/V2 with pos_freq 10 (baseline.yaml) and V3 with pos_freq 12 and dino_dim 64/128 (dino_nerf.yaml, lora.yaml, multiscale.yaml) are";
    return NRF_EUNSUPPORTED;
}

int launch_forward_v1(const DeviceNet& net, int mode, const float* x_enc, int64_t n, float* out4, hipStream_t s, std::string& err) {
    if (!check_net(net, mode, err)) return NRF_EINVAL;
    if (net.arch.net != NRF_NET_V1) { err = "nrf_mlp_forward_v1 needs a V1 model"; return NRF_EINVAL; }
    if (n <= 0) return NRF_OK;
    ForwardKArgs k{};
    k.x_enc = x_enc; k.n = n; k.out4 = out4;
    if (net.arch.pos_freq == 10) return forward_v1(net, mode, k, s, err);
    err = "no V1 forward built for this pos_freq";
    return NRF_EUNSUPPORTED;
}

int launch_forward(const DeviceNet& net, int mode, const float* pos, const float* dir, const float* dino, int64_t n,
                   float* rgb, float* density, hipStream_t s, std::string& err) {
    if (!check_net(net, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    ForwardKArgs k{};
    k.pos = pos; k.dir = dir; k.dino = dino; k.n = n; k.rgb = rgb; k.density = density;
    if (net.arch.net == NRF_NET_V2 && net.arch.pos_freq == 10) return forward_v2(net, mode, k, s, err);
    if (net.arch.net == NRF_NET_V3 && net.arch.pos_freq == 12) {
        if (!dino) { err = "V3 forward needs per-sample dino features"; return NRF_EINVAL; }
        if (net.arch.dino_dim == 64) return forward_v3(net, mode, k, s, err);
        if (net.arch.dino_dim == 128) return forward_v3w(net, mode, k, s, err);
    }
    err = "no forward built for this (net, pos_freq)";
    return NRF_EUNSUPPORTED;
}

}  // namespace nrf

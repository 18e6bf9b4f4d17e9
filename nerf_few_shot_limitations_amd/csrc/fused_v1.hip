// fused_v1.hip -- V1 (nerf_model.NeRFMLP, pos_freq 10) instantiations of the fused renderer / staged forward
#include "fused_impl.hpp"

namespace nrf {

int NRF_TU_NAME(render_v1)(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE(run_render, NetV1, 10, net, mode, a, s, err) }
int NRF_TU_NAME(forward_v1)(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE(run_forward, NetV1, 10, net, mode, k, s, err) }

}  // namespace nrf

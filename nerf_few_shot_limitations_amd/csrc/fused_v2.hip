// fused_v2.hip -- V2 (DensityMLP + ColorMLP, pos_freq 10) instantiations
#include "fused_impl.hpp"

namespace nrf {

int NRF_TU_NAME(render_v2)(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_render, NRF_NET_V2_10, 10, 2, 4, net, mode, a, s, err) }
int NRF_TU_NAME(forward_v2)(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_forward, NRF_NET_V2_10, 10, 2, 4, net, mode, k, s, err) }

}  // namespace nrf

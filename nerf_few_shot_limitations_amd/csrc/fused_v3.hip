// fused_v3.hip -- V3 (NeRFWithDINO, pos_freq 12, 64-d features) instantiations
#include "fused_impl.hpp"

namespace nrf {

int NRF_TU_NAME(render_v3)(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_render, NRF_NET_V3_12_64, 12, 1, 8, net, mode, a, s, err) }
int NRF_TU_NAME(forward_v3)(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_forward, NRF_NET_V3_12_64, 12, 1, 8, net, mode, k, s, err) }

}  // namespace nrf

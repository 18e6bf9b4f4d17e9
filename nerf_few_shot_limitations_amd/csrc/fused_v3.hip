// fused_v3.hip -- V3 (NeRFWithDINO, pos_freq 12, 64-d features) instantiations
#include "fused_impl.hpp"

#ifndef NRF_V3_NT
#define NRF_V3_NT 2          // 16-bit modes: 4 waves x 64 columns (fused_impl.hpp, "Workgroup geometry")
#define NRF_V3_WAVES 4
#endif

namespace nrf {

int NRF_TU_NAME(render_v3)(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_render, NRF_NET_V3_12_64, 12, NRF_V3_NT, NRF_V3_WAVES, net, mode, a, s, err) }
int NRF_TU_NAME(forward_v3)(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_forward, NRF_NET_V3_12_64, 12, NRF_V3_NT, NRF_V3_WAVES, net, mode, k, s, err) }

}  // namespace nrf

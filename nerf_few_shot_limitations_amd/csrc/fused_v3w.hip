// fused_v3w.hip -- V3 with 128-d (multi-scale) features
#include "fused_impl.hpp"

#ifndef NRF_V3_NT
#define NRF_V3_NT 2          // 16-bit modes: 4 waves x 64 columns (fused_impl.hpp, "Workgroup geometry")
#define NRF_V3_WAVES 4
#endif

namespace nrf {

int NRF_TU_NAME(render_v3w)(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_render, NRF_NET_V3_12_128, 12, NRF_V3_NT, NRF_V3_WAVES, net, mode, a, s, err) }
int NRF_TU_NAME(forward_v3w)(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) { NRF_DISPATCH_MODE1(run_forward, NRF_NET_V3_12_128, 12, NRF_V3_NT, NRF_V3_WAVES, net, mode, k, s, err) }

}  // namespace nrf

// packing.hpp -- host side: turn nn.Linear weights into the fragment stream the
// MFMA kernels consume (see mlp_core.hpp for the layout contract).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nerfhip.h"

namespace nrf {

struct HostLinear {
    std::vector<float> w, b;   // (out_f,in_f) row-major, (out_f)
    int out_f = 0, in_f = 0;
};

// One packed layer = MT output tiles x KT input tiles of 32.
struct LayerPlan {
    int KT = 0, MT = 0;
    std::vector<int> col;                        // 32*KT: source column of the Linear(s), -1 = zero
    std::vector<std::pair<int, int>> row;        // 32*MT: (linear index, row of it), (-1,0) = zero row
    int bias_off = 0;                            // offset of this layer's 32*MT biases in the bias table
};

struct NetPlan {
    std::vector<LayerPlan> layers;               // in the order the kernel walks them
    int n_bias = 0;                              // floats in the bias table
    int64_t flops_per_sample = 0;                // 2*MAC of the reference's Linear layers
};

// Validates `arch` against the Linear list and lays the network out.  Returns false + err.
bool make_plan(const nrf_arch& arch, const std::vector<HostLinear>& lin, NetPlan& plan, std::string& err);

// Number of Linear layers a state_dict of `arch` must hold (0 = unknown arch).
int expected_linears(const nrf_arch& arch);

struct PackedStream {
    std::vector<uint8_t> bytes;                  // n_chunks * 16 KiB
    uint32_t n_chunks = 0;
};
PackedStream pack_stream(const NetPlan& plan, const std::vector<HostLinear>& lin, int mma_mode);
std::vector<float> pack_bias(const NetPlan& plan, const std::vector<HostLinear>& lin);

uint16_t f32_to_bf16(float x);
uint16_t f32_to_f16(float x);

}  // namespace nrf

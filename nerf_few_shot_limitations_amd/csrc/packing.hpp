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
    // backward-chain layer (dX = W^T dZ): element (r, k) = lin[krow[k].first].w[krow[k].second][rcol[r]]
    bool transposed = false;
    std::vector<std::pair<int, int>> krow;       // 32*KT: (linear index, row of it) feeding K index k, (-1,0) = zero
    std::vector<int> rcol;                       // 32*MT: column of those Linears produced by output row r, -1 = zero
};

// The flat parameter vector of the training path: the Linears in list (state_dict) order, each as
// weight (out_f*in_f, row-major) followed by bias (out_f).
struct ParamLayout {
    std::vector<int64_t> w_off, b_off;
    std::vector<int> in_f;
    int64_t total = 0;
};
ParamLayout param_layout(const std::vector<HostLinear>& lin);

struct NetPlan {
    std::vector<LayerPlan> layers;               // in the order the kernel walks them
    int n_bias = 0;                              // floats in the bias table
    int64_t flops_per_sample = 0;                // 2*MAC of the reference's Linear layers
};

// Validates `arch` against the Linear list and lays the network out.  Returns false + err.
bool make_plan(const nrf_arch& arch, const std::vector<HostLinear>& lin, NetPlan& plan, std::string& err);

// Number of Linear layers a state_dict of `arch` must hold (0 = unknown arch).
int expected_linears(const nrf_arch& arch);

// The backward chain of a network (train_impl.hpp walks the layers in this order); V1 and V2 are built.
bool make_backward_plan(const nrf_arch& arch, const std::vector<HostLinear>& lin, NetPlan& plan, std::string& err);

// Where every element of a packed stream comes from: flat-parameter offset, -1 = zero.  Elements in stream
// order (fragment, lane, element); 512 per fragment in the 16-bit modes, 256 in the fp32 mode.  The host packer
// and the device re-packer (after every optimizer step) are both a gather through this table.
// kind: kStream16 (bf16 / f16: 2 fragments of 8 elements per tile pair), kStreamF32 (4 fragments of 4 fp32), kStreamX3 (the split
// f16 mode: 4 fragments of 8 -- hi and lo parts of the two 16-bit fragments, interleaved).
enum { kStream16 = 0, kStreamF32 = 1, kStreamX3 = 2 };
inline int stream_kind(int mma_mode) { return mma_mode == NRF_MMA_F32 ? kStreamF32 : (mma_mode == NRF_MMA_F16X3 ? kStreamX3 : kStream16); }
std::vector<int32_t> stream_sources(const NetPlan& plan, const ParamLayout& lay, int kind);
std::vector<int32_t> bias_sources(const NetPlan& plan, const ParamLayout& lay);

// One weight-gradient job = one Linear (or the head pseudo-layer): dW[o][i] += sum_samples dZ[o] * X[i].
// Saved-tensor slots: train_impl.hpp.
struct GradJobPlan {
    int x_slot = 0, dz_slot = 0, KT = 0, MT = 0, x_first = 0;
    std::vector<int32_t> row_w, row_b;           // 32*MT: flat offset of the weight row / of the bias, -1 = none
    std::vector<int32_t> col;                    // 32*KT: column inside the weight row, -1 = none
};
struct TrainPlan {
    std::vector<int> slot_tiles;                 // feature tiles (of 32) per saved-tensor slot
    int n_mask_slots = 0;                        // ReLU-mask bit planes, one per layer followed by a ReLU (except the heads)
    int aux_floats = 0;                          // extra fp32 values saved per sample (V3: the softmax gate)
    std::vector<GradJobPlan> jobs;
};
bool make_train_plan(const nrf_arch& arch, const NetPlan& fwd, const ParamLayout& lay, TrainPlan& tp, std::string& err);

struct PackedStream {
    std::vector<uint8_t> bytes;                  // n_chunks * 16 KiB
    uint32_t n_chunks = 0;
};
PackedStream pack_stream(const NetPlan& plan, const std::vector<HostLinear>& lin, int mma_mode);
std::vector<float> pack_bias(const NetPlan& plan, const std::vector<HostLinear>& lin);

uint16_t f32_to_bf16(float x);
uint16_t f32_to_f16(float x);
float f16_to_f32(uint16_t h);

}  // namespace nrf

// feature_map.hpp -- where each positional-encoding feature of the reference
// lives inside the kernel's operand tiles.  Shared by the host packer (to
// permute the first Linear's columns) and the device encoder.
//
// Reference order (positional_encoding.py:27-33 == nerf_mlp.py:24-33), input
// dim 3, L frequencies, include_input: index 0..2 = x,y,z; 3+6f+c = sin(2^f x_c);
// 3+6f+3+c = cos(2^f x_c).
//
// Kernel order: operand tile t (32 K-rows), register r (0..15) of lane half h
// holds K index  32t + (r&3) + 8(r>>2) + 4h  (the MFMA accumulator row map), and
// we give lane half 0 the sines and lane half 1 the cosines of the same
// (frequency, coordinate) so both halves run one instruction stream:
//   slot u = 16t + r:  u < 3L     -> f = u/3, c = u%3: sin (h=0) / cos (h=1)
//                      u == 3L    -> x (h=0) / z (h=1)
//                      u == 3L+1  -> y (h=0) / unused (h=1)
#pragma once

namespace nrf {

constexpr __host__ __device__ int pe_tiles(int L) { return (3 * L + 2 + 15) / 16; }
constexpr __host__ __device__ int pe_dim(int L) { return 3 * (2 * L + 1); }

// reference feature index held by (slot u, half h), or -1 for padding
constexpr __host__ __device__ int pe_ref_index(int L, int u, int h) {
    if (u < 3 * L) return 3 + 6 * (u / 3) + 3 * h + (u % 3);
    if (u == 3 * L) return h ? 2 : 0;
    if (u == 3 * L + 1) return h ? -1 : 1;
    return -1;
}

// K index (within a layer's padded input) -> (tile, half, reg, slot)
constexpr __host__ __device__ int k_tile(int k) { return k >> 5; }
constexpr __host__ __device__ int k_half(int k) { return (k >> 2) & 1; }
constexpr __host__ __device__ int k_reg(int k) { return (k & 3) | (((k & 31) >> 3) << 2); }
constexpr __host__ __device__ int k_slot(int k) { return 16 * k_tile(k) + k_reg(k); }
// and back: (tile-local slot u, half h) -> K index
constexpr __host__ __device__ int slot_k(int u, int h) {
    const int t = u >> 4, r = u & 15;
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

}  // namespace nrf

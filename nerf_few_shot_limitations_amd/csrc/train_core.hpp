// train_core.hpp -- device helpers of the training path: the activation store,
// the ReLU-mask epilogue of the backward chain and the MFMA transpose the weight-gradient kernel is built on.
//
// Saved tensors ("context").  Every operand tile of the forward chain (32 features x 32 samples of one wave,
// mlp_core.hpp) is written to HBM exactly as the lanes hold it: sizeof(Act)/16 vectors of 64 lanes x 16 B, each
// vector one coalesced 1-KiB store.  Tile (sample tile st, feature tile t) of a slot with KT tiles sits at
//     slot_base + ((st * KT + t) * kVecs) KiB.
// The backward chain writes its masked gradients dZ in the same format.  Features live in the lane's
// registers and samples on the lanes, which is the wrong way round for dW = dZ * X^T (a contraction over
// SAMPLES), so the weight-gradient kernel first transposes each tile on the matrix core:
//     T = X^T = A(X) * E        A = the stored tile used as the MFMA A operand (lane <-> sample row),
//                               E = a 0/1 selection matrix that undoes the K permutation of the tile
// (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand": as the A operand a tile
// computes X^T * B).  T has the feature on the lane and 16 samples in the registers; two such tiles, of dZ and
// of X, ARE the A and B operands of dW += dZ_tile * X_tile^T with the same (permuted) sample order in both.
#pragma once
#include "mlp_core.hpp"

namespace nrf {

typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

template <class Mode> struct ActIO;

template <class V8>
struct ActIO16 {
    static constexpr int kVecs = 2;
    template <class Act>
    __device__ static __forceinline__ void store(char* p, const Act& a) {
#pragma unroll
        for (int v = 0; v < 2; ++v) *(i32x4*)(p + v * kFragBytes) = __builtin_bit_cast(i32x4, a.f[v]);
    }
    template <class Act>
    __device__ static __forceinline__ Act load(const char* p) {
        Act a;
#pragma unroll
        for (int v = 0; v < 2; ++v) a.f[v] = __builtin_bit_cast(V8, *(const i32x4*)(p + v * kFragBytes));
        return a;
    }
};
template <> struct ActIO<ModeBF16> : ActIO16<bf16x8> {};
template <> struct ActIO<ModeF16> : ActIO16<f16x8> {};
template <>
struct ActIO<ModeF32> {
    static constexpr int kVecs = 4;
    template <class Act>
    __device__ static __forceinline__ void store(char* p, const Act& a) {
#pragma unroll
        for (int v = 0; v < 4; ++v) *(f32x4*)(p + v * kFragBytes) = f32x4{a.r[4 * v], a.r[4 * v + 1], a.r[4 * v + 2], a.r[4 * v + 3]};
    }
    template <class Act>
    __device__ static __forceinline__ Act load(const char* p) {
        Act a;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f32x4 q = *(const f32x4*)(p + v * kFragBytes);
#pragma unroll
            for (int e = 0; e < 4; ++e) a.r[4 * v + e] = q[e];
        }
        return a;
    }
};

template <class Mode>
constexpr __host__ __device__ int tile_bytes() { return ActIO<Mode>::kVecs * kFragBytes; }

// dZ = dH where the forward activation was positive, else 0 (ReLU'), converted to the operand type.
// x holds relu(.) >= 0 in the operand type: "positive" == "non-zero bit pattern".
// 0xffff in every 16-bit half of x that is non-zero (x = two non-negative 16-bit floats): min(x, 1) is 0 / 1 per half,
// 0 - that is 0 / 0xffff.  Inline asm: the packed-integer idiom written with clang's vector builtins was folded into
// ONE mask for all four dwords of a fragment (seen in the ISA and in the parity test).
__device__ __forceinline__ int nonzero_halves(int x) {
    int m;
    const int ones = 0x00010001;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(x), "v"(ones));
    asm("v_pk_sub_u16 %0, 0, %1" : "=v"(m) : "v"(m));
    return m;
}

template <class Mode> struct Masked;
template <class V8, class Mode16>
struct Masked16 {
    typedef typename Mode16::Act Act;
    __device__ static __forceinline__ Act apply(const f32x16& v, const Act& x) {
        Act o = Mode16::template to_act<false>(v);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            i32x4 q = __builtin_bit_cast(i32x4, o.f[s]);
            const i32x4 xi = __builtin_bit_cast(i32x4, x.f[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] &= nonzero_halves(xi[j]);
            o.f[s] = __builtin_bit_cast(V8, q);
        }
        return o;
    }
};
template <> struct Masked<ModeBF16> : Masked16<bf16x8, ModeBF16> {};
template <> struct Masked<ModeF16> : Masked16<f16x8, ModeF16> {};
template <>
struct Masked<ModeF32> {
    typedef ModeF32::Act Act;
    __device__ static __forceinline__ Act apply(const f32x16& v, const Act& x) {
        Act o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o.r[r] = x.r[r] > 0.0f ? v[r] : 0.0f;
        return o;
    }
};

// T = X^T through the matrix core (header comment).  Result: lane (c, h) register r = X[feature c][sample row(r, h)].
template <class Mode> struct Transposer;
template <class V8, class E1>
struct Transposer16 {
    V8 E[2];
    __device__ __forceinline__ void init(int lane) {
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) E[s][j] = (c == 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) ? (E1)1.0f : (E1)0.0f;
    }
};
template <>
struct Transposer<ModeBF16> : Transposer16<bf16x8, __bf16> {
    __device__ __forceinline__ f32x16 run(const ModeBF16::Act& x) const {
        f32x16 t = {};
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.f[0], E[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.f[1], E[1], t, 0, 0, 0);
        return t;
    }
};
template <>
struct Transposer<ModeF16> : Transposer16<f16x8, _Float16> {
    __device__ __forceinline__ f32x16 run(const ModeF16::Act& x) const {
        f32x16 t = {};
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.f[0], E[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.f[1], E[1], t, 0, 0, 0);
        return t;
    }
};
template <>
struct Transposer<ModeF32> {
    int c, h;
    __device__ __forceinline__ void init(int lane) { c = lane & 31; h = lane >> 5; }
    __device__ __forceinline__ f32x16 run(const ModeF32::Act& x) const {
        f32x16 t = {};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float e = (c == (q & 3) + 8 * (q >> 2) + 4 * h) ? 1.0f : 0.0f;
            t = __builtin_amdgcn_mfma_f32_32x32x2f32(x.r[q], e, t, 0, 0, 0);
        }
        return t;
    }
};

// acc[o][i] += sum over the 32 samples of Ta[o][n] * Tb[i][n], both operands transposed tiles in operand type
template <class Mode> struct OuterMma;
template <>
struct OuterMma<ModeBF16> {
    __device__ static __forceinline__ void run(f32x16& acc, const ModeBF16::Act& a, const ModeBF16::Act& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f[0], b.f[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f[1], b.f[1], acc, 0, 0, 0);
    }
};
template <>
struct OuterMma<ModeF16> {
    __device__ static __forceinline__ void run(f32x16& acc, const ModeF16::Act& a, const ModeF16::Act& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.f[0], b.f[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.f[1], b.f[1], acc, 0, 0, 0);
    }
};
template <>
struct OuterMma<ModeF32> {
    __device__ static __forceinline__ void run(f32x16& acc, const ModeF32::Act& a, const ModeF32::Act& b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.r[q], b.r[q], acc, 0, 0, 0);
    }
};

}  // namespace nrf

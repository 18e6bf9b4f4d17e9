// train_core.hpp -- device helpers of the training path: the activation store,
// the ReLU-mask bits of the backward chain and the MFMA transpose the weight-gradient kernel is built on.
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

template <class Mode> struct ActIO;

template <class V8>
struct ActIO16 {
    static constexpr int kVecs = 2;
    template <class Act>
    __device__ static __forceinline__ void store(char* p, const Act& a) {
#pragma unroll
        for (int v = 0; v < 2; ++v) *(i32x4*)(p + v * kFragBytes) = __builtin_bit_cast(i32x4, a.f[v]);
    }
    // the saved tiles in HBM are written once and read once or twice by LATER kernels, hundreds of MB per step: streaming
    // (non-temporal) accesses keep them from being allocated in the L2 / MALL on their way (same-box A/B, V1 bf16, 2048 x 32:
    // stores -7.7 % of the step, loads another -3 %; profiles/r03_ab_train_nontemporal.txt)
    template <class Act>
    __device__ static __forceinline__ void store_g(char* p, const Act& a) {
#ifdef NRF_TRAIN_NO_STORE          // timing experiment only (results are wrong): what the chain kernels cost without their stores
        if (p == nullptr)
#endif
#pragma unroll
        for (int v = 0; v < 2; ++v) __builtin_nontemporal_store(__builtin_bit_cast(i32x4, a.f[v]), (i32x4*)(p + v * kFragBytes));
    }
    template <class Act>
    __device__ static __forceinline__ Act load(const char* p) {
        Act a;
#pragma unroll
        for (int v = 0; v < 2; ++v) a.f[v] = __builtin_bit_cast(V8, *(const i32x4*)(p + v * kFragBytes));
        return a;
    }
    template <class Act>
    __device__ static __forceinline__ Act load_g(const char* p) {
        Act a;
#pragma unroll
        for (int v = 0; v < 2; ++v) a.f[v] = __builtin_bit_cast(V8, __builtin_nontemporal_load((const i32x4*)(p + v * kFragBytes)));
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
    __device__ static __forceinline__ void store_g(char* p, const Act& a) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
            __builtin_nontemporal_store(f32x4{a.r[4 * v], a.r[4 * v + 1], a.r[4 * v + 2], a.r[4 * v + 3]}, (f32x4*)(p + v * kFragBytes));
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
    template <class Act>
    __device__ static __forceinline__ Act load_g(const char* p) {
        Act a;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f32x4 q = __builtin_nontemporal_load((const f32x4*)(p + v * kFragBytes));
#pragma unroll
            for (int e = 0; e < 4; ++e) a.r[4 * v + e] = q[e];
        }
        return a;
    }
};

template <class Mode>
constexpr __host__ __device__ int tile_bytes() { return ActIO<Mode>::kVecs * kFragBytes; }

// ReLU' travels from the forward to the backward chain as BITS, not as the activation tiles themselves (which the
// backward chain would otherwise re-read: 2 KiB per tile against 128 B): per lane and layer one 16-byte word group, tile m
// in half (m & 1) of dword (m >> 1), accumulator register r at bit 15 - r of that half.
//   forward : bit = (pre-activation > 0), gathered with one subtract + one v_alignbit per register;
//   backward: dZ[r] = dH[r] AND (0 - bit) before the conversion to the operand type.
constexpr int kMaskBytes = kFragBytes;       // per sample tile and masked layer: 64 lanes x 16 B

__device__ __forceinline__ uint32_t relu_bits(const f32x16& v) {
    uint32_t bits = 0;
    float t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float f = v[r];
        // t = 0 - f: sign(t) = 1 exactly when f > 0 (0 - 0 = +0);  bits = (bits << 1) | sign(t).
        // One asm statement per register keeps t a single short-lived scratch register (written as plain C++ the
        // compiler computed all 16 differences first and spilled: the saving forward sits at the 256-register limit).
        asm volatile("v_sub_f32 %1, 0, %2\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(bits), "=&v"(t) : "v"(f));
    }
    return bits & 0xffffu;
}

template <int M>
__device__ __forceinline__ void put_bits(i32x4& w, uint32_t bits16) {
    if constexpr ((M & 1) == 0) w[M >> 1] = (int)bits16;           // the even tile of a dword comes first and initialises it
    else w[M >> 1] |= (int)(bits16 << 16);
}

template <class Mode, int M>
__device__ __forceinline__ typename Mode::Act masked_act(const f32x16& v, const i32x4& w) {
    const uint32_t word = (uint32_t)w[M >> 1];
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        constexpr int kBase = 16 * (M & 1) + 15;
        const int keep = ((int)(word << (31 - (kBase - r)))) >> 31;       // 0 or -1: sign-extended bit
        const float f = v[r];     // a named scalar: __builtin_bit_cast applied directly to the vector-element expression read element 0
        o[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, f) & keep);
    }
    return Mode::template to_act<false>(o);
}

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

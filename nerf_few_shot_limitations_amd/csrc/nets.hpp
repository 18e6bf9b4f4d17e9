// nets.hpp -- the encoders and the per-family layer walks on top of mlp_core.hpp.
// The order of dense() calls here IS the order of layers in packing.cpp's plan.
#pragma once
#include "device_math.hpp"
#include "feature_map.hpp"
#include "mlp_core.hpp"

namespace nrf {

// 1/(2 pi) split in two floats for an exact-ish turn count: rev = x*C_HI (+ error term)
constexpr float kInv2PiHi = 0x1.45f306p-3f;     // fl32(1/(2 pi))
constexpr float kInv2PiLo = 0x1.b9391p-28f;     // 1/(2 pi) - kInv2PiHi

__device__ __attribute__((noinline)) float precise_sin_or_cos(float arg, int want_cos) {
    float s, c;
    sincosf(arg, &s, &c);
    return want_cos ? c : s;
}

// Positional encoding of one 3-vector into KT operand tiles, for lane half h
// (feature_map.hpp: half 0 = sines + x,y; half 1 = cosines + z).
// positional_encoding.py:27-33: sin/cos of x * 2^f, arguments exact (power-of-two scaling).
template <class Mode, int L>
__device__ __forceinline__ void encode3(const float p[3], int h, typename Mode::Act (&out)[pe_tiles(L)]) {
    constexpr int KT = pe_tiles(L);
    f32x16 e[KT];
    float hi[3], lo[3];
    if constexpr (Mode::FAST_TRIG) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[c] = p[c] * kInv2PiHi;
            lo[c] = __builtin_fmaf(p[c], kInv2PiHi, -hi[c]) + p[c] * kInv2PiLo;
        }
    }
    const float quarter = h ? 0.25f : 0.0f;
    static_for<16 * KT>([&](auto u_) {
        constexpr int u = decltype(u_)::value;
        constexpr int t = u / 16, r = u % 16;
        float v = 0.0f;
        if constexpr (u < 3 * L) {
            constexpr int f = u / 3, c = u % 3;
            constexpr float scale = (float)(1u << f);
            if constexpr (Mode::FAST_TRIG) {
                // turns: fract() of the exact scaled high part + scaled low part; cos = sin a quarter turn on
                const float rev = __builtin_fmaf(lo[c], scale, __builtin_amdgcn_fractf(hi[c] * scale)) + quarter;
                v = __builtin_amdgcn_sinf(rev);
            } else {
                v = precise_sin_or_cos(p[c] * scale, h);
            }
        } else if constexpr (u == 3 * L) {
            v = h ? p[2] : p[0];
        } else if constexpr (u == 3 * L + 1) {
            v = h ? 0.0f : p[1];
        }
        e[t][r] = v;
    });
#pragma unroll
    for (int t = 0; t < KT; ++t) out[t] = Mode::template to_act<false>(e[t]);
}

// ---------------------------------------------------------------------------
// V1: nerf_model.py:16-24   PE -> n x (Linear+ReLU) -> [rgb_out | sigma_out]
// out4[n] = {rgb logits (pre-sigmoid) x3, raw sigma} of sample column (lane&31) of tile n
// ---------------------------------------------------------------------------
template <class Mode, int NT, int LP>
struct NetV1 {
    static constexpr int KT0 = pe_tiles(LP);
    static constexpr int HT = 8;
    static constexpr bool kNeedsDir = false;
    typedef typename Mode::Act Act;

    template <class P>
    __device__ static __forceinline__ void eval(P& pipe, const NRF_LDS float* bias, int h, int n_layers,
                                                const Act (&enc)[KT0][NT], const Act (&)[1][NT], float (&out4)[NT][4]) {
        Act A[HT][NT], B[HT][NT];
        dense_act<Mode, KT0, HT, NT, true>(pipe, bias, h, enc, A);
        int boff = 32 * HT;
        const int hidden = n_layers - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            dense_act<Mode, HT, HT, NT, true>(pipe, bias + boff, h, A, B); boff += 32 * HT;
            dense_act<Mode, HT, HT, NT, true>(pipe, bias + boff, h, B, A); boff += 32 * HT;
        }
        f32x16 head[NT];
        if (hidden & 1) {
            dense_act<Mode, HT, HT, NT, true>(pipe, bias + boff, h, A, B); boff += 32 * HT;
            dense_head<Mode, HT, NT>(pipe, bias + boff, h, B, head);
        } else {
            dense_head<Mode, HT, NT>(pipe, bias + boff, h, A, head);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int k = 0; k < 4; ++k) out4[n][k] = head[n][k];
    }
};

// ---------------------------------------------------------------------------
// V2: nerf_mlp.py:60-66,82-84   PE(pos) -> DensityMLP -> ColorMLP(cat[feature, PE(dir)])
// ---------------------------------------------------------------------------
template <class Mode, int NT, int LP>
struct NetV2 {
    static constexpr int KT0 = pe_tiles(LP);
    static constexpr int HT = 8;
    static constexpr bool kNeedsDir = true;
    typedef typename Mode::Act Act;

    // density_head, feature_head, colour layers; X = trunk output, Y = scratch of the same shape
    template <class P>
    __device__ static __forceinline__ void tail(P& pipe, const NRF_LDS float* bias, int h, const Act (&X)[HT][NT],
                                                Act (&Y)[HT][NT], const Act (&dir)[1][NT], float (&out4)[NT][4]) {
        f32x16 dens[NT];
        dense_head<Mode, HT, NT>(pipe, bias, h, X, dens);
        dense_act<Mode, HT, HT, NT, false>(pipe, bias + 32, h, X, Y);                     // feature_head: no activation
        Act in9[HT + 1][NT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) in9[t][n] = Y[t][n];
#pragma unroll
        for (int n = 0; n < NT; ++n) in9[HT][n] = dir[0][n];
        Act c0[HT / 2][NT], c1[HT / 4][NT];
        dense_act<Mode, HT + 1, HT / 2, NT, true>(pipe, bias + 32 + 32 * HT, h, in9, c0);
        dense_act<Mode, HT / 2, HT / 4, NT, true>(pipe, bias + 32 + 32 * HT + 16 * HT, h, c0, c1);
        f32x16 rgb[NT];
        dense_head<Mode, HT / 4, NT>(pipe, bias + 32 + 32 * HT + 16 * HT + 8 * HT, h, c1, rgb);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            out4[n][0] = rgb[n][0]; out4[n][1] = rgb[n][1]; out4[n][2] = rgb[n][2];
            out4[n][3] = dens[n][0];
        }
    }

    template <class P>
    __device__ static __forceinline__ void eval(P& pipe, const NRF_LDS float* bias, int h, int n_layers,
                                                const Act (&enc)[KT0][NT], const Act (&dir)[1][NT], float (&out4)[NT][4]) {
        Act A[HT][NT], B[HT][NT];
        dense_act<Mode, KT0, HT, NT, true>(pipe, bias, h, enc, A);
        int boff = 32 * HT;
        const int hidden = n_layers - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            dense_act<Mode, HT, HT, NT, true>(pipe, bias + boff, h, A, B); boff += 32 * HT;
            dense_act<Mode, HT, HT, NT, true>(pipe, bias + boff, h, B, A); boff += 32 * HT;
        }
        if (hidden & 1) {
            dense_act<Mode, HT, HT, NT, true>(pipe, bias + boff, h, A, B); boff += 32 * HT;
            tail(pipe, bias + boff, h, B, A, dir, out4);
        } else {
            tail(pipe, bias + boff, h, A, B, dir, out4);
        }
    }
};

}  // namespace nrf

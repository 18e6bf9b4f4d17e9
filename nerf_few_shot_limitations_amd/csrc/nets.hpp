// nets.hpp -- the encoders and the per-family layer walks on top of mlp_core.hpp.
// The order of dense() calls here IS the order of layers in packing.cpp's plan.
#pragma once
#include "device_math.hpp"
#include "feature_map.hpp"
#include "mlp_core.hpp"

#include <type_traits>

namespace nrf {

// 1/(2 pi) split in two floats for an exact-ish turn count: rev = x*C_HI (+ error term)
constexpr float kInv2PiHi = 0x1.45f306p-3f;     // fl32(1/(2 pi))
constexpr float kInv2PiLo = 0x1.b9391p-28f;     // 1/(2 pi) - kInv2PiHi

// sin(2 pi v) for |v| <= 0.2512 turns: 2 pi v + v^3 Q(v^2), Q a degree-3 least-squares fit on Chebyshev nodes (fit error 7e-9);
// with the leading coefficient split in two floats the fp32 evaluation stays within 1.5e-7 of the exact value.
__device__ __forceinline__ float sin_turns_poly(float v) {
    constexpr float k2PiHi = 0x1.921fb6p+2f, k2PiLo = -0x1.777a5cp-23f;      // 2 pi = hi + lo
    const float w = __fmul_rn(v, v);
    float q = 0x1.419126p+5f;
    q = __builtin_fmaf(q, w, -0x1.328822p+6f);
    q = __builtin_fmaf(q, w, 0x1.466ad6p+6f);
    q = __builtin_fmaf(q, w, -0x1.4abbcep+5f);
    const float t = __builtin_fmaf(q, w, k2PiLo);
    return __builtin_fmaf(v, k2PiHi, __fmul_rn(v, t));
}

// sin (want_cos = 0) or cos (1) of 2 pi (g + small), g in [-0.5, 0.5] turns exactly reduced, `u` = g plus the low-order part of
// the turn count: fold to [-0.25, 0.25] -- cos(2 pi u) = sin(2 pi (1/4 - |u|)); sin(2 pi u) = sgn(u) sin(2 pi min(|u|, 1/2 - |u|))
// -- every subtraction of the fold is exact (Sterbenz) except 1/4 - |u| below 1/8, where it loses < 1e-8 turns.
__device__ __forceinline__ float sin_or_cos_turns(float u, int want_cos) {
    const float au = __builtin_fabsf(u);
    const float vs = fminf(au, __fsub_rn(0.5f, au));
    const float vsin = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, vs) ^ (__builtin_bit_cast(uint32_t, u) & 0x80000000u));
    const float vcos = __fsub_rn(0.25f, au);
    return sin_turns_poly(want_cos ? vcos : vsin);
}

__device__ __attribute__((noinline)) float precise_sin_or_cos(float arg, int want_cos) {
    float s, c;
    sincosf(arg, &s, &c);
    return want_cos ? c : s;
}

// Positional encoding of one 3-vector into KT operand tiles, for lane half h
// (feature_map.hpp: half 0 = sines + x,y; half 1 = cosines + z).
// positional_encoding.py:27-33: sin/cos of x * 2^f, arguments exact (power-of-two scaling).
// `scale` multiplies every feature (NeRFDINOFusion's attention weight, lora_dino.py:187); 1 elsewhere.
template <class Mode, int L>
__device__ __forceinline__ void encode3(const float p[3], int h, typename Mode::Act (&out)[pe_tiles(L)], float scale = 1.0f) {
    constexpr int KT = pe_tiles(L);
    f32x16 e[KT];
    float hi[3], lo[3];
    if constexpr (Mode::TRIG != 0) {
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
            if constexpr (Mode::TRIG == 1) {
                // turns: fract() of the exact scaled high part + scaled low part; cos = sin a quarter turn on
                const float rev = __builtin_fmaf(lo[c], scale, __builtin_amdgcn_fractf(hi[c] * scale)) + quarter;
                v = __builtin_amdgcn_sinf(rev);
            } else if constexpr (Mode::TRIG == 2) {
                // parity-grade and cheap: the same exact reduction (the scaled high part minus its nearest integer is exact),
                // then a polynomial instead of v_sin: <= 2e-7 abs for |x| <= 8 and every frequency up to 2^14
                const float a = hi[c] * scale;
                const float g = __fsub_rn(a, __builtin_rintf(a));
                v = sin_or_cos_turns(__builtin_fmaf(lo[c], scale, g), h);
            } else {
                v = precise_sin_or_cos(p[c] * scale, h);
            }
        } else if constexpr (u == 3 * L) {
            v = h ? p[2] : p[0];
        } else if constexpr (u == 3 * L + 1) {
            v = h ? 0.0f : p[1];
        }
        e[t][r] = v * scale;
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

    static constexpr bool kDino = false;
    // `inputs(w0, w1, x, pass)` fills the first layer's operand tiles (the encoder lives with the caller: fused
    // renderer = from the ray, staged forward = from memory); w0/w1 are only meaningful for NetV3
    template <class P, class Inputs, class Dirs>
    __device__ static __forceinline__ void eval(P& pipe, const NRF_LDS float* bias, int h, int n_layers,
                                                Inputs&& inputs, Dirs&&, float (&out4)[NT][4]) {
        Act A[HT][NT], B[HT][NT];
        // the layers of a pass are chained (mlp_core.hpp "Chained layers"): each leaves its last tile and the next layer's first
        // fragments / bias in `cy`
        Carry<Mode, NT> cy;
        {
            Act enc[KT0][NT];
            float one[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) one[n] = 1.0f;
            inputs(one, one, enc, std::integral_constant<int, 0>{});
            dense_act_chain<Mode, KT0, HT, NT, true, kCarryNone, kCarryTail>(pipe, bias, bias + 32 * HT, h, enc, A, cy);
        }
        int boff = 32 * HT;
        const int hidden = n_layers - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, B, A, cy); boff += 32 * HT;
        }
        f32x16 head[NT];
        if (hidden & 1) {
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
            dense_head_chain<Mode, HT, NT, kCarryTail, kCarryNone>(pipe, bias + boff, bias + boff, h, B, head, cy);
        } else {
            dense_head_chain<Mode, HT, NT, kCarryTail, kCarryNone>(pipe, bias + boff, bias + boff, h, A, head, cy);
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
    static constexpr bool kDino = false;
    typedef typename Mode::Act Act;

    // density_head, feature_head, colour layers; X = trunk output, Y = scratch of the same shape
    // `dir(tile)` builds the direction-encoding tile only now, right before the colour branch needs it: its
    // registers are not held across the trunk.  CIN = kCarryTail: the trunk's last layer left X's last tile (and this branch's
    // first fragments / bias) in `cy` (chained layers, mlp_core.hpp); the layers of the branch are chained among themselves.
    template <int CIN, bool TAIL_RELU = true, class P, class Dirs>
    __device__ static __forceinline__ void tail(P& pipe, const NRF_LDS float* bias, int h, Act (&X)[HT][NT],
                                                Act (&Y)[HT][NT], Dirs&& dir, float (&out4)[NT][4], Carry<Mode, NT>& cy) {
        const NRF_LDS float* b_feat = bias + 32;
        const NRF_LDS float* b_c0 = b_feat + 32 * HT;
        const NRF_LDS float* b_c1 = b_c0 + 16 * HT;
        const NRF_LDS float* b_rgb = b_c1 + 8 * HT;
        {   // keep only the density scalar alive across the colour branch, not its 16-register tile
            f32x16 dens[NT];
            dense_head_chain<Mode, HT, NT, CIN, kCarryPre, TAIL_RELU>(pipe, bias, b_feat, h, X, dens, cy);
#pragma unroll
            for (int n = 0; n < NT; ++n) out4[n][3] = dens[n][0];
        }
        dense_act_chain<Mode, HT, HT, NT, false, kCarryPre, kCarryTail>(pipe, b_feat, b_c0, h, X, Y, cy);      // feature_head: no activation
        Act in9[HT + 1][NT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) in9[t][n] = Y[t][n];
        {
            Act dt[1][NT];
            dir(dt);
#pragma unroll
            for (int n = 0; n < NT; ++n) in9[HT][n] = dt[0][n];
        }
        Act c0[HT / 2][NT], c1[HT / 4][NT];
        // the feature tile still in `cy` is operand tile HT-1 of the concatenation, produced without activation
        dense_act_chain<Mode, HT + 1, HT / 2, NT, true, kCarryTail, kCarryPre, false, HT - 1>(pipe, b_c0, b_c1, h, in9, c0, cy);
        dense_act_chain<Mode, HT / 2, HT / 4, NT, true, kCarryPre, kCarryPre>(pipe, b_c1, b_rgb, h, c0, c1, cy);
        f32x16 rgb[NT];
        dense_head_chain<Mode, HT / 4, NT, kCarryPre, kCarryNone>(pipe, b_rgb, b_rgb, h, c1, rgb, cy);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            out4[n][0] = rgb[n][0]; out4[n][1] = rgb[n][1]; out4[n][2] = rgb[n][2];
        }
    }

    template <class P, class Inputs, class Dirs>
    __device__ static __forceinline__ void eval(P& pipe, const NRF_LDS float* bias, int h, int n_layers,
                                                Inputs&& inputs, Dirs&& dir, float (&out4)[NT][4]) {
        Act A[HT][NT], B[HT][NT];
        Carry<Mode, NT> cy;
        {
            Act enc[KT0][NT];
            float one[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) one[n] = 1.0f;
            inputs(one, one, enc, std::integral_constant<int, 0>{});
            dense_act_chain<Mode, KT0, HT, NT, true, kCarryNone, kCarryTail>(pipe, bias, bias + 32 * HT, h, enc, A, cy);
        }
        int boff = 32 * HT;
        const int hidden = n_layers - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, B, A, cy); boff += 32 * HT;
        }
        if (hidden & 1) {
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
            tail<kCarryTail>(pipe, bias + boff, h, B, A, dir, out4, cy);
        } else {
            tail<kCarryTail>(pipe, bias + boff, h, A, B, dir, out4, cy);
        }
    }
};

// ---------------------------------------------------------------------------
// V3: nerf_mlp.py:134-158 NeRFWithDINO = NeRFDINOFusion (lora_dino.py:171-193) -> DensityMLP -> ColorMLP
//   fused = fusion(cat[pe, dino]); w = softmax(attention(fused));
//   x     = output_proj(fusion(cat[pe*w0, dino*w1]))        (the same `fusion` weights, streamed twice)
// The first-layer operand tiles are rebuilt for the second pass (re-encode + re-fetch, scaled) instead of
// being kept in registers across the first pass: 40 VGPRs cheaper than holding them.
// ---------------------------------------------------------------------------
template <class Mode, int NT, int LP, int DT_>
struct NetV3 {
    static constexpr int PT = pe_tiles(LP);
    static constexpr int DT = DT_;                 // dino_dim / 32: 2 (single-scale, 64) or 4 (multi-scale, 128)
    static constexpr int KT0 = PT + DT;
    static constexpr int HT = 8;
    static constexpr bool kNeedsDir = true;
    static constexpr bool kDino = true;
    typedef typename Mode::Act Act;

    template <class P, class Inputs, class Dirs>
    __device__ static __forceinline__ void eval(P& pipe, const NRF_LDS float* bias, int h, int n_layers,
                                                Inputs&& inputs, Dirs&& dir, float (&out4)[NT][4]) {
        Act A[HT][NT], B[HT][NT];
        float w0[NT], w1[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) w0[n] = w1[n] = 1.0f;
        // chained layers (mlp_core.hpp): bias offsets of the walk, in stream order
        const NRF_LDS float* b_f1 = bias;                       // fusion layer 1 (first pass)
        const NRF_LDS float* b_f2 = b_f1 + 32 * HT;             // fusion layer 2
        const NRF_LDS float* b_a0 = b_f2 + 32 * HT;             // attention hidden (HT/4 tiles)
        const NRF_LDS float* b_a1 = b_a0 + 8 * HT;              // attention logits (one tile)
        const NRF_LDS float* b_g1 = b_a1 + 32;                  // fusion layer 1 (second pass, gated inputs)
        const NRF_LDS float* b_g2 = b_g1 + 32 * HT;
        const NRF_LDS float* b_op = b_g2 + 32 * HT;             // output_proj
        const NRF_LDS float* b_tr = b_op + 32 * HT;             // density trunk
        Carry<Mode, NT> cy;
        {
            Act x[KT0][NT];
            inputs(w0, w1, x, std::integral_constant<int, 0>{});
            dense_act_chain<Mode, KT0, HT, NT, true, kCarryNone, kCarryTail>(pipe, b_f1, b_f2, h, x, A, cy);
        }
        dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, b_f2, b_a0, h, A, B, cy);
        {   // attention: Linear(256->64)+ReLU, Linear(64->2), softmax over the pair (lora_dino.py:162-167,184)
            Act a0[HT / 4][NT];
            dense_act_chain<Mode, HT, HT / 4, NT, true, kCarryTail, kCarryPre>(pipe, b_a0, b_a1, h, B, a0, cy);
            f32x16 lg[NT];
            dense_head_chain<Mode, HT / 4, NT, kCarryPre, kCarryPre>(pipe, b_a1, b_g1, h, a0, lg, cy);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float d = lg[n][1] - lg[n][0];
                w0[n] = 1.0f / (1.0f + (Mode::FAST_EXP ? __expf(d) : expf(d)));
                w1[n] = 1.0f - w0[n];
            }
        }
        {
            Act x[KT0][NT];
            inputs(w0, w1, x, std::integral_constant<int, 1>{});
            dense_act_chain<Mode, KT0, HT, NT, true, kCarryPre, kCarryTail>(pipe, b_g1, b_g2, h, x, A, cy);
        }
        dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, b_g2, b_op, h, A, B, cy);
        dense_act_chain<Mode, HT, HT, NT, false, kCarryTail, kCarryTail>(pipe, b_op, b_tr, h, B, A, cy);       // output_proj: no activation
        int boff = (int)(b_tr - bias);
        // the first trunk layer completes the un-activated output_proj tile, the others ReLU tiles
        if (n_layers >= 2) {
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail, false>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, B, A, cy); boff += 32 * HT;
            for (int p = 1; p < n_layers / 2; ++p) {
                dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
                dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, B, A, cy); boff += 32 * HT;
            }
            if (n_layers & 1) {
                dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
                NetV2<Mode, NT, LP>::template tail<kCarryTail>(pipe, bias + boff, h, B, A, dir, out4, cy);
            } else {
                NetV2<Mode, NT, LP>::template tail<kCarryTail>(pipe, bias + boff, h, A, B, dir, out4, cy);
            }
        } else if (n_layers == 1) {
            dense_act_chain<Mode, HT, HT, NT, true, kCarryTail, kCarryTail, false>(pipe, bias + boff, bias + boff + 32 * HT, h, A, B, cy); boff += 32 * HT;
            NetV2<Mode, NT, LP>::template tail<kCarryTail>(pipe, bias + boff, h, B, A, dir, out4, cy);
        } else {
            NetV2<Mode, NT, LP>::template tail<kCarryTail, false>(pipe, bias + boff, h, A, B, dir, out4, cy);
        }
    }
};

// ---------------------------------------------------------------------------
// DINO side channel (ray_utils.py:176-210 + dino_feature_model.py:114-148): project a sample into the source
// view, bilinear taps of the (1,Hp,Wp,C) map with zeros padding, align_corners=False.
// ---------------------------------------------------------------------------
struct DinoTaps {
    int off[4];      // element offset of the tap's channel 0, or -1 outside the map
    float w[4];
};

template <class DinoT>
__device__ __forceinline__ DinoTaps dino_taps(const DinoT& d, const float p[3]) {
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        pc[i] = d.inv_pose[4 * i + 0] * p[0] + d.inv_pose[4 * i + 1] * p[1] + d.inv_pose[4 * i + 2] * p[2] + d.inv_pose[4 * i + 3];
    // the camera's wave-uniform constants go through SGPRs (device_math.hpp:uniform_f): as hoisted VGPR values they were spilled
    const float Wf = uniform_f((float)d.W), Hf = uniform_f((float)d.H), Wpf = uniform_f((float)d.Wp), Hpf = uniform_f((float)d.Hp);
    const float halfW = uniform_f((float)d.W / 2.0f), halfH = uniform_f((float)d.H / 2.0f);
    const float Wp1 = uniform_f((float)(d.Wp - 1)), Hp1 = uniform_f((float)(d.Hp - 1));
    const float zi = pc[2] + 1e-8f;
    const float xn = (pc[0] / zi * d.focal + halfW) / Wf * 2.0f - 1.0f;
    const float yn = (pc[1] / zi * d.focal + halfH) / Hf * 2.0f - 1.0f;
    const float gx = ((xn + 1.0f) * Wpf - 1.0f) * 0.5f;
    const float gy = ((yn + 1.0f) * Hpf - 1.0f) * 0.5f;
    const float x0 = floorf(gx), y0 = floorf(gy);
    DinoTaps t;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float xi = x0 + dx, yi = y0 + dy;
            const float wx = dx ? gx - x0 : x0 + 1.0f - gx;
            const float wy = dy ? gy - y0 : y0 + 1.0f - gy;
            const bool ok = xi >= 0.0f && xi <= Wp1 && yi >= 0.0f && yi <= Hp1;
            t.off[2 * dy + dx] = ok ? ((int)yi * d.Wp + (int)xi) * d.C : -1;
            t.w[2 * dy + dx] = ok ? wx * wy : 0.0f;
        }
    return t;
}

// the fetched channels of lane half h, blended in fp32: e[16 t + 4 g + q] = channel 32t + 8g + 4h + q
template <int DT>
__device__ __forceinline__ void dino_blend(const float* __restrict__ feat, const DinoTaps& tp, int h, float (&e)[16 * DT]) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * t + 8 * g + 4 * h;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (tp.off[k] >= 0) {
                    const f32x4 v = *(const f32x4*)(feat + tp.off[k] + ch);
                    acc += v * tp.w[k];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) e[16 * t + 4 * g + q] = acc[q];
        }
    }
}

// The same gather in two steps, so that the caller can put work between them: issue() starts all 16 DT tap loads of a lane half
// unconditionally (a tap outside the map reads the map's first texel with weight 0: 0 * v adds exactly nothing, as zeros padding does),
// finish() blends them in the order of dino_blend (bit-identical result).  The loads are L2 hits whose latency a lone wave per SIMD
// cannot cover by itself: with the positional encoding of the same column (~600 VALU instructions) between the two calls it is
// (ablation: the gather cost 3.7 % of a V3 frame, the same with every load on one hot address).
template <int DT>
struct DinoRaw {
    f32x4 v[DT][4][4];       // [tile][register group][tap]
    __device__ __forceinline__ void issue(const float* __restrict__ feat, const DinoTaps& tp, int h) {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    v[t][g][k] = *(const f32x4*)(feat + (tp.off[k] >= 0 ? tp.off[k] : 0) + 32 * t + 8 * g + 4 * h);
    }
    __device__ __forceinline__ void finish(const DinoTaps& tp, float (&e)[16 * DT]) const {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += v[t][g][k] * tp.w[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) e[16 * t + 4 * g + q] = acc[q];
            }
    }
};

// ... as DT operand tiles, scaled by the fusion gate
template <class Mode, int DT>
__device__ __forceinline__ void dino_scaled_tiles(const float (&e)[16 * DT], float scale, typename Mode::Act (&out)[DT]) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = e[16 * t + r] * scale;
        out[t] = Mode::template to_act<false>(v);
    }
}

// What a column of the fused V3 renderer keeps of its gathered feature-map channels between NetV3's two fusion passes (the
// gather -- two rounds of global loads whose latency a lone wave cannot cover -- happens once per sample).
//   * default: the blended fp32 channels, 16 DT registers per operand tile; both passes round them once, exactly as the
//     reference's arithmetic does (round16(e * w1)).
//   * f16 at dino_dim 128 (DT = 4) only: the FIRST pass's own operand tiles (the channels rounded to 16 bits, 8 DT registers: half
//     the hold); the second pass rescales THOSE by the gate: round16(round16(e) * w1), one more rounding of an input (<= 2^-11
//     relative).  With fp32 channels that kernel spills 53 registers inside the MLP, with the packed hold 4.  Measured before
//     choosing (profiles/r03_ab_v3_hold.txt): at dino_dim 64 the two holds run at the same speed (57.0 ms either way) once the
//     other spill sources are gone, f16 loses 0.2 dB of its 87 dB against the oracle -- and bf16 loses 24 dB (70.8 -> 46.7 dB, max
//     rgb error 0.009 -> 0.85: a second 2^-8 rounding in front of the softmax gate), so bf16 never holds packed.
#ifndef NRF_DINO_HOLD_F32
#define NRF_DINO_HOLD_F32 0      // 1 (A/B builds): every mode holds fp32 channels
#endif
#ifndef NRF_DINO_HOLD_PACKED
#define NRF_DINO_HOLD_PACKED 0   // 1 (A/B builds): both 16-bit modes hold packed tiles at every width (the first round-3 build)
#endif
template <class Mode, int DT, bool PACKED = (sizeof(typename Mode::Act) == 32 && !NRF_DINO_HOLD_F32 &&
                                              (NRF_DINO_HOLD_PACKED || (std::is_same<Mode, ModeF16>::value && DT > 2)))>
struct DinoHeld {
    float e[16 * DT];
    __device__ __forceinline__ void gather(const float* __restrict__ feat, const DinoTaps& tp, int h) { dino_blend<DT>(feat, tp, h, e); }
    __device__ __forceinline__ void finish(const DinoRaw<DT>& raw, const DinoTaps& tp) { raw.finish(tp, e); }
    template <int PASS>
    __device__ __forceinline__ void tiles(float scale, typename Mode::Act (&out)[DT]) const { dino_scaled_tiles<Mode, DT>(e, scale, out); }
};

template <class Mode, int DT>
struct DinoHeld<Mode, DT, true> {
    typename Mode::Act t[DT];
    __device__ __forceinline__ void gather(const float* __restrict__ feat, const DinoTaps& tp, int h) {
        float e[16 * DT];
        dino_blend<DT>(feat, tp, h, e);
        dino_scaled_tiles<Mode, DT>(e, 1.0f, t);
    }
    __device__ __forceinline__ void finish(const DinoRaw<DT>& raw, const DinoTaps& tp) {
        float e[16 * DT];
        raw.finish(tp, e);
        dino_scaled_tiles<Mode, DT>(e, 1.0f, t);
    }
    template <int PASS>
    __device__ __forceinline__ void tiles(float scale, typename Mode::Act (&out)[DT]) const {
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            if constexpr (PASS == 0) {
                out[k] = t[k];                                   // first pass: the gate is 1
            } else {
                typedef __attribute__((ext_vector_type(8))) float f32x8;
                f32x16 v;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f32x8 w = __builtin_convertvector(t[k].f[s], f32x8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[8 * s + j] = w[j] * scale;
                }
                out[k] = Mode::template to_act<false>(v);
            }
        }
    }
};

template <class Mode, int DT>
__device__ __forceinline__ void dino_tiles(const float* __restrict__ feat, const DinoTaps& tp, int h, float scale,
                                           typename Mode::Act (&out)[DT]) {
    float e[16 * DT];
    dino_blend<DT>(feat, tp, h, e);
    dino_scaled_tiles<Mode, DT>(e, scale, out);
}

}  // namespace nrf

// train_v3_impl.hpp -- training kernels of the V3 network (nerf_mlp.py:86-158 NeRFWithDINO with lora_dino.py:146-193
// NeRFDINOFusion in front of the V2 body).  Same machinery as train_impl.hpp / train_v2_impl.hpp.
//
//   fused = fusion(cat[pe, dino]);  (w0, w1) = softmax(attention(fused));
//   x     = output_proj(fusion(cat[pe * w0, dino * w1]))          -- the SAME fusion weights, twice
//   then DensityMLP(x) and ColorMLP as in V2.
// No gradient with respect to positions, directions or the per-sample DINO features (the reference's feature extractor is
// outside the path).
//
// Saved-tensor slots (n = trunk layers, KT0 = PE tiles + dino tiles, D = 11 + n):
//   0  [pe | dino] (KT0)            4  [pe*w0 | dino*w1] (KT0)      8+j   trunk layer j output (8)
//   1  fusion.0 out, pass 1 (8)     5  fusion.0 out, pass 2 (8)     8+n   [feature_vec | PE(dir)] (9)
//   2  fusion.2 out = fused (8)     6  fusion.2 out, pass 2 (8)     9+n   colour layer 0 output (4)
//   3  attention.0 out (2)          7  output_proj out (8)          10+n  colour layer 2 output (2)
//   D+0 dZ fusion.0 p1   D+1 dZ fusion.2 p1   D+2 dZ attention.0 (2)   D+3 dZ attention.2 = d logits (1)
//   D+4 dZ fusion.0 p2   D+5 dZ fusion.2 p2   D+6 dZ output_proj       D+7+j dZ trunk j
//   D+7+n dZ density_head (1)   D+8+n d feature_vec   D+9+n dZ colour 0 (4)   D+10+n dZ colour 2 (2)   D+11+n d rgb logits (1)
// ReLU bit planes: 0 fusion.0 p1, 1 fusion.2 p1, 2 attention.0, 3 fusion.0 p2, 4 fusion.2 p2, 5+j trunk j, 5+n colour 0,
// 6+n colour 2.  Aux: the gate (w0, w1) per sample.
#pragma once
#include "train_impl.hpp"

namespace nrf {

// operand tile -> fp32 registers (element r of the result = accumulator-row order of the tile)
template <class Mode> struct ActF32;
template <>
struct ActF32<ModeBF16> {
    __device__ static __forceinline__ f32x16 get(const ModeBF16::Act& a) {
        f32x16 o;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const i32x4 q = __builtin_bit_cast(i32x4, a.f[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = q[j];
                o[8 * s + 2 * j] = __builtin_bit_cast(float, d << 16);
                o[8 * s + 2 * j + 1] = __builtin_bit_cast(float, (int)((uint32_t)d & 0xffff0000u));
            }
        }
        return o;
    }
};
template <>
struct ActF32<ModeF16> {
    __device__ static __forceinline__ f32x16 get(const ModeF16::Act& a) {
        f32x16 o;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) o[8 * s + j] = (float)a.f[s][j];
        return o;
    }
};
template <>
struct ActF32<ModeF32> {
    __device__ static __forceinline__ f32x16 get(const ModeF32::Act& a) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = a.r[r];
        return o;
    }
};

__device__ __forceinline__ float2* gate_ptr(const TrainKArgs& P, int64_t sample) { return (float2*)(P.ctx + P.aux_off) + sample; }

template <class Mode, int WAVES, int LP, int LD, int DT>
__global__ void __launch_bounds__(WAVES * 64) train_forward_v3_kernel(const TrainKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    constexpr int PT = pe_tiles(LP), KT0 = PT + DT, HT = 8;

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    load_bias_table(bias, P.net.bias, P.net.n_bias);
    Pipe<WAVES> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, 0);
    pipe.start();
    const int n = P.net.n_layers;

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const int64_t st = tile * WAVES + wave;
        const int64_t raw = st * 32 + c;
        const int64_t sid = raw < P.n ? raw : P.n - 1;
        float p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = P.pos[sid * 3 + k];
        // first-layer operand tiles [pe * w0 | dino * w1], saved into `slot` (lora_dino.py:181,187-191)
        auto inputs = [&](float w0, float w1, Act (&x)[KT0][1], int slot) {
            Act e1[PT];
            encode3<Mode, LP>(p, h, e1, w0);
#pragma unroll
            for (int t = 0; t < PT; ++t) x[t][0] = e1[t];
            const float* f = P.dino + sid * (32 * DT);
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                f32x16 e;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *(const f32x4*)(f + 32 * t + 8 * g + 4 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[4 * g + q] = v[q] * w1;
                }
                x[PT + t][0] = Mode::template to_act<false>(e);
            }
#pragma unroll
            for (int t = 0; t < KT0; ++t) IO::store_g(tile_ptr<Mode>(P, slot, st, t, lane), x[t][0]);
        };
        // Linear + ReLU with saved output (slot) and ReLU bits (plane)
        auto relu_layer = [&](const auto& in, auto& out, auto kt_, auto mt_, int slot, int plane, int boff) {
            constexpr int KT = decltype(kt_)::value, MT = decltype(mt_)::value;
            i32x4 mw = {};
            dense<Mode, KT, MT, 1>(pipe, bias + boff, h, in, [&](auto m_, f32x16(&acc)[1]) {
                constexpr int m = decltype(m_)::value;
                out[m][0] = Mode::template to_act<true>(acc[0]);
                __builtin_amdgcn_sched_barrier(0);   // relu_bits is inline asm: it must come after a compiler-visible read of the accumulators (MFMA -> VALU hazard)
                put_bits<m>(mw, relu_bits(acc[0]));
                IO::store_g(tile_ptr<Mode>(P, slot, st, m, lane), out[m][0]);
                if constexpr (m == MT - 1) *mask_ptr(P, plane, st, lane) = mw;
            });
        };
        typedef std::integral_constant<int, KT0> K0;
        typedef std::integral_constant<int, HT> K8;
        typedef std::integral_constant<int, HT / 4> K2;

        Act A[HT][1], B[HT][1];
        int boff = 0;
        {
            Act x[KT0][1];
            inputs(1.0f, 1.0f, x, 0);
            relu_layer(x, A, K0{}, K8{}, 1, 0, boff); boff += 32 * HT;
        }
        relu_layer(A, B, K8{}, K8{}, 2, 1, boff); boff += 32 * HT;
        float w0, w1;
        {   // attention: Linear(256->64)+ReLU, Linear(64->2), softmax (lora_dino.py:162-167,184)
            Act a0[HT / 4][1];
            relu_layer(B, a0, K8{}, K2{}, 3, 2, boff); boff += 8 * HT;
            f32x16 lg[1];
            dense_head<Mode, HT / 4, 1>(pipe, bias + boff, h, a0, lg); boff += 32;
            const float d = lg[0][1] - lg[0][0];
            w0 = 1.0f / (1.0f + (Mode::FAST_EXP ? __expf(d) : expf(d)));
            w1 = 1.0f - w0;
            if (h == 0) *gate_ptr(P, raw) = make_float2(w0, w1);              // padded samples included: the context is padded
        }
        {
            Act x[KT0][1];
            inputs(w0, w1, x, 4);
            relu_layer(x, A, K0{}, K8{}, 5, 3, boff); boff += 32 * HT;
        }
        relu_layer(A, B, K8{}, K8{}, 6, 4, boff); boff += 32 * HT;
        dense<Mode, HT, HT, 1>(pipe, bias + boff, h, B, [&](auto m_, f32x16(&acc)[1]) {          // output_proj: no activation
            constexpr int m = decltype(m_)::value;
            A[m][0] = Mode::template to_act<false>(acc[0]);
            IO::store_g(tile_ptr<Mode>(P, 7, st, m, lane), A[m][0]);
        });
        boff += 32 * HT;

        float dens_raw = 0.0f, logit[3];
        auto tail = [&](const Act (&X)[HT][1], int tb) {
            {
                f32x16 dens[1];
                dense_head<Mode, HT, 1>(pipe, bias + tb, h, X, dens);
                dens_raw = dens[0][0];
            }
            Act in9[HT + 1][1];
            dense<Mode, HT, HT, 1>(pipe, bias + tb + 32, h, X, [&](auto m_, f32x16(&acc)[1]) {     // feature_head: no activation
                constexpr int m = decltype(m_)::value;
                in9[m][0] = Mode::template to_act<false>(acc[0]);
                IO::store_g(tile_ptr<Mode>(P, 8 + n, st, m, lane), in9[m][0]);
            });
            {
                float dd[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) dd[k] = P.dir[sid * 3 + k];
                Act t1[pe_tiles(LD)];
                encode3<Mode, LD>(dd, h, t1);
                in9[HT][0] = t1[0];
                IO::store_g(tile_ptr<Mode>(P, 8 + n, st, HT, lane), t1[0]);
            }
            Act c0[HT / 2][1], c1[HT / 4][1];
            relu_layer(in9, c0, std::integral_constant<int, HT + 1>{}, std::integral_constant<int, HT / 2>{}, 9 + n, 5 + n, tb + 32 + 32 * HT);
            relu_layer(c0, c1, std::integral_constant<int, HT / 2>{}, K2{}, 10 + n, 6 + n, tb + 32 + 32 * HT + 16 * HT);
            f32x16 rgb[1];
            dense_head<Mode, HT / 4, 1>(pipe, bias + tb + 32 + 32 * HT + 16 * HT + 8 * HT, h, c1, rgb);
            logit[0] = rgb[0][0]; logit[1] = rgb[0][1]; logit[2] = rgb[0][2];
        };
        int slot = 8, plane = 5;
        for (int q = 0; q < n / 2; ++q) {
            relu_layer(A, B, K8{}, K8{}, slot++, plane++, boff); boff += 32 * HT;
            relu_layer(B, A, K8{}, K8{}, slot++, plane++, boff); boff += 32 * HT;
        }
        if (n & 1) {
            relu_layer(A, B, K8{}, K8{}, slot++, plane++, boff); boff += 32 * HT;
            tail(B, boff);
        } else {
            tail(A, boff);
        }
        if (h == 0 && raw < P.n) {
            P.rgb[raw * 3 + 0] = sigmoid_sel<Mode::FAST_EXP>(logit[0]);
            P.rgb[raw * 3 + 1] = sigmoid_sel<Mode::FAST_EXP>(logit[1]);
            P.rgb[raw * 3 + 2] = sigmoid_sel<Mode::FAST_EXP>(logit[2]);
            P.density[raw] = fmaxf(dens_raw, 0.0f);
        }
    }
    pipe.drain();
}

template <class Mode, int WAVES, int LP, int DT>
__global__ void __launch_bounds__(WAVES * 64) train_backward_v3_kernel(const TrainKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* zero_bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    constexpr int PT = pe_tiles(LP), KT0 = PT + DT, HT = 8;

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32 * HT; i += blockDim.x) zero_bias[i] = 0.0f;
    __syncthreads();
    Pipe<WAVES> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, 0);
    pipe.start();
    const int n = P.net.n_layers, D = 11 + n;

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const int64_t st = tile * WAVES + wave;
        const int64_t raw = st * 32 + c;
        i32x4 mcur = *mask_ptr(P, 6 + n, st, lane), mnext = *mask_ptr(P, 5 + n, st, lane);
        // dZ = dH under the ReLU bits in mcur, saved into slot_dz
        auto masked = [&](auto m_, f32x16(&acc)[1], auto& out, int slot_dz) {
            constexpr int m = decltype(m_)::value;
            out[m][0] = masked_act<Mode, m>(acc[0], mcur);
            IO::store_g(tile_ptr<Mode>(P, slot_dz, st, m, lane), out[m][0]);
        };
        // dZ = dH (the layer had no activation)
        auto plain = [&](auto m_, f32x16(&acc)[1], auto& out, int slot_dz) {
            constexpr int m = decltype(m_)::value;
            out[m][0] = Mode::template to_act<false>(acc[0]);
            IO::store_g(tile_ptr<Mode>(P, slot_dz, st, m, lane), out[m][0]);
        };

        Act in9[HT + 1][1];
        {
            Act G[1][1], d1[HT / 4][1], d0[HT / 2][1];
            {
                f32x16 e = {};
                float ds = 0.0f;
                if (h == 0 && raw < P.n) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float o = P.rgb[raw * 3 + k];
                        e[k] = P.g_rgb[raw * 3 + k] * o * (1.0f - o);
                    }
                    ds = P.density[raw] > 0.0f ? P.g_density[raw] : 0.0f;
                }
                G[0][0] = Mode::template to_act<false>(e);
                IO::store_g(tile_ptr<Mode>(P, D + 11 + n, st, 0, lane), G[0][0]);
                f32x16 e2 = {};
                e2[0] = ds;
                in9[HT][0] = Mode::template to_act<false>(e2);
                IO::store_g(tile_ptr<Mode>(P, D + 7 + n, st, 0, lane), in9[HT][0]);
            }
            dense<Mode, 1, HT / 4, 1>(pipe, zero_bias, h, G, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, d1, D + 10 + n); });
            mcur = mnext;
            mnext = *mask_ptr(P, 5 + n - 1, st, lane);                   // trunk layer n-1
            dense<Mode, HT / 4, HT / 2, 1>(pipe, zero_bias, h, d1, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, d0, D + 9 + n); });
            mcur = mnext;
            dense<Mode, HT / 2, HT, 1>(pipe, zero_bias, h, d0, [&](auto m_, f32x16(&acc)[1]) { plain(m_, acc, in9, D + 8 + n); });
        }
        Act A[HT][1], B[HT][1];
        int below = n - 2;                                               // trunk layer whose bits come next
        auto prefetch = [&]() { if (below >= 0) mnext = *mask_ptr(P, 5 + below, st, lane); --below; };
        prefetch();
        dense<Mode, HT + 1, HT, 1>(pipe, zero_bias, h, in9, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, A, D + 7 + n - 1); });
        mcur = mnext;
        // below the trunk: X = dZ of trunk layer 0, Y = scratch
        auto fusion = [&](const Act (&X)[HT][1], Act (&Y)[HT][1], Act (&Xw)[HT][1]) {
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, X, [&](auto m_, f32x16(&acc)[1]) { plain(m_, acc, Y, D + 6); });       // trunk 0^T -> d output_proj
            mcur = *mask_ptr(P, 4, st, lane);
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, Y, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, Xw, D + 5); });     // output_proj^T -> dZ fusion.2 p2
            mcur = *mask_ptr(P, 3, st, lane);
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, Xw, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, Y, D + 4); });     // fusion.2^T -> dZ fusion.0 p2
            // fusion.0^T: d [pe*w0 | dino*w1]; its dot products with the unscaled inputs are d w0, d w1 (lora_dino.py:187-190)
            float dw0 = 0.0f, dw1 = 0.0f;
            dense<Mode, HT, KT0, 1>(pipe, zero_bias, h, Y, [&](auto m_, f32x16(&acc)[1]) {
                constexpr int m = decltype(m_)::value;
                const f32x16 xin = ActF32<Mode>::get(IO::template load_g<Act>(tile_ptr<Mode>(P, 0, st, m, lane)));
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s = __builtin_fmaf(acc[0][r], xin[r], s);
                if constexpr (m < PT) dw0 += s; else dw1 += s;
            });
            dw0 += __shfl_xor(dw0, 32, 64);                              // the two lane halves hold different features of the same sample
            dw1 += __shfl_xor(dw1, 32, 64);
            Act G2[1][1], da0[HT / 4][1];
            {
                const float2 w = *gate_ptr(P, raw);
                const float s = w.x * dw0 + w.y * dw1;                   // softmax': d logit_i = w_i (d w_i - sum_j w_j d w_j)
                f32x16 e = {};
                if (h == 0) { e[0] = w.x * (dw0 - s); e[1] = w.y * (dw1 - s); }
                G2[0][0] = Mode::template to_act<false>(e);
                IO::store_g(tile_ptr<Mode>(P, D + 3, st, 0, lane), G2[0][0]);
            }
            mcur = *mask_ptr(P, 2, st, lane);
            dense<Mode, 1, HT / 4, 1>(pipe, zero_bias, h, G2, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, da0, D + 2); });   // attention.2^T
            mcur = *mask_ptr(P, 1, st, lane);
            dense<Mode, HT / 4, HT, 1>(pipe, zero_bias, h, da0, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, Xw, D + 1); });  // attention.0^T -> dZ fusion.2 p1
            mcur = *mask_ptr(P, 0, st, lane);
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, Xw, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, Y, D + 0); });        // fusion.2^T -> dZ fusion.0 p1
        };
        const int hidden = n - 1;
        int slot = D + 7 + n - 2;
        for (int q = 0; q < hidden / 2; ++q) {
            prefetch();
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, A, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, B, slot); });
            mcur = mnext; --slot;
            prefetch();
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, B, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, A, slot); });
            mcur = mnext; --slot;
        }
        if (hidden & 1) {
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, A, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, B, slot); });
            fusion(B, A, B);
        } else {
            fusion(A, B, A);
        }
    }
    pipe.drain();
}

}  // namespace nrf

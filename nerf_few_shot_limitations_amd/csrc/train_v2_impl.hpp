// train_v2_impl.hpp -- training kernels of the V2 network (the one src/training/train.py:82-89 builds with
// use_dino=False: PE(pos) -> DensityMLP -> ColorMLP(cat[feature, PE(dir)]), nerf_mlp.py:41-84).  Same machinery as
// train_impl.hpp (V1); the chain is longer on both sides.
//
// Saved-tensor slots (n = trunk layers):
//   0         PE(pos) (KT0 tiles)              n+4 .. 2n+3   dZ of density_layers.{0..n-1} (8 tiles each)
//   1 .. n    trunk activations (8 tiles)      2n+4          dZ of density_head (1 tile, row 0)
//   n+1       [feature_vec | PE(dir)] (9)      2n+5          dZ of feature_head = d feature_vec (8)
//   n+2       colour layer 0 output (4)        2n+6          dZ of color_layers.0 (4)
//   n+3       colour layer 2 output (2)        2n+7          dZ of color_layers.2 (2)
//                                              2n+8          dZ of color_layers.4 = d rgb logits (1)
// Backward stream order (packing.cpp:make_backward_plan): color_layers.4^T, .2^T, .0^T (feature columns only),
// [feature_head | density_head]^T (K = 8 + 1 tiles), density_layers.{n-1..1}^T.
#pragma once
#include "train_impl.hpp"

namespace nrf {

template <class Mode, int WAVES, int LP, int LD>
__global__ void __launch_bounds__(WAVES * 64) train_forward_v2_kernel(const TrainKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    constexpr int KT0 = pe_tiles(LP), HT = 8;

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
        Act A[HT][1], B[HT][1];
        {
            float p[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = P.pos[sid * 3 + k];
            Act e1[KT0], enc[KT0][1];
            encode3<Mode, LP>(p, h, e1);
#pragma unroll
            for (int t = 0; t < KT0; ++t) {
                enc[t][0] = e1[t];
                IO::store_g(tile_ptr<Mode>(P, 0, st, t, lane), e1[t]);
            }
            i32x4 mw;
            dense<Mode, KT0, HT, 1>(pipe, bias, h, enc, [&](auto m_, f32x16(&acc)[1]) {
                constexpr int m = decltype(m_)::value;                A[m][0] = Mode::template to_act<true>(acc[0]);
                __builtin_amdgcn_sched_barrier(0);   // relu_bits is inline asm: it must come after a compiler-visible read of the accumulators (MFMA -> VALU hazard)

                put_bits<m>(mw, relu_bits(acc[0]));
                IO::store_g(tile_ptr<Mode>(P, 1, st, m, lane), A[m][0]);
                if constexpr (m == HT - 1) *mask_ptr(P, 0, st, lane) = mw;
            });
        }
        auto layer = [&](const Act (&in)[HT][1], Act (&out)[HT][1], int slot, int boff) {
            i32x4 mw;
            dense<Mode, HT, HT, 1>(pipe, bias + boff, h, in, [&](auto m_, f32x16(&acc)[1]) {
                constexpr int m = decltype(m_)::value;                out[m][0] = Mode::template to_act<true>(acc[0]);
                __builtin_amdgcn_sched_barrier(0);   // relu_bits is inline asm: it must come after a compiler-visible read of the accumulators (MFMA -> VALU hazard)

                put_bits<m>(mw, relu_bits(acc[0]));
                IO::store_g(tile_ptr<Mode>(P, slot, st, m, lane), out[m][0]);
                if constexpr (m == HT - 1) *mask_ptr(P, slot - 1, st, lane) = mw;
            });
        };
        float dens_raw = 0.0f, logit[3];
        // density_head, feature_head, colour layers on the trunk output X (nets.hpp:NetV2::tail with stores)
        auto tail = [&](const Act (&X)[HT][1], int boff) {
            {
                f32x16 dens[1];
                dense_head<Mode, HT, 1>(pipe, bias + boff, h, X, dens);
                dens_raw = dens[0][0];
            }
            Act in9[HT + 1][1];
            dense<Mode, HT, HT, 1>(pipe, bias + boff + 32, h, X, [&](auto m_, f32x16(&acc)[1]) {     // feature_head: no activation
                constexpr int m = decltype(m_)::value;
                in9[m][0] = Mode::template to_act<false>(acc[0]);
                IO::store_g(tile_ptr<Mode>(P, n + 1, st, m, lane), in9[m][0]);
            });
            {
                float dd[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) dd[k] = P.dir[sid * 3 + k];
                Act t1[pe_tiles(LD)];
                encode3<Mode, LD>(dd, h, t1);
                in9[HT][0] = t1[0];
                IO::store_g(tile_ptr<Mode>(P, n + 1, st, HT, lane), t1[0]);
            }
            Act c0[HT / 2][1], c1[HT / 4][1];
            i32x4 mw = {};
            dense<Mode, HT + 1, HT / 2, 1>(pipe, bias + boff + 32 + 32 * HT, h, in9, [&](auto m_, f32x16(&acc)[1]) {
                constexpr int m = decltype(m_)::value;                c0[m][0] = Mode::template to_act<true>(acc[0]);
                __builtin_amdgcn_sched_barrier(0);   // relu_bits is inline asm: it must come after a compiler-visible read of the accumulators (MFMA -> VALU hazard)

                put_bits<m>(mw, relu_bits(acc[0]));
                IO::store_g(tile_ptr<Mode>(P, n + 2, st, m, lane), c0[m][0]);
                if constexpr (m == HT / 2 - 1) *mask_ptr(P, n, st, lane) = mw;
            });
            i32x4 mw1 = {};
            dense<Mode, HT / 2, HT / 4, 1>(pipe, bias + boff + 32 + 32 * HT + 16 * HT, h, c0, [&](auto m_, f32x16(&acc)[1]) {
                constexpr int m = decltype(m_)::value;                c1[m][0] = Mode::template to_act<true>(acc[0]);
                __builtin_amdgcn_sched_barrier(0);   // relu_bits is inline asm: it must come after a compiler-visible read of the accumulators (MFMA -> VALU hazard)

                put_bits<m>(mw1, relu_bits(acc[0]));
                IO::store_g(tile_ptr<Mode>(P, n + 3, st, m, lane), c1[m][0]);
                if constexpr (m == HT / 4 - 1) *mask_ptr(P, n + 1, st, lane) = mw1;
            });
            f32x16 rgb[1];
            dense_head<Mode, HT / 4, 1>(pipe, bias + boff + 32 + 32 * HT + 16 * HT + 8 * HT, h, c1, rgb);
            logit[0] = rgb[0][0]; logit[1] = rgb[0][1]; logit[2] = rgb[0][2];
        };
        int boff = 32 * HT, slot = 2;
        const int hidden = n - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            layer(A, B, slot++, boff); boff += 32 * HT;
            layer(B, A, slot++, boff); boff += 32 * HT;
        }
        if (hidden & 1) {
            layer(A, B, slot++, boff); boff += 32 * HT;
            tail(B, boff);
        } else {
            tail(A, boff);
        }
        if (h == 0 && raw < P.n) {
            P.rgb[raw * 3 + 0] = sigmoid_sel<Mode::FAST_EXP>(logit[0]);
            P.rgb[raw * 3 + 1] = sigmoid_sel<Mode::FAST_EXP>(logit[1]);
            P.rgb[raw * 3 + 2] = sigmoid_sel<Mode::FAST_EXP>(logit[2]);
            P.density[raw] = fmaxf(dens_raw, 0.0f);                                 // nerf_mlp.py:63
        }
    }
    pipe.drain();
}

template <class Mode, int WAVES, int LP>
__global__ void __launch_bounds__(WAVES * 64) train_backward_v2_kernel(const TrainKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* zero_bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    constexpr int HT = 8;

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32 * HT; i += blockDim.x) zero_bias[i] = 0.0f;
    __syncthreads();
    Pipe<WAVES> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, 0);
    pipe.start();
    const int n = P.net.n_layers;

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const int64_t st = tile * WAVES + wave;
        const int64_t raw = st * 32 + c;
        // ReLU' bit planes (train_core.hpp) in the order the chain needs them: colour layer 2 (plane n+1), colour layer 0
        // (plane n), trunk layers n-1 .. 0; each is loaded while the layer before it runs
        i32x4 mcur = *mask_ptr(P, n + 1, st, lane), mnext = *mask_ptr(P, n, st, lane);
        auto masked = [&](auto m_, f32x16(&acc)[1], auto& out, int slot_dz) {
            constexpr int m = decltype(m_)::value;
            out[m][0] = masked_act<Mode, m>(acc[0], mcur);
            IO::store_g(tile_ptr<Mode>(P, slot_dz, st, m, lane), out[m][0]);
        };

        Act in9[HT + 1][1];
        {
            Act G[1][1], d1[HT / 4][1], d0[HT / 2][1];
            {   // d rgb -> d logits (sigmoid'), rows 0..2
                f32x16 e = {};
                float ds = 0.0f;
                if (h == 0 && raw < P.n) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float o = P.rgb[raw * 3 + k];
                        e[k] = P.g_rgb[raw * 3 + k] * o * (1.0f - o);
                    }
                    ds = P.density[raw] > 0.0f ? P.g_density[raw] : 0.0f;          // relu' of density_head (nerf_mlp.py:63)
                }
                G[0][0] = Mode::template to_act<false>(e);
                IO::store_g(tile_ptr<Mode>(P, 2 * n + 8, st, 0, lane), G[0][0]);
                f32x16 e2 = {};
                e2[0] = ds;
                in9[HT][0] = Mode::template to_act<false>(e2);
                IO::store_g(tile_ptr<Mode>(P, 2 * n + 4, st, 0, lane), in9[HT][0]);
            }
            dense<Mode, 1, HT / 4, 1>(pipe, zero_bias, h, G, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, d1, 2 * n + 7); });
            mcur = mnext;
            mnext = *mask_ptr(P, n - 1, st, lane);
            dense<Mode, HT / 4, HT / 2, 1>(pipe, zero_bias, h, d1, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, d0, 2 * n + 6); });
            mcur = mnext;
            dense<Mode, HT / 2, HT, 1>(pipe, zero_bias, h, d0, [&](auto m_, f32x16(&acc)[1]) {          // d feature_vec: no activation to undo
                constexpr int m = decltype(m_)::value;
                in9[m][0] = Mode::template to_act<false>(acc[0]);
                IO::store_g(tile_ptr<Mode>(P, 2 * n + 5, st, m, lane), in9[m][0]);
            });
        }
        Act A[HT][1], B[HT][1];
        int below = n - 2;            // mask plane of the trunk layer under the one being produced
        auto prefetch = [&]() { if (below >= 0) mnext = *mask_ptr(P, below, st, lane); --below; };
        // [feature_head | density_head]^T -> dZ of density_layers.{n-1}
        prefetch();
        dense<Mode, HT + 1, HT, 1>(pipe, zero_bias, h, in9, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, A, 2 * n + 3); });
        mcur = mnext;
        const int hidden = n - 1;
        int slot = 2 * n + 2;
        for (int p = 0; p < hidden / 2; ++p) {
            prefetch();
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, A, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, B, slot); });
            mcur = mnext; --slot;
            prefetch();
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, B, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, A, slot); });
            mcur = mnext; --slot;
        }
        if (hidden & 1) {
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, A, [&](auto m_, f32x16(&acc)[1]) { masked(m_, acc, B, slot); });
        }
    }
    pipe.drain();
}

}  // namespace nrf

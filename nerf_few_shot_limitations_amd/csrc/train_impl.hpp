// train_impl.hpp -- the training path of the V1 network (reference loop:
// src/training/train.py:244-292, loss.backward() through nerf_model.py:16-24):
//
//   train_forward_kernel   forward chain of fused_impl.hpp's staged forward, one 32-sample tile per wave, which also
//                          saves every layer's operand tiles (train_core.hpp);
//   train_backward_kernel  dZ chain: head^T, then layers.{n-1..1}^T streamed like the forward weights, ReLU' taken
//                          from the forward's bit planes, every dZ tile saved;
//   weight_grad_kernel     dW = dZ X^T, db = sum dZ over the samples: per Linear one 256 x 256 (or smaller) output
//                          held in the accumulators of a workgroup, the sample axis split over workgroups, partial
//                          sums left in the context and added up in a fixed order by weight_grad_reduce_kernel;
//   repack_kernel          flat fp32 parameters -> the packed operand streams (after every optimizer step);
//   adam_kernel            torch.optim.Adam's update on the flat vectors (train.py:113-118).
//
// LANE <-> SAMPLE here (not ray): a training batch is a few thousand rays (baseline.yaml:32), so the sample
// axis, not the ray axis, has to fill the chip; compositing and its backward are the staged kernels.
#pragma once
#include <algorithm>

#include "fused_impl.hpp"
#include "train_core.hpp"

namespace nrf {

struct TrainKArgs {
    NetArgs net;              // forward: forward stream + bias table; backward: the transposed stream
    const float* x_enc;       // V1: (P, pe_dim)
    const float* pos;         // V2: (P,3)
    const float* dir;         // V2: (P,3)
    const float* dino;        // V3: (P, dino_dim) per-sample features
    int64_t n;                // samples
    int64_t n_tiles;          // workgroup tiles of WAVES*32 samples
    float* out4;              // V1 forward: (P,4) written; backward: the same tensor, read (sigmoid')
    const float* g_out4;      // V1 backward: dL/d out4 (P,4)
    float* rgb;               // V2: (P,3) written by forward, read by backward
    float* density;           // V2: (P,1)
    const float* g_rgb;       // V2 backward: dL/d rgb (P,3)
    const float* g_density;   // V2 backward: dL/d density (P,1)
    char* ctx;                // saved tensors
    int64_t slot_off[kMaxSlots];
    int slot_tiles[kMaxSlots];
    int64_t mask_off[kMaxMaskSlots];   // ReLU-mask bit planes: kMaskBytes per sample tile each
    int64_t aux_off;                   // V3: (padded samples, 2) fp32 softmax gate
    int64_t partial_off;               // weight-gradient partial sums: one kPartialFloats block per workgroup
};

// ---- host-side helpers shared by train_v1.hip / train_v2.hip -------------------------------------------------
constexpr int kWgSamples = 256;          // the context is laid out for whole 256-sample groups, whatever the geometry

inline int64_t tiles32(int64_t n) { return (n + kWgSamples - 1) / kWgSamples * (kWgSamples / 32); }

inline int tile_bytes_of(int mode) { return mode == NRF_MMA_F32 ? 4 * kFragBytes : 2 * kFragBytes; }

// Weight-gradient grid (weight_grad_kernel): one workgroup per CU (128 KiB of LDS); small batches get ONE round of
// workgroups, large ones two.  The workgroups of a round are dealt to the jobs in proportion to the bytes a job reads per
// sample (KT + MT saved tiles, with a floor: a stage costs a load latency + a barrier however few tiles it moves), so that
// all of them finish together and the grid never exceeds the round (a 257th workgroup would run alone after the other 256).
// Returns the grid size; first_block[j] .. first_block[j+1] are job j's workgroups.
constexpr int kPartialFloats = 8 * 8 * 16 * 64 + 8 * 32;      // a workgroup's 256 x 256 accumulators + its 8 bias rows of 32

inline int wgrad_grid(const TrainDev& t, int mode, int64_t n_tiles32, int* first_block) {
    const int ST = mode == NRF_MMA_F32 ? 1 : 2;
    const int rounds = n_tiles32 >= 8192 ? 2 : 1;
    constexpr int min_stages = 4, cost_floor = 10;
    const int64_t max_splits = std::max<int64_t>(1, (n_tiles32 + min_stages * ST - 1) / (min_stages * ST));
    const int budget = rounds * t.cu_count;      // fewer workgroups were measured slower at every batch size (75 %: equal, 50 %: +10 %)
    auto cost = [&](int j) { return std::max(t.job_KT[j] + t.job_MT[j], cost_floor); };
    int cost_sum = 0;
    for (int j = 0; j < t.n_jobs; ++j) cost_sum += cost(j);
    int next = 0;
    for (int j = 0; j < t.n_jobs; ++j) {
        int64_t sp = (int64_t)budget * cost(j) / std::max(cost_sum, 1);      // floor: the sum stays within the budget
        sp = std::max<int64_t>(1, std::min<int64_t>(sp, max_splits));
        if (first_block) first_block[j] = next;
        next += (int)sp;
    }
    if (first_block) first_block[t.n_jobs] = next;
    return next;
}

inline bool fill_slots(const TrainDev& t, int mode, int64_t n, TrainKArgs& k, std::string& err) {
    if (t.n_slots < 1 || t.n_slots > kMaxSlots) { err = "training plan missing"; return false; }
    const int64_t nt = tiles32(n);
    int64_t off = 0;
    for (int i = 0; i < t.n_slots; ++i) {
        k.slot_off[i] = off;
        k.slot_tiles[i] = t.slot_tiles[i];
        off += nt * t.slot_tiles[i] * tile_bytes_of(mode);
    }
    if (t.n_mask_slots < 0 || t.n_mask_slots > kMaxMaskSlots) { err = "training plan: too many masked layers"; return false; }
    for (int i = 0; i < t.n_mask_slots; ++i) {
        k.mask_off[i] = off;
        off += nt * kFragBytes;
    }
    k.aux_off = off;
    off += nt * 32 * (int64_t)t.aux_floats * 4;
    k.partial_off = off;
    return true;
}

// Geometry of the chain kernels: 8 waves x 32 samples per workgroup (two waves per SIMD) is the throughput shape; a batch
// that does not even give every CU one such workgroup runs 4 waves x 32 instead (twice the workgroups, one wave per SIMD:
// the pass of a lone wave is latency, not throughput, and the reference's last schedule stage is 512 rays x 64 samples).
inline bool small_batch(const DeviceNet& net, int64_t n) {
    static const int forced = [] { const char* e = getenv("NRF_TRAIN_WAVES"); return e ? atoi(e) : 0; }();      // A/B runs: 4 or 8
    if (forced == 4) return true;
    if (forced == 8) return false;
    return tiles32(n) / 4 <= net.cu_count;
}

inline bool check_train_common(const DeviceNet& net, const TrainDev& t, int mode, std::string& err) {
    if (mode < 0 || mode > 2) { err = "unknown mma_mode"; return false; }
    if (net.arch.pos_freq != (net.arch.net == NRF_NET_V3 ? 12 : 10)) { err = "the training path is built for pos_freq 10 (V1, V2) and 12 (V3)"; return false; }
    if (!t.bstream[mode] || !t.maps) { err = "model not prepared for training"; return false; }
    if (net.n_bias > kBiasMaxFloats) { err = "bias table exceeds the LDS carve-out"; return false; }
    return true;
}

// dW/db of every Linear from the saved tensors (defined in train_v1.hip; network independent)
int launch_weight_grad(const DeviceNet& net, const TrainDev& t, int mode, const TrainKArgs& k, float* grad, hipStream_t s, std::string& err);

__device__ __forceinline__ i32x4* mask_ptr(const TrainKArgs& P, int mslot, int64_t st, int lane) {
    return (i32x4*)(P.ctx + P.mask_off[mslot] + st * (int64_t)kMaskBytes + lane * 16);
}

template <class Mode>
__device__ __forceinline__ char* tile_ptr(const TrainKArgs& P, int slot, int64_t st, int t, int lane) {
    return P.ctx + P.slot_off[slot] + ((st * P.slot_tiles[slot] + t) * (int64_t)tile_bytes<Mode>()) + lane * 16;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <class Mode, int WAVES, int LP>
__global__ void __launch_bounds__(WAVES * 64) train_forward_kernel(const TrainKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    constexpr int KT0 = pe_tiles(LP), HT = 8, PE = pe_dim(LP);

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    load_bias_table(bias, P.net.bias, P.net.n_bias);
    Pipe<WAVES> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, 0);
    pipe.start();

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const int64_t st = tile * WAVES + wave;                  // this wave's 32-sample tile
        const int64_t raw = st * 32 + c;
        const int64_t sid = raw < P.n ? raw : P.n - 1;
        Act A[HT][1], B[HT][1];
        {
            Act enc[KT0][1];
            const float* xin = P.x_enc + sid * PE;
            f32x16 e[KT0];
            static_for<16 * KT0>([&](auto u_) {                  // positional_encoding.py order -> operand order (feature_map.hpp)
                constexpr int u = decltype(u_)::value;
                constexpr int i0 = pe_ref_index(LP, u, 0), i1 = pe_ref_index(LP, u, 1);
                float val = 0.0f;
                if constexpr (i0 >= 0 && i1 >= 0) val = xin[h ? i1 : i0];
                else if constexpr (i0 >= 0) val = h ? 0.0f : xin[i0];
                else if constexpr (i1 >= 0) val = h ? xin[i1] : 0.0f;
                e[u / 16][u % 16] = val;
            });
#pragma unroll
            for (int t = 0; t < KT0; ++t) {
                enc[t][0] = Mode::template to_act<false>(e[t]);
                IO::store_g(tile_ptr<Mode>(P, 0, st, t, lane), enc[t][0]);
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
        // trunk layer writing activation slot `slot` (and the mask plane slot - 1)
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
        int boff = 32 * HT, slot = 2;
        const int hidden = P.net.n_layers - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            layer(A, B, slot++, boff); boff += 32 * HT;
            layer(B, A, slot++, boff); boff += 32 * HT;
        }
        f32x16 head[1];
        if (hidden & 1) {
            layer(A, B, slot++, boff); boff += 32 * HT;
            dense_head<Mode, HT, 1>(pipe, bias + boff, h, B, head);
        } else {
            dense_head<Mode, HT, 1>(pipe, bias + boff, h, A, head);
        }
        if (h == 0 && raw < P.n) {
            const float r = sigmoid_sel<Mode::FAST_EXP>(head[0][0]), g = sigmoid_sel<Mode::FAST_EXP>(head[0][1]),
                        b = sigmoid_sel<Mode::FAST_EXP>(head[0][2]);
            *(float4*)(P.out4 + raw * 4) = make_float4(r, g, b, head[0][3]);      // nerf_model.py:22-24
        }
    }
    pipe.drain();
}

// ---------------------------------------------------------------------------------------------
// backward chain
// ---------------------------------------------------------------------------------------------
// ReLU' comes from the forward's bit planes (train_core.hpp), 16 B per lane and layer, loaded one layer ahead
template <class Mode, int WAVES, int LP>
__global__ void __launch_bounds__(WAVES * 64) train_backward_kernel(const TrainKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* zero_bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    constexpr int HT = 8;

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32 * HT; i += blockDim.x) zero_bias[i] = 0.0f;     // the chain has no bias: accumulators start at 0
    __syncthreads();
    Pipe<WAVES> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, 0);
    pipe.start();
    const int n = P.net.n_layers;

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const int64_t st = tile * WAVES + wave;
        const int64_t raw = st * 32 + c;
        // ReLU' bits of trunk layer L (mask plane L - 1), loaded one layer ahead of their use
        i32x4 mcur = *mask_ptr(P, n - 1, st, lane), mnext = mcur;

        Act G[1][1];
        {   // d out4 -> d [rgb logits, sigma]: rows 0..3 of one operand tile (registers 0..3 of lane half 0)
            f32x16 e = {};
            if (h == 0 && raw < P.n) {
                const float4 o = *(const float4*)(P.out4 + raw * 4);
                const float4 g = *(const float4*)(P.g_out4 + raw * 4);
                e[0] = g.x * o.x * (1.0f - o.x);                                    // sigmoid'
                e[1] = g.y * o.y * (1.0f - o.y);
                e[2] = g.z * o.z * (1.0f - o.z);
                e[3] = g.w;                                                         // sigma_out has no activation
            }
            G[0][0] = Mode::template to_act<false>(e);
            IO::store_g(tile_ptr<Mode>(P, 2 * n + 1, st, 0, lane), G[0][0]);
        }
        Act A[HT][1], B[HT][1];
        auto epilogue = [&](auto m_, f32x16(&acc)[1], Act (&out)[HT][1], int slot_dz) {
            constexpr int m = decltype(m_)::value;
            out[m][0] = masked_act<Mode, m>(acc[0], mcur);
            IO::store_g(tile_ptr<Mode>(P, slot_dz, st, m, lane), out[m][0]);
        };
        int below = n - 2;            // mask plane of the layer under the one being produced
        auto prefetch = [&]() { if (below >= 0) mnext = *mask_ptr(P, below, st, lane); --below; };
        // head^T -> dZ of layers.{n-1}
        prefetch();
        dense<Mode, 1, HT, 1>(pipe, zero_bias, h, G, [&](auto m_, f32x16(&acc)[1]) { epilogue(m_, acc, A, 2 * n); });
        mcur = mnext;
        // layers.l^T, l = n-1 .. 1: dZ_l -> dZ_{l-1}
        const int hidden = n - 1;
        int slot = 2 * n - 1;
        for (int p = 0; p < hidden / 2; ++p) {
            prefetch();
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, A, [&](auto m_, f32x16(&acc)[1]) { epilogue(m_, acc, B, slot); });
            mcur = mnext; --slot;
            prefetch();
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, B, [&](auto m_, f32x16(&acc)[1]) { epilogue(m_, acc, A, slot); });
            mcur = mnext; --slot;
        }
        if (hidden & 1) {
            dense<Mode, HT, HT, 1>(pipe, zero_bias, h, A, [&](auto m_, f32x16(&acc)[1]) { epilogue(m_, acc, B, slot); });
        }
    }
    pipe.drain();
}

// ---------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------
struct GradJob {
    int64_t x_off, dz_off;      // slot bases inside the context
    int KT, MT;                 // feature tiles of X / of dZ used by this job
    int x_stride, dz_stride;    // feature tiles per sample tile of the two slots
    int x_first;                // first X tile of the window (a Linear fed by a concatenation is split into windows of <= 8 tiles)
    int map_off;                // this job's row_w | row_b | col tables inside `maps` (ints)
};

struct GradKArgs {
    const char* ctx;
    float* partial;             // per-workgroup partial sums (kPartialFloats each), reduced by weight_grad_reduce_kernel
    float* grad;                // flat gradient vector, accumulated into
    const int32_t* maps;
    GradJob jobs[kMaxJobs];
    int first_block[kMaxJobs + 1];   // job j owns workgroups [first_block[j], first_block[j+1]): its slabs of the sample axis
    int n_jobs;
    int64_t n_tiles32;          // 32-sample tiles
};

// One workgroup (8 waves) = one Linear x one slab of samples; its 256 x 256 fp32 output lives in the accumulators
// (wave w: rows 64*(w&3).., columns 128*(w>>2)..).  HBM-bound: 1 KiB of saved operands per sample and layer against
// 131 kFLOP.  Per stage of ST sample tiles, wave w transposes dZ tile w and X tile w of every sample tile on the
// matrix core (train_core.hpp) -- each tile exactly once per workgroup -- and parks the transposed operand tiles in
// LDS, where all waves read the 2 + 4 tiles their outputs need; the saved tiles of the next stage are already in
// flight (registers) while the current one is multiplied.  Two LDS buffers, one barrier per stage.
// PF = stages of saved tiles in flight per wave (registers): the kernel is a stream of 1-KiB tile loads whose only latency cover is
// what is already in flight -- at PF = 1 (rounds 1-2) a workgroup keeps 2 ST KiB per wave on the wire, 8 MB chip-wide, about half of
// what 8 TB/s x ~2 us of loaded HBM latency asks for, and the kernel sat at 3.8 TB/s.  PF = 2 doubles it in the 16-bit modes
// (2 x ST x 16 more registers; the fp32 mode's tiles are twice as large and stay at PF = 1).
template <class Mode, int ST, int PF = 1>
__global__ void __launch_bounds__(512) weight_grad_kernel(const GradKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mode::Act Act;
    typedef ActIO<Mode> IO;
    static_assert(PF == 1 || PF == 2, "one or two stages of saved tiles in flight");
    constexpr int RT = 2, CT = 4;
    constexpr int TB = tile_bytes<Mode>();
    constexpr int kStageBytes = ST * 16 * TB;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int job = 0;
    while (job + 1 < P.n_jobs && (int)blockIdx.x >= P.first_block[job + 1]) ++job;
    const int split = blockIdx.x - P.first_block[job], splits = P.first_block[job + 1] - P.first_block[job];
    const GradJob J = P.jobs[job];
    const int row0 = (wave & 3) * RT, col0 = (wave >> 2) * CT;
    const bool has_z = wave < J.MT, has_x = wave < J.KT;            // transposition duty: dZ tile `wave`, X tile `wave`

    Transposer<Mode> tr;
    tr.init(lane);
    f32x16 acc[RT][CT];
    float bsum = 0.0f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x16{};

    const int64_t per = (P.n_tiles32 + splits - 1) / splits;
    const int64_t t0 = split * per, t1 = (t0 + per < P.n_tiles32) ? t0 + per : P.n_tiles32;
    const char* xb = P.ctx + J.x_off + lane * 16;
    const char* zb = P.ctx + J.dz_off + lane * 16;
    const Act zero = Mode::template to_act<false>(f32x16{});
    Act rz[PF][ST], rx[PF][ST];
    auto fetch = [&](auto b_, int64_t st0) {
        constexpr int b = decltype(b_)::value;
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            const bool in = st0 + q < t1;
            rz[b][q] = (in && has_z) ? IO::template load_g<Act>(zb + ((st0 + q) * J.dz_stride + wave) * (int64_t)TB) : zero;
            rx[b][q] = (in && has_x) ? IO::template load_g<Act>(xb + ((st0 + q) * J.x_stride + J.x_first + wave) * (int64_t)TB) : zero;
        }
    };
    // one stage: transpose + park the register set `b`, refill it with the stage PF ahead, multiply
    auto stage_step = [&](auto b_, int64_t st0, int buf) {
        constexpr int b = decltype(b_)::value;
        char* stage = smem + buf * kStageBytes + lane * 16;
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            if (has_z) {
                const f32x16 t = tr.run(rz[b][q]);
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += t[r];
                bsum += s;
                IO::store(stage + (q * 16 + wave) * TB, Mode::template to_act<false>(t));
            }
            if (has_x) IO::store(stage + (q * 16 + 8 + wave) * TB, Mode::template to_act<false>(tr.run(rx[b][q])));
        }
        if (st0 + PF * ST < t1) fetch(b_, st0 + PF * ST);         // the saved tiles PF stages ahead: in flight across the barrier and the MFMAs
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            Act tz[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i)
                if (row0 + i < J.MT) tz[i] = IO::template load<Act>(stage + (q * 16 + row0 + i) * TB);
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                if (col0 + j >= J.KT) continue;
                const Act tx = IO::template load<Act>(stage + (q * 16 + 8 + col0 + j) * TB);
#pragma unroll
                for (int i = 0; i < RT; ++i)
                    if (row0 + i < J.MT) OuterMma<Mode>::run(acc[i][j], tz[i], tx);
            }
        }
    };
    if (t0 < t1) fetch(std::integral_constant<int, 0>{}, t0);
    if constexpr (PF == 2)
        if (t0 + ST < t1) fetch(std::integral_constant<int, 1>{}, t0 + ST);
    int buf = 0;
    for (int64_t st0 = t0; st0 < t1; st0 += PF * ST) {
        stage_step(std::integral_constant<int, 0>{}, st0, buf);
        buf ^= 1;
        if constexpr (PF == 2) {
            if (st0 + ST < t1) {
                stage_step(std::integral_constant<int, 1>{}, st0 + ST, buf);
                buf ^= 1;
            }
        }
    }
    // hand the partial sums over: [wave][tile i*CT+j][register group of 4][lane][4 floats], 16 B per lane and store
    float* part = P.partial + (int64_t)blockIdx.x * kPartialFloats;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        if (row0 + i >= J.MT) continue;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (col0 + j >= J.KT) continue;
            f32x4* dst = (f32x4*)(part + ((wave * 8 + i * CT + j) * 16) * 64) + lane;
#pragma unroll
            // plain stores: the 61 MB of partial sums are read back by the very next kernel and fit the MALL (streaming them -- like
            // the saved tiles -- made the reduction 2.4 us slower)
            for (int q = 0; q < 4; ++q) dst[q * 64] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
    }
    if (has_z) {                                                   // bias: the wave that transposed dZ tile `wave` summed it
        const float s = bsum + __shfl_xor(bsum, 32, 64);
        if (h == 0) part[8 * 8 * 16 * 64 + 32 * wave + c] = s;
    }
}

}  // namespace nrf

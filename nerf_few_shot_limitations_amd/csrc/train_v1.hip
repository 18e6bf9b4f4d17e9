// train_v1.hip -- host launchers of the training path (V1, pos_freq 10): train_impl.hpp
#include <algorithm>

#include "train_impl.hpp"

namespace nrf {

// ---------------------------------------------------------------------------------------------
// parameter re-pack and Adam
// ---------------------------------------------------------------------------------------------
// out element i = convert(flat[src[i]]) (0 where src < 0); mode selects the operand type
// one 16-bit operand pair of the stream: bf16, f16, or the split mode's hi / lo part (pair i lies in fragment i/256; odd
// fragments carry the low parts: packing.cpp:pack_stream)
__device__ __forceinline__ uint32_t convert_pair(float a, float b, int mode, int64_t pair) {
    if (mode == NRF_MMA_BF16) return (uint32_t)pack_pair<bf16x2, false>(a, b);
    a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f);      // f16-typed streams saturate (packing.cpp:pack_stream does the same on the host)
    b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
    if (mode == NRF_MMA_F16X3 && ((pair >> 8) & 1)) {
        const f32x2 ab = {a, b};
        const f32x2 hf = __builtin_convertvector(__builtin_convertvector(ab, f16x2), f32x2);
        return (uint32_t)pack_pair<f16x2, false>(__fsub_rn(a, hf[0]), __fsub_rn(b, hf[1]));
    }
    return (uint32_t)pack_pair<f16x2, false>(a, b);
}

__global__ void __launch_bounds__(256) repack16_kernel(const float* __restrict__ flat, const int32_t* __restrict__ src, int64_t n_pairs,
                                                      int mode, uint32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        const int2 s = *(const int2*)(src + 2 * i);
        const float a = s.x >= 0 ? flat[s.x] : 0.0f, b = s.y >= 0 ? flat[s.y] : 0.0f;
        out[i] = convert_pair(a, b, mode, i);
    }
}

__global__ void __launch_bounds__(256) repack32_kernel(const float* __restrict__ flat, const int32_t* __restrict__ src, int64_t n,
                                                      float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = src[i];
        out[i] = s >= 0 ? flat[s] : 0.0f;
    }
}

// forward stream, backward stream and bias table in ONE launch (an optimisation step re-packs all three): segment k holds
// n[k] work items -- pairs of 16-bit elements, or single fp32 values when mode[k] is NRF_MMA_F32
struct Repack3Args {
    const int32_t* src[3];
    void* out[3];
    int64_t n[3];
    int mode[3];
};
__global__ void __launch_bounds__(256) repack3_kernel(const float* __restrict__ flat, const Repack3Args a) {
    const int64_t total = a.n[0] + a.n[1] + a.n[2];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k = 0;
        int64_t j = i;
        if (j >= a.n[0]) { j -= a.n[0]; k = 1; if (j >= a.n[1]) { j -= a.n[1]; k = 2; } }
        if (a.mode[k] == NRF_MMA_F32) {
            const int sidx = a.src[k][j];
            ((float*)a.out[k])[j] = sidx >= 0 ? flat[sidx] : 0.0f;
        } else {
            const int2 sp = *(const int2*)(a.src[k] + 2 * j);
            const float x = sp.x >= 0 ? flat[sp.x] : 0.0f, y = sp.y >= 0 ? flat[sp.y] : 0.0f;
            ((uint32_t*)a.out[k])[j] = convert_pair(x, y, a.mode[k], j);
        }
    }
}

// torch.optim.Adam (train.py:113-118; no amsgrad, weight decay added to the gradient, bias-corrected moments):
//   g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Side job (FusedStep): the loss VALUE of the step, loss_weight * mean over 3 n_rays values = loss_weight * sum(ray_loss) / (3 n_rays)
// with the rays' squared errors left by composite_mse_backward_kernel, added by the last workgroup in a fixed order (thread t: rays
// t, t + 256, ...; one butterfly; four partial sums): reproducible run to run.
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                  float wd, float bc1, float bc2_sqrt, const float* __restrict__ ray_loss, int64_t n_rays,
                                                  float loss_weight, float* __restrict__ loss) {
    const int extra = ray_loss ? 1 : 0;
    if (extra && blockIdx.x == 0) {                       // one workgroup more than the update needs, the first to start: it does nothing else
        __shared__ float part[4];
        float t = 0.0f;
        int64_t r = threadIdx.x;
        for (; r + 7 * 256 < n_rays; r += 8 * 256) {      // eight loads in flight, added in order
            float q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = ray_loss[r + k * 256];
#pragma unroll
            for (int k = 0; k < 8; ++k) t += q[k];
        }
        for (; r < n_rays; r += 256) t += ray_loss[r];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) *loss = loss_weight * (((part[0] + part[1]) + part[2]) + part[3]) / (3.0f * (float)n_rays);
        return;
    }
    const int64_t stride = (int64_t)(gridDim.x - extra) * blockDim.x;
    for (int64_t i = (blockIdx.x - extra) * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.0f) gi = __fadd_rn(gi, __fmul_rn(wd, pi));
        const float mi = __fadd_rn(__fmul_rn(b1, m[i]), __fmul_rn(1.0f - b1, gi));
        const float vi = __fadd_rn(__fmul_rn(b2, v[i]), __fmul_rn(__fmul_rn(1.0f - b2, gi), gi));
        m[i] = mi; v[i] = vi;
        const float denom = __fadd_rn(sqrtf(vi) / bc2_sqrt, eps);
        p[i] = __fsub_rn(pi, __fmul_rn(lr / bc1, mi / denom));
    }
}

// Second stage of the weight gradients: sum the workgroups' partial sums of a job in a fixed order and scatter them
// through the job's row / column maps into the flat gradient vector.  One block per (job, wave, tile) = 16 registers x 64
// lanes.  The only atomics left are the adds into `grad` (a weight shared by two jobs -- the fusion block of V3 -- receives
// two of them; a + b = b + a, so the result does not depend on their order): gradients are bit-reproducible run to run.
__global__ void __launch_bounds__(256) weight_grad_reduce_kernel(const GradKArgs P) {
    const int job = blockIdx.x >> 6, wt = blockIdx.x & 63, wave = wt >> 3, tile = wt & 7;
    const GradJob J = P.jobs[job];
    constexpr int RT = 2, CT = 4;
    const int i = tile / CT, j = tile % CT;
    const int row0 = (wave & 3) * RT, col0 = (wave >> 2) * CT;
    const int b0 = P.first_block[job], b1 = P.first_block[job + 1];
    const int32_t* row_w = P.maps + J.map_off;
    const int32_t* row_b = row_w + 320;
    const int32_t* colm = row_b + 320;
    if (row0 + i < J.MT && col0 + j < J.KT) {
        // thread t: register group q = t >> 6, lane = t & 63: the 16 bytes that lane stored for registers 4q .. 4q+3
        const int q = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
        const f32x4* src = (const f32x4*)(P.partial + (int64_t)b0 * kPartialFloats + (wave * 8 + tile) * 16 * 64) + threadIdx.x;
        // the partial sums are added in workgroup order (fixed: bit-reproducible), but LOADED sixteen at a time: one load per
        // iteration and a wait on it is 32 HBM latencies in a row at the reference's batch (25 us for 61 MB, round-3 trace)
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        constexpr int64_t kStep = kPartialFloats / 4;
        int b = b0;
        for (; b + 16 <= b1; b += 16, src += 16 * kStep) {
            f32x4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[k * kStep];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += v[k];
        }
        for (; b + 4 <= b1; b += 4, src += 4 * kStep) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = src[k * kStep];
#pragma unroll
            for (int k = 0; k < 4; ++k) s += v[k];
        }
        for (; b < b1; ++b, src += kStep) s += *src;
        const int col = colm[32 * (col0 + j) + c];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int r = 4 * q + sub;
            const int o = 32 * (row0 + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int w = row_w[o];
            if (w >= 0 && col >= 0) unsafeAtomicAdd(P.grad + w + col, s[sub]);
        }
    }
    if (tile == 0 && wave < J.MT && threadIdx.x < 32) {
        const float* src = P.partial + (int64_t)b0 * kPartialFloats + 8 * 8 * 16 * 64 + 32 * wave + threadIdx.x;
        float s = 0.0f;
        int b = b0;
        for (; b + 16 <= b1; b += 16, src += 16 * (int64_t)kPartialFloats) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[k * (int64_t)kPartialFloats];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += v[k];
        }
        for (; b < b1; ++b, src += kPartialFloats) s += *src;
        const int bo = row_b[32 * wave + threadIdx.x];
        if (bo >= 0) unsafeAtomicAdd(P.grad + bo, s);
    }
}

namespace {

template <class Mode, int WAVES>
int run_train_forward(const DeviceNet& net, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_forward_kernel<Mode, WAVES, 10>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err);
    if (prepared != NRF_OK) return prepared;
    k.net = net_args(net, mode);
    k.net.ablate = 0;
    k.n_tiles = tiles32(k.n) / WAVES;
    const int64_t grid = k.n_tiles < net.cu_count ? k.n_tiles : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytes, s, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("train forward launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

template <class Mode, int WAVES>
int run_train_backward(const DeviceNet& net, const TrainDev& t, int mode, TrainKArgs k, hipStream_t s, std::string& err) {
    auto kernel = train_backward_kernel<Mode, WAVES, 10>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err);
    if (prepared != NRF_OK) return prepared;
    k.net = net_args(net, mode);
    k.net.ablate = 0;
    k.net.stream = t.bstream[mode];
    k.net.n_chunks = t.n_bchunks[mode];
    k.n_tiles = tiles32(k.n) / WAVES;
    const int64_t grid = k.n_tiles < net.cu_count ? k.n_tiles : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytes, s, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("train backward launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

template <class Mode> constexpr int mode_of() { return NRF_MMA_F32; }
template <> constexpr int mode_of<ModeBF16>() { return NRF_MMA_BF16; }
template <> constexpr int mode_of<ModeF16>() { return NRF_MMA_F16; }

#ifndef NRF_WGRAD_PF
#define NRF_WGRAD_PF 1          // stages of saved tiles in flight per wave (train_impl.hpp:weight_grad_kernel); 2 measured equal (profiles/r03_ab_wgrad_prefetch.txt)
#endif

template <class Mode, int ST, int PF>
int run_weight_grad(const DeviceNet& net, const TrainDev& t, const TrainKArgs& k, float* grad, hipStream_t s, std::string& err) {
    auto kernel = weight_grad_kernel<Mode, ST, PF>;
    constexpr int kLds = 2 * ST * 16 * tile_bytes<Mode>();
    static_assert(kLds <= 160 * 1024, "weight-gradient staging exceeds the LDS");
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err, kLds);
    if (prepared != NRF_OK) return prepared;
    GradKArgs g{};
    g.ctx = k.ctx; g.grad = grad; g.maps = t.maps; g.n_jobs = t.n_jobs; g.n_tiles32 = tiles32(k.n);
    for (int j = 0; j < t.n_jobs; ++j) {
        g.jobs[j].x_off = k.slot_off[t.job_x_slot[j]];
        g.jobs[j].dz_off = k.slot_off[t.job_dz_slot[j]];
        g.jobs[j].KT = t.job_KT[j]; g.jobs[j].MT = t.job_MT[j];
        g.jobs[j].x_stride = t.slot_tiles[t.job_x_slot[j]]; g.jobs[j].dz_stride = t.slot_tiles[t.job_dz_slot[j]];
        g.jobs[j].x_first = t.job_x_first[j];
        g.jobs[j].map_off = j * kMapStride;
    }
    const unsigned grid = (unsigned)wgrad_grid(t, mode_of<Mode>(), g.n_tiles32, g.first_block);
    g.partial = (float*)(k.ctx + k.partial_off);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), kLds, s, g);
    hipLaunchKernelGGL(weight_grad_reduce_kernel, dim3((unsigned)(64 * t.n_jobs)), dim3(256), 0, s, g);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("weight gradient launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

bool check_train(const DeviceNet& net, const TrainDev& t, int mode, std::string& err) {
    if (!check_train_common(net, t, mode, err)) return false;
    if (net.arch.net != NRF_NET_V1) { err = "nrf_mlp_*_train_v1 needs a V1 model"; return false; }
    return true;
}

}  // namespace

int64_t train_ctx_bytes(const TrainDev& t, int mode, int64_t n) {
    int64_t tiles = 0;
    for (int i = 0; i < t.n_slots; ++i) tiles += t.slot_tiles[i];
    if (n <= 0) return 0;
    const int64_t nt = tiles32(n);
    return nt * (tiles * tile_bytes_of(mode) + (int64_t)t.n_mask_slots * kFragBytes + 32 * (int64_t)t.aux_floats * 4) +
           (int64_t)wgrad_grid(t, mode, nt, nullptr) * kPartialFloats * 4;
}

int launch_train_forward(const DeviceNet& net, const TrainDev& t, int mode, const float* x_enc, int64_t n, float* out4, void* ctx,
                         hipStream_t s, std::string& err) {
    if (!check_train(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.x_enc = x_enc; k.n = n; k.out4 = out4; k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    const bool small = small_batch(net, n);
    switch (mode) {
        case NRF_MMA_BF16: return small ? run_train_forward<ModeBF16, 4>(net, mode, k, s, err) : run_train_forward<ModeBF16, 8>(net, mode, k, s, err);
        case NRF_MMA_F16:  return small ? run_train_forward<ModeF16, 4>(net, mode, k, s, err) : run_train_forward<ModeF16, 8>(net, mode, k, s, err);
        default:           return run_train_forward<ModeF32, 4>(net, mode, k, s, err);
    }
}

int launch_train_backward(const DeviceNet& net, const TrainDev& t, int mode, const float* out4, const float* g_out4, int64_t n,
                          void* ctx, float* grad, hipStream_t s, std::string& err) {
    if (!check_train(net, t, mode, err)) return NRF_EINVAL;
    if (n <= 0) return NRF_OK;
    TrainKArgs k{};
    k.n = n; k.out4 = const_cast<float*>(out4); k.g_out4 = g_out4; k.ctx = (char*)ctx;
    if (!fill_slots(t, mode, n, k, err)) return NRF_EINVAL;
    int r;
    const bool small = small_batch(net, n);
    switch (mode) {
        case NRF_MMA_BF16: r = small ? run_train_backward<ModeBF16, 4>(net, t, mode, k, s, err) : run_train_backward<ModeBF16, 8>(net, t, mode, k, s, err); break;
        case NRF_MMA_F16:  r = small ? run_train_backward<ModeF16, 4>(net, t, mode, k, s, err) : run_train_backward<ModeF16, 8>(net, t, mode, k, s, err); break;
        default:           r = run_train_backward<ModeF32, 4>(net, t, mode, k, s, err); break;
    }
    if (r != NRF_OK) return r;
    return launch_weight_grad(net, t, mode, k, grad, s, err);
}

int launch_weight_grad(const DeviceNet& net, const TrainDev& t, int mode, const TrainKArgs& k, float* grad, hipStream_t s, std::string& err) {
    switch (mode) {
        case NRF_MMA_BF16: return run_weight_grad<ModeBF16, 2, NRF_WGRAD_PF>(net, t, k, grad, s, err);
        case NRF_MMA_F16:  return run_weight_grad<ModeF16, 2, NRF_WGRAD_PF>(net, t, k, grad, s, err);
        default:           return run_weight_grad<ModeF32, 1, 1>(net, t, k, grad, s, err);
    }
}

int launch_repack(const float* flat, const int32_t* src, int64_t n_elems, int mode, void* out, hipStream_t s) {
    if (n_elems <= 0) return NRF_OK;
    if (mode == NRF_MMA_F32) {
        hipLaunchKernelGGL(repack32_kernel, dim3((unsigned)((n_elems + 255) / 256 < 4096 ? (n_elems + 255) / 256 : 4096)), dim3(256), 0, s, flat, src,
                           n_elems, (float*)out);
    } else {
        const int64_t pairs = n_elems / 2;
        hipLaunchKernelGGL(repack16_kernel, dim3((unsigned)((pairs + 255) / 256 < 4096 ? (pairs + 255) / 256 : 4096)), dim3(256), 0, s, flat, src, pairs,
                           mode, (uint32_t*)out);
    }
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_repack3(const float* flat, const int32_t* const src[3], const int64_t n_elems[3], const int modes[3], void* const out[3], hipStream_t s) {
    Repack3Args a{};
    int64_t total = 0;
    for (int k = 0; k < 3; ++k) {
        a.src[k] = src[k]; a.out[k] = out[k]; a.mode[k] = modes[k];
        a.n[k] = src[k] ? (modes[k] == NRF_MMA_F32 ? n_elems[k] : n_elems[k] / 2) : 0;
        total += a.n[k];
    }
    if (total <= 0) return NRF_OK;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(repack3_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, flat, a);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step,
                const float* ray_loss, int64_t n_rays, float loss_weight, float* loss, hipStream_t s) {
    if (n <= 0) return NRF_OK;
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, (float)step));
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096) + (ray_loss ? 1u : 0u);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                       wd, bc1, bc2_sqrt, ray_loss, n_rays, loss_weight, loss);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

}  // namespace nrf

// staged_kernels.hip -- one small gfx950 kernel per reference leaf (rays, samples, encoding,
// compositor, inverse-cdf resampling, feature fetch).  They back the drop-in Python surface
// (get_rays, sample_points_along_rays, PositionalEncoding, VolumeRenderer, ...) and the
// stage-wise parity tests.  All of them are HBM-bound elementwise / per-ray kernels: the
// layouts are the reference's own row-major tensors, reads and writes are coalesced along the
// innermost axis wherever the reference layout allows it.
#include <algorithm>

#include "kernels.hpp"

namespace nrf {

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n, int block, int64_t cap = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---- a1: ray_sampler.py:4-30 ---------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) get_rays_kernel(Camera cam, int64_t ray_begin, int64_t n, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float o[3], d[3];
        camera_ray(cam, ray_begin + i, o, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) { rays_o[i * 3 + k] = o[k]; rays_d[i * 3 + k] = d[k]; }
    }
}

// ---- a2: ray_utils.py:39-84 ----------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) sample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t n_rays,
                                                        DepthLadder lad, int perturb, const float* __restrict__ t_rand, uint64_t seed,
                                                        float* __restrict__ pts, float* __restrict__ z_vals) {
    const int64_t total = n_rays * lad.S;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / lad.S;
        const int s = (int)(i - r * lad.S);
        float z;
        if (perturb) {
            const float u = t_rand ? t_rand[i] : counter_uniform(seed, (uint64_t)r, (uint32_t)s);
            z = ladder_z_jitter(lad, s, u);
        } else {
            z = ladder_z(lad, s);
        }
        if (z_vals) z_vals[i] = z;
        if (pts) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pts[i * 3 + k] = point_on_ray(rays_o[r * 3 + k], rays_d[r * 3 + k], z);
        }
    }
}

// ---- a4: positional_encoding.py:20-33 ------------------------------------------------------
__global__ void __launch_bounds__(kBlock) encode_kernel(const float* __restrict__ x, int64_t n, int dim, int L, int include_input,
                                                        const float* __restrict__ freq_bands, float* __restrict__ out) {
    const int d_out = dim * (2 * L + (include_input ? 1 : 0));
    const int64_t total = n * d_out;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t row = i / d_out;
        int f = (int)(i - row * d_out);
        float v;
        if (include_input && f < dim) {
            v = x[row * dim + f];
        } else {
            if (include_input) f -= dim;
            const int band = f / (2 * dim);
            const int rem = f - band * 2 * dim;
            const int j = rem >= dim ? rem - dim : rem;
            // log_sampling (every caller of the reference): 2^band, an exact scale; otherwise the caller's table
            // (positional_encoding.py:18: torch.linspace(1, 2^(L-1), L))
            const float arg = __fmul_rn(x[row * dim + j], freq_bands ? freq_bands[band] : (float)(1u << band));
            v = rem >= dim ? cosf(arg) : sinf(arg);
        }
        out[i] = v;
    }
}

// ---- a9/a10: nerf_mlp.py:165-215, volume_renderer.py:4-43 ------------------------------------
// HBM-bound: 20 B read (+4 B written when weights are requested) per ray-sample.  One WAVE per ray, LANE <-> sample, so
// every load/store of z, sigma, rgb and weights is a contiguous 64-element segment of the reference's (R,S,*) rows; the
// exclusive transmittance product (cumprod, :196-199) is a wave prefix product, the four sums are wave reductions.
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float u = __shfl_up(v, d, 64);
        if (lane >= d) v = __fmul_rn(v, u);
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = __fadd_rn(v, __shfl_xor(v, d, 64));
    return v;
}

// One ray on one wave (LANE <-> sample, 64-sample segments front to back): every lane returns the ray's sums
struct RaySums { float r, g, b, depth, acc; };
__device__ __forceinline__ RaySums composite_ray(const float* __restrict__ rgb, int rgb_stride, const float* __restrict__ sigma, int sigma_stride,
                                                 const float* __restrict__ z, const float* __restrict__ rays_d, int64_t r, int S, int lane,
                                                 float* __restrict__ out_w) {
    const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    const float norm = ray_norm(d);
    float T_in = 1.0f;                                   // transmittance entering this 64-sample segment
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sd = 0.0f, sa = 0.0f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool valid = s < S;
        const int64_t i = r * S + (valid ? s : S - 1);
        const float zc = z[i];
        float zn = __shfl_down(zc, 1, 64);
        if (lane == 63 && s + 1 < S) zn = z[i + 1];
        const bool last = (s + 1 == S);
        const float dist = last ? __fmul_rn(1e10f, norm) : __fmul_rn(__fsub_rn(zn, zc), norm);
        float alpha = 0.0f;
        if (valid) alpha = __fsub_rn(1.0f, expf(__fmul_rn(-fmaxf(sigma[i * sigma_stride], 0.0f), dist)));
        const float f = valid ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;
        const float incl = wave_incl_prod(f, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = __fmul_rn(alpha, __fmul_rn(T_in, excl));
        if (valid) {
            if (out_w) out_w[i] = w;
            sr = __fadd_rn(sr, __fmul_rn(w, rgb[i * rgb_stride]));
            sg = __fadd_rn(sg, __fmul_rn(w, rgb[i * rgb_stride + 1]));
            sb = __fadd_rn(sb, __fmul_rn(w, rgb[i * rgb_stride + 2]));
            sd = __fadd_rn(sd, __fmul_rn(w, zc));
            sa = __fadd_rn(sa, w);
        }
        T_in = __fmul_rn(T_in, __shfl(incl, 63, 64));
    }
    RaySums o;
    o.r = wave_sum(sr); o.g = wave_sum(sg); o.b = wave_sum(sb); o.depth = wave_sum(sd); o.acc = wave_sum(sa);
    return o;
}

__global__ void __launch_bounds__(kBlock) composite_kernel(const float* __restrict__ rgb, int rgb_stride, const float* __restrict__ sigma,
                                                           int sigma_stride, const float* __restrict__ z, const float* __restrict__ rays_d,
                                                           int64_t n_rays, int S, int white_bkgd, float* __restrict__ out_rgb,
                                                           float* __restrict__ out_depth, float* __restrict__ out_w) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t r = wave0; r < n_rays; r += n_waves) {
        RaySums o = composite_ray(rgb, rgb_stride, sigma, sigma_stride, z, rays_d, r, S, lane, out_w);
        if (lane == 0) {
            if (white_bkgd) {
                const float bg = __fsub_rn(1.0f, o.acc);
                o.r = __fadd_rn(o.r, bg); o.g = __fadd_rn(o.g, bg); o.b = __fadd_rn(o.b, bg);
            }
            out_rgb[r * 3] = o.r; out_rgb[r * 3 + 1] = o.g; out_rgb[r * 3 + 2] = o.b;
            if (out_depth) out_depth[r] = o.depth;
        }
    }
}

// ---- backward of a9 (autograd through nerf_mlp.py:181-212) -------------------------
// With v_i = g_rgb . c_i + g_depth * z_i + g_w[i] - [white_bkgd] * sum(g_rgb)  (so that dL = sum_i v_i dw_i):
//   dL/dc_i     = w_i * g_rgb
//   dL/dalpha_i = T_i * v_i - (sum_{j>i} w_j v_j) / (1 - alpha_i + 1e-10)
//   dL/dsigma_i = dL/dalpha_i * dist_i * exp(-relu(sigma_i) dist_i) * [sigma_i > 0]
// The suffix sum is a true reverse scan (segments walked back to front): "total - prefix" would lose every digit behind
// an opaque sample, where the reference's +1e-10 makes the divisor 1e-10.  One WAVE per ray, LANE <-> sample.
__device__ __forceinline__ float wave_incl_sum_rev(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float u = __shfl_down(v, d, 64);
        if (lane + d < 64) v = __fadd_rn(v, u);
    }
    return v;
}

constexpr int kMaxSegments = 64;     // S <= 4096

// One ray on one wave; gr, gg, gb, gd = dL/d rgb_map, dL/d depth of the ray (wave-uniform), g_w = dL/d weights or NULL;
// seg_T = the wave's LDS row of kMaxSegments floats
__device__ __forceinline__ void composite_backward_ray(const float* __restrict__ rgb, int rgb_stride, const float* __restrict__ sigma,
                                                       int sigma_stride, const float* __restrict__ z, const float* __restrict__ rays_d,
                                                       int64_t r, int S, int lane, int white_bkgd, float gr, float gg, float gb, float gd,
                                                       const float* __restrict__ g_w, float* __restrict__ d_rgb, int d_rgb_stride,
                                                       float* __restrict__ d_sigma, int d_sigma_stride, float* seg_T) {
    const int n_seg = (S + 63) / 64;
    const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    const float norm = ray_norm(d);
    const float bg = white_bkgd ? __fadd_rn(__fadd_rn(gr, gg), gb) : 0.0f;
    auto sample = [&](int s0, float& alpha, float& e, float& dist, float& f, float& zc, bool& valid, int64_t& i) {
        const int s = s0 + lane;
        valid = s < S;
        i = r * S + (valid ? s : S - 1);
        zc = z[i];
        float zn = __shfl_down(zc, 1, 64);
        if (lane == 63 && s + 1 < S) zn = z[i + 1];
        const bool last = (s + 1 == S);
        dist = last ? __fmul_rn(1e10f, norm) : __fmul_rn(__fsub_rn(zn, zc), norm);
        e = valid ? expf(__fmul_rn(-fmaxf(sigma[i * sigma_stride], 0.0f), dist)) : 1.0f;
        alpha = valid ? __fsub_rn(1.0f, e) : 0.0f;
        f = valid ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;
    };
    // pass 1, front to back: transmittance entering every 64-sample segment
    float T_in = 1.0f;
    for (int k = 0; k < n_seg; ++k) {
        float alpha, e, dist, f, zc; bool valid; int64_t i;
        sample(64 * k, alpha, e, dist, f, zc, valid, i);
        if (lane == 0) seg_T[k] = T_in;
        const float incl = wave_incl_prod(f, lane);
        T_in = __fmul_rn(T_in, __shfl(incl, 63, 64));
    }
    // pass 2, back to front
    float carry = 0.0f;                                  // sum of w_j v_j over all later segments
    for (int k = n_seg - 1; k >= 0; --k) {
        float alpha, e, dist, f, zc; bool valid; int64_t i;
        sample(64 * k, alpha, e, dist, f, zc, valid, i);
        const float incl = wave_incl_prod(f, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = __fmul_rn(seg_T[k], excl);
        const float w = __fmul_rn(alpha, T);
        float v = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (valid) {
            cr = rgb[i * rgb_stride]; cg = rgb[i * rgb_stride + 1]; cb = rgb[i * rgb_stride + 2];
            v = __fadd_rn(__fadd_rn(__fmul_rn(gr, cr), __fmul_rn(gg, cg)), __fmul_rn(gb, cb));
            v = __fadd_rn(v, __fmul_rn(gd, zc));
            if (g_w) v = __fadd_rn(v, g_w[i]);
            v = __fsub_rn(v, bg);
        }
        const float wv_ = valid ? __fmul_rn(w, v) : 0.0f;
        const float incl_rev = wave_incl_sum_rev(wv_, lane);
        const float suffix = __fadd_rn(__fsub_rn(incl_rev, wv_), carry);     // strictly later samples
        carry = __fadd_rn(carry, __shfl(incl_rev, 0, 64));
        if (valid) {
            const float d_alpha = __fsub_rn(__fmul_rn(T, v), suffix / f);
            const float sg = sigma[i * sigma_stride];
            d_sigma[i * d_sigma_stride] = sg > 0.0f ? __fmul_rn(__fmul_rn(d_alpha, dist), e) : 0.0f;
            d_rgb[i * d_rgb_stride] = __fmul_rn(w, gr);
            d_rgb[i * d_rgb_stride + 1] = __fmul_rn(w, gg);
            d_rgb[i * d_rgb_stride + 2] = __fmul_rn(w, gb);
        }
    }
}

__global__ void __launch_bounds__(kBlock) composite_backward_kernel(const float* __restrict__ rgb, int rgb_stride, const float* __restrict__ sigma,
                                                                    int sigma_stride, const float* __restrict__ z,
                                                                    const float* __restrict__ rays_d, int64_t n_rays, int S, int white_bkgd,
                                                                    const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                                    const float* __restrict__ g_w, float* __restrict__ d_rgb,
                                                                    int d_rgb_stride, float* __restrict__ d_sigma, int d_sigma_stride) {
    __shared__ float seg_T[kBlock / 64][kMaxSegments];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave0 = (blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t r = wave0; r < n_rays; r += n_waves) {
        const float gr = g_rgb ? g_rgb[r * 3] : 0.0f, gg = g_rgb ? g_rgb[r * 3 + 1] : 0.0f, gb = g_rgb ? g_rgb[r * 3 + 2] : 0.0f;
        const float gd = g_depth ? g_depth[r] : 0.0f;
        composite_backward_ray(rgb, rgb_stride, sigma, sigma_stride, z, rays_d, r, S, lane, white_bkgd, gr, gg, gb, gd, g_w, d_rgb, d_rgb_stride,
                               d_sigma, d_sigma_stride, seg_T[wv]);
    }
}

// The three launches between the network's forward and its backward in a FusedStep -- compositor, `rgb_weight * nn.MSELoss()` with
// its gradient, compositor backward (train.py:236,36-44,285) -- as ONE: the loss gradient of a ray needs nothing but the ray's own
// prediction and target, d loss / d pred = 2 w (pred - target) / (3 R).  Same per-ray arithmetic as the three kernels (the two bodies
// above), so d_rgb / d_sigma are bit-equal to the staged sequence.  The loss VALUE needs all rays: every ray leaves its squared error
// in `ray_loss` and a later launch adds them up in a fixed order (adam_kernel's side job, train_v1.hip) -- a device-wide
// "last workgroup sums" inside this kernel costs a release fence (an L2 write-back on this chip) per workgroup: measured 53 us
// against 16 us for the three separate launches.  Side job: `zero_buf` (the caller's flat gradient vector, which the
// weight-gradient reduction adds into) is cleared by the same launch.
__global__ void __launch_bounds__(kBlock) composite_mse_backward_kernel(const float* __restrict__ rgb, int rgb_stride,
                                                                        const float* __restrict__ sigma, int sigma_stride,
                                                                        const float* __restrict__ z, const float* __restrict__ rays_d,
                                                                        int64_t n_rays, int S, int white_bkgd, const float* __restrict__ target,
                                                                        float weight, float* __restrict__ pred, float* __restrict__ d_rgb,
                                                                        int d_rgb_stride, float* __restrict__ d_sigma, int d_sigma_stride,
                                                                        float* __restrict__ ray_loss, float* __restrict__ zero_buf, int64_t zero_n) {
    __shared__ float seg_T[kBlock / 64][kMaxSegments];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t gtid = blockIdx.x * (int64_t)kBlock + threadIdx.x, n_threads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = gtid; i < zero_n; i += n_threads) zero_buf[i] = 0.0f;
    const float count = 3.0f * (float)n_rays;
    const float scale = 2.0f * weight / count;
    for (int64_t r = gtid >> 6; r < n_rays; r += n_threads >> 6) {
        RaySums o = composite_ray(rgb, rgb_stride, sigma, sigma_stride, z, rays_d, r, S, lane, nullptr);
        if (white_bkgd) {
            const float bg = __fsub_rn(1.0f, o.acc);
            o.r = __fadd_rn(o.r, bg); o.g = __fadd_rn(o.g, bg); o.b = __fadd_rn(o.b, bg);
        }
        const float dr = o.r - target[r * 3], dg = o.g - target[r * 3 + 1], db = o.b - target[r * 3 + 2];
        if (lane == 0) {
            if (pred) { pred[r * 3] = o.r; pred[r * 3 + 1] = o.g; pred[r * 3 + 2] = o.b; }
            ray_loss[r] = dr * dr + dg * dg + db * db;
        }
        composite_backward_ray(rgb, rgb_stride, sigma, sigma_stride, z, rays_d, r, S, lane, white_bkgd, scale * dr, scale * dg, scale * db, 0.0f,
                               nullptr, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, seg_T[wv]);
    }
}


// ---- a3: ray_utils.py:86-143 (intent) --------------------------------------------------------
// One WAVE per ray, LANE <-> sample: the rows of z, weights, the new samples and the union are read and written as contiguous
// 64-element segments; the cdf is a wave prefix sum; every search (inverse cdf, merge ranks) is a branch-free binary search over
// the wave's private LDS rows.  HBM-bound: 8 S + 4 Ni + 4 (S + Ni) bytes per ray.  The coarse depths z must ascend (they are a
// depth ladder); the new samples may arrive in any order (perturbed u) and are ranked before the merge.
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float u = __shfl_up(v, d, 64);
        if (lane >= d) v = __fadd_rn(v, u);
    }
    return v;
}
// number of leading elements of the ascending row a[0..n) that are <= x (upper bound) / < x (lower bound)
__device__ __forceinline__ int count_le(const float* a, int n, int top, float x) {
    int pos = 0;
    for (int step = top; step > 0; step >>= 1)
        if (pos + step <= n && a[pos + step - 1] <= x) pos += step;
    return pos;
}
__device__ __forceinline__ int count_lt(const float* a, int n, int top, float x) {
    int pos = 0;
    for (int step = top; step > 0; step >>= 1)
        if (pos + step <= n && a[pos + step - 1] < x) pos += step;
    return pos;
}
__host__ __device__ inline int top_pow2(int n) { int t = 1; while (t * 2 <= n) t *= 2; return t; }

// ---- the two reductions of ray_utils.py:107-109 in the order PyTorch's CPU kernels use (what "the reference's CPU renderer" computes) ----
// `weights.sum(-1)` (:107) is ATen's cascade_sum (aten/src/ATen/native/cpu/SumKernel.cpp): the contiguous row is read as vectors of
// 8 floats (also on AVX-512 hosts: checked bit for bit against torch 2.10 for S = 16 ... 4096 in the build container), four
// interleaved vector accumulators (ILP), each a cascade of 16-vector blocks over up to four levels; then the scalar tail, then the
// eight lanes left to right.  Lane l < 8 of the wave plays vector lane l; the result is broadcast.
__device__ __forceinline__ int ceil_log2_i(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

__device__ float torch_cpu_row_sum(const float* x, int S, int lane) {
    constexpr int V = 8, ILP = 4, LEVELS = 4;
    const int vec_size = S / V, size_ilp = vec_size / ILP;
    const int l = lane & (V - 1);
    float acc[LEVELS][ILP];
#pragma unroll
    for (int j = 0; j < LEVELS; ++j)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[j][k] = 0.0f;
    const int quarter = ceil_log2_i(size_ilp) / LEVELS;
    const int level_power = quarter > 4 ? quarter : 4;
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    int i = 0;
    while (i + level_step <= size_ilp) {
        for (int j = 0; j < level_step; ++j, ++i)
#pragma unroll
            for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], x[(i * ILP + k) * V + l]);
        bool go = true;
#pragma unroll
        for (int j = 1; j < LEVELS; ++j) {
            if (go) {
#pragma unroll
                for (int k = 0; k < ILP; ++k) { acc[j][k] = __fadd_rn(acc[j][k], acc[j - 1][k]); acc[j - 1][k] = 0.0f; }
                if ((i & (level_mask << (j * level_power))) != 0) go = false;
            }
        }
    }
    for (; i < size_ilp; ++i)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], x[(i * ILP + k) * V + l]);
#pragma unroll
    for (int j = 1; j < LEVELS; ++j)
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[0][k] = __fadd_rn(acc[0][k], acc[j][k]);
    for (int v = size_ilp * ILP; v < vec_size; ++v) acc[0][0] = __fadd_rn(acc[0][0], x[v * V + l]);
#pragma unroll
    for (int k = 1; k < ILP; ++k) acc[0][0] = __fadd_rn(acc[0][0], acc[0][k]);
    float fin = 0.0f;
    for (int k = vec_size * V; k < S; ++k) fin = __fadd_rn(fin, x[k]);
#pragma unroll
    for (int k = 0; k < V; ++k) fin = __fadd_rn(fin, __shfl(acc[0][0], k, 64));
    return fin;
}

// `torch.cumsum` (:108) on the CPU accumulates in DOUBLE and rounds every prefix to float (ReduceOpsKernel.cpp: acc_type<float,
// false>; verified against torch 2.10).  For compositing weights (pdf >= 2^-18, sums < 2) every partial sum is a multiple of 2^-41
// below 2, i.e. exact in a double: the wave-parallel scan below then equals the sequential one bit for bit.
__device__ __forceinline__ double wave_incl_sum_f64(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double u = __shfl_up(v, d, 64);
        if (lane >= d) v = v + u;
    }
    return v;
}

__global__ void sample_pdf_kernel(const float* __restrict__ z, const float* __restrict__ w, int64_t n_rays, int S, int Ni,
                                  const float* __restrict__ u_in, int64_t u_ray_stride, float* __restrict__ samples, float* __restrict__ z_union) {
    extern __shared__ float lds_rows[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_len = 3 * S + 3 * Ni + 1;
    float* cdf = lds_rows + (size_t)wave * row_len;         // S + 1 knots
    float* zl = cdf + S + 1;                                // S coarse depths
    float* smp = zl + S;                                    // Ni new samples, in the order of u
    float* ssort = smp + Ni;                                // Ni new samples, ascending
    float* uni = ssort + Ni;                                // S + Ni merged depths
    const int64_t wave0 = blockIdx.x * (int64_t)(blockDim.x >> 6) + wave;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int topS1 = top_pow2(S + 1), topS = top_pow2(S), topN = top_pow2(Ni);
    const float ustep = Ni > 1 ? 1.0f / (float)(Ni - 1) : 0.0f;
    for (int64_t r = wave0; r < n_rays; r += n_waves) {
        const float* zr = z + r * S;
        const float* wr = w + r * S;
        // weights + 1e-5 (:104), parked in the cdf row; their sum (:107) in PyTorch's CPU order
        for (int s = lane; s < S; s += 64) {
            cdf[s + 1] = __fadd_rn(wr[s], 1e-5f);
            zl[s] = zr[s];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const float total = torch_cpu_row_sum(cdf + 1, S, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // cdf = [0, cumsum(pdf)] (:108-109): double-precision prefix sums of 64-sample segments, carried from segment to segment,
        // every knot rounded to float -- torch.cumsum's CPU arithmetic, so that the `denom < 1e-5` guard (:131) decides alike
        double carry = 0.0;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const float pdf = s < S ? cdf[s + 1] / total : 0.0f;
            const double incl = carry + wave_incl_sum_f64((double)pdf, lane);
            if (s < S) cdf[s + 1] = (float)incl;
            carry = __shfl(incl, 63, 64);
        }
        if (lane == 0) cdf[0] = 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // inverse cdf (:112-133)
        for (int j = lane; j < Ni; j += 64) {
            float u;
            if (u_in) u = u_in[r * u_ray_stride + j];
            else u = Ni == 1 ? 0.0f : ((j < Ni / 2) ? __fmul_rn(ustep, (float)j) : __fsub_rn(1.0f, __fmul_rn(ustep, (float)(Ni - 1 - j))));
            const int idx = count_le(cdf, S + 1, topS1, u);          // searchsorted(cdf, u, right=True)  (:120)
            const int below = idx - 1 > 0 ? idx - 1 : 0;
            const int above = idx < S ? idx : S;
            // bin edges [z0, mids.., z_{S-1}]: the stratification intervals of ray_utils.py:73-75
            auto edge = [&](int k) -> float {
                if (k == 0) return zl[0];
                if (k == S) return zl[S - 1];
                return __fmul_rn(0.5f, __fadd_rn(zl[k], zl[k - 1]));
            };
            float denom = __fsub_rn(cdf[above], cdf[below]);
            if (denom < 1e-5f) denom = 1.0f;                                                 // (:131)
            const float t = __fsub_rn(u, cdf[below]) / denom;
            const float eb = edge(below), ea = edge(above);
            const float v = __fadd_rn(eb, __fmul_rn(t, __fsub_rn(ea, eb)));                  // (:133)
            smp[j] = v;
            if (samples) samples[r * Ni + j] = v;
        }
        if (!z_union) continue;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // sorted union (:136): rank the new samples among themselves (stable), then merge ranks by binary search -- a coarse
        // depth goes in front of an equal new sample
        for (int j = lane; j < Ni; j += 64) {
            const float v = smp[j];
            int rank = 0;
            for (int i = 0; i < Ni; ++i) {
                const float o = smp[i];                                                      // same address in every lane: broadcast
                rank += (o < v || (o == v && i < j)) ? 1 : 0;
            }
            ssort[rank] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (int k = lane; k < Ni; k += 64) {
            const float v = ssort[k];
            uni[k + count_le(zl, S, topS, v)] = v;
        }
        for (int a = lane; a < S; a += 64) {
            const float v = zl[a];
            uni[a + count_lt(ssort, Ni, topN, v)] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        float* out = z_union + r * (S + Ni);
        for (int k = lane; k < S + Ni; k += 64) out[k] = uni[k];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// ---- a8: ray_utils.py:176-210 + dino_feature_model.py:114-148 ---------------------------------
__device__ __forceinline__ void project_point(const DinoDev& d, const float p[3], float& xn, float& yn) {
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        pc[i] = d.inv_pose[4 * i + 0] * p[0] + d.inv_pose[4 * i + 1] * p[1] + d.inv_pose[4 * i + 2] * p[2] + d.inv_pose[4 * i + 3];
    const float zi = pc[2] + 1e-8f;
    const float x = pc[0] / zi * d.focal + (float)d.W / 2.0f;
    const float y = pc[1] / zi * d.focal + (float)d.H / 2.0f;
    xn = x / (float)d.W * 2.0f - 1.0f;
    yn = y / (float)d.H * 2.0f - 1.0f;
}

// points2d: `points` already are normalised image coordinates (n,2) (sample_features_at_points on its own,
// dino_feature_model.py:114-148); otherwise world points (n,3) projected into the source view first
__global__ void __launch_bounds__(kBlock) project_fetch_kernel(DinoDev d, const float* __restrict__ points, int64_t n, float* __restrict__ feats,
                                                               float* __restrict__ xy, int points2d) {
    const int64_t total = n * d.C;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t pi = i / d.C;
        const int ch = (int)(i - pi * d.C);
        float xn, yn;
        if (points2d) {
            xn = points[pi * 2]; yn = points[pi * 2 + 1];
        } else {
            const float p[3] = {points[pi * 3], points[pi * 3 + 1], points[pi * 3 + 2]};
            project_point(d, p, xn, yn);
        }
        if (xy && ch == 0) { xy[pi * 2] = xn; xy[pi * 2 + 1] = yn; }
        // grid_sample, bilinear, zeros padding, align_corners=False
        const float gx = ((xn + 1.0f) * (float)d.Wp - 1.0f) * 0.5f;
        const float gy = ((yn + 1.0f) * (float)d.Hp - 1.0f) * 0.5f;
        const float x0 = floorf(gx), y0 = floorf(gy);
        float acc = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float xi = x0 + dx, yi = y0 + dy;
                const float wx = dx ? gx - x0 : x0 + 1.0f - gx;
                const float wy = dy ? gy - y0 : y0 + 1.0f - gy;
                if (xi >= 0.0f && xi <= (float)(d.Wp - 1) && yi >= 0.0f && yi <= (float)(d.Hp - 1))
                    acc += d.features[((int64_t)yi * d.Wp + (int64_t)xi) * d.C + ch] * (wx * wy);
            }
        feats[i] = acc;
    }
}

}  // namespace

int launch_get_rays(const Camera& cam, int64_t ray_begin, int64_t n, float* rays_o, float* rays_d, hipStream_t s) {
    if (n <= 0) return NRF_OK;
    hipLaunchKernelGGL(get_rays_kernel, dim3(grid_for(n, kBlock, 4096)), dim3(kBlock), 0, s, cam, ray_begin, n, rays_o, rays_d);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_sample(const float* rays_o, const float* rays_d, int64_t n_rays, float near, float far, int S, int lindisp, int perturb,
                  const float* t_rand, const float* z_ladder, uint64_t seed, float* pts, float* z_vals, hipStream_t s) {
    if (n_rays <= 0) return NRF_OK;
    const DepthLadder lad = make_ladder(near, far, S, lindisp, z_ladder);
    hipLaunchKernelGGL(sample_kernel, dim3(grid_for(n_rays * S, kBlock, 8192)), dim3(kBlock), 0, s, rays_o, rays_d, n_rays, lad, perturb,
                       t_rand, seed, pts, z_vals);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_encode(const float* x, int64_t n, int dim, int L, int include_input, const float* freq_bands, float* out, hipStream_t s) {
    if (n <= 0) return NRF_OK;
    const int64_t total = n * dim * (2 * L + (include_input ? 1 : 0));
    hipLaunchKernelGGL(encode_kernel, dim3(grid_for(total, kBlock, 8192)), dim3(kBlock), 0, s, x, n, dim, L, include_input, freq_bands, out);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z, const float* rays_d,
                     int64_t n_rays, int S, int white_bkgd, float* out_rgb, float* out_depth, float* out_w, hipStream_t s) {
    if (n_rays <= 0) return NRF_OK;
    hipLaunchKernelGGL(composite_kernel, dim3(grid_for(n_rays * 64, kBlock, 16384)), dim3(kBlock), 0, s, rgb, rgb_stride, sigma, sigma_stride, z,
                       rays_d, n_rays, S, white_bkgd, out_rgb, out_depth, out_w);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_composite_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z, const float* rays_d,
                              int64_t n_rays, int S, int white_bkgd, const float* g_rgb, const float* g_depth, const float* g_w,
                              float* d_rgb, int d_rgb_stride, float* d_sigma, int d_sigma_stride, hipStream_t s) {
    if (n_rays <= 0) return NRF_OK;
    if (S > 64 * kMaxSegments) return NRF_EINVAL;
    hipLaunchKernelGGL(composite_backward_kernel, dim3(grid_for(n_rays * 64, kBlock, 16384)), dim3(kBlock), 0, s, rgb, rgb_stride, sigma,
                       sigma_stride, z, rays_d, n_rays, S, white_bkgd, g_rgb, g_depth, g_w, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

// loss = w * mean((pred - target)^2) and d loss / d pred in one launch (train.py:36-44: rgb_weight * nn.MSELoss()).
// One block: the reference's ray batches are a few thousand values; fixed summation order.
__global__ void __launch_bounds__(1024) mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, int n, float weight,
                                                        float* __restrict__ g_pred, float* __restrict__ loss) {
    __shared__ float part[16];
    const float scale = 2.0f * weight / (float)n;
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float d = pred[i] - target[i];
        g_pred[i] = scale * d;
        s += d * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int k = 0; k < 16; ++k) t += part[k];
        *loss = weight * t / (float)n;
    }
}

int launch_composite_mse_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z, const float* rays_d,
                                  int64_t n_rays, int S, int white_bkgd, const float* target, float weight, float* pred, float* d_rgb,
                                  int d_rgb_stride, float* d_sigma, int d_sigma_stride, float* ray_loss, float* zero_buf,
                                  int64_t zero_n, hipStream_t s) {
    if (n_rays <= 0 || S > 64 * kMaxSegments) return NRF_EINVAL;
    const int64_t work = std::max(n_rays * 64, (zero_n + 3) / 4);
    hipLaunchKernelGGL(composite_mse_backward_kernel, dim3(grid_for(work, kBlock, 16384)), dim3(kBlock), 0, s, rgb, rgb_stride, sigma,
                       sigma_stride, z, rays_d, n_rays, S, white_bkgd, target, weight, pred, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, ray_loss,
                       zero_buf, zero_n);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_mse_grad(const float* pred, const float* target, int64_t n, float weight, float* g_pred, float* loss, hipStream_t s) {
    if (n <= 0 || n > (1 << 22)) return NRF_EINVAL;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(1024), 0, s, pred, target, (int)n, weight, g_pred, loss);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_sample_pdf(const float* z, const float* w, int64_t n_rays, int S, int Ni, const float* u, int64_t u_ray_stride, float* samples,
                      float* z_union, hipStream_t s) {
    if (n_rays <= 0) return NRF_OK;
    const int row_bytes = (3 * S + 3 * Ni + 1) * 4;          // one wave's LDS rows (cdf, depths, samples, sorted samples, union)
    int waves = 4;
    while (waves > 1 && waves * row_bytes > 60 * 1024) waves >>= 1;
    if (waves * row_bytes > 60 * 1024) return NRF_EINVAL;
    int64_t blocks = (n_rays + waves - 1) / waves;
    if (blocks > 256 * 16) blocks = 256 * 16;                // grid-stride over the rays beyond that
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)blocks), dim3(waves * 64), (size_t)waves * row_bytes, s, z, w, n_rays, S, Ni, u, u_ray_stride,
                       samples, z_union);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_project_fetch(const DinoDev& d, const float* points, int64_t n, float* feats, float* xy, hipStream_t s) {
    if (n <= 0) return NRF_OK;
    hipLaunchKernelGGL(project_fetch_kernel, dim3(grid_for(n * d.C, kBlock, 8192)), dim3(kBlock), 0, s, d, points, n, feats, xy, 0);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

int launch_sample_features(const float* features, int Hp, int Wp, int C, const float* points_2d, int64_t n, float* feats, hipStream_t s) {
    if (n <= 0) return NRF_OK;
    DinoDev d{};
    d.features = features; d.Hp = Hp; d.Wp = Wp; d.C = C;
    hipLaunchKernelGGL(project_fetch_kernel, dim3(grid_for(n * C, kBlock, 8192)), dim3(kBlock), 0, s, d, points_2d, n, feats, (float*)nullptr, 1);
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_EHIP;
}

}  // namespace nrf

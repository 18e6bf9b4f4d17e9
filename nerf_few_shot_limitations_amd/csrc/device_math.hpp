// device_math.hpp -- per-ray / per-sample scalar maths shared by the staged
// kernels and the fused renderer.  Every function restates one reference leaf
// (file:line cited, paths relative to the reference root) with the SAME
// operation order and NO fma contraction where the reference's torch ops round
// each product separately, so that rays, depths and sample positions agree
// with the CPU path to the last bit wherever torch itself is deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrf {

struct Camera {           // ray_sampler.py:4-30
    int   H, W;
    float focal;
    float r[3][3];        // c2w[:3,:3]
    float t[3];           // c2w[:3,3]
};

// Ray r = y*W + x  ->  origin/direction.  dirs = [(x-W/2)/f, -(y-H/2)/f, -1]
// (ray_sampler.py:24); rays_d[i] = sum_j dirs[j]*R[i][j] as three rounded
// products added left to right (ray_sampler.py:27); rays_o = t (:28).
__device__ __forceinline__ void camera_ray(const Camera& c, int64_t ray, float o[3], float d[3]) {
    const int y = (int)(ray / c.W);
    const int x = (int)(ray - (int64_t)y * c.W);
    const float dx = ((float)x - (float)c.W * 0.5f) / c.focal;
    const float dy = -(((float)y - (float)c.H * 0.5f) / c.focal);
    const float dz = -1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float p0 = __fmul_rn(dx, c.r[i][0]);
        const float p1 = __fmul_rn(dy, c.r[i][1]);
        const float p2 = __fmul_rn(dz, c.r[i][2]);
        d[i] = __fadd_rn(__fadd_rn(p0, p1), p2);
        o[i] = c.t[i];
    }
}

struct DepthLadder {      // ray_utils.py:58-66
    float near, far, step;   // step = 1/(S-1) of linspace(0,1,S)
    float inv_near, inv_far; // 1/near, 1/far of the disparity form (:62)
    int   S, lindisp;
    const float* table;      // optional caller-computed ladder z_0..z_{S-1} (device); see ladder_z
};

__host__ __device__ __forceinline__ DepthLadder make_ladder(float near, float far, int S, int lindisp, const float* table) {
    DepthLadder L;
    L.near = near; L.far = far; L.S = S; L.lindisp = lindisp;
    L.step = S > 1 ? 1.0f / (float)(S - 1) : 0.0f;
    L.inv_near = 1.0f / near; L.inv_far = 1.0f / far;
    L.table = table;
    return L;
}

// A wave-uniform float computed by the VALU lives in a VGPR; loop-invariant ones get hoisted, stay live across the MLP (whose register
// budget is full) and are spilled to SCRATCH -- VMEM traffic in front of the LDS-DMA queue.  Read back through v_readfirstlane they
// live in SGPRs, whose overflow goes to VGPR lanes (v_writelane), not to memory.
__device__ __forceinline__ float uniform_f(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

__device__ __forceinline__ DepthLadder make_ladder_uniform(float near, float far, int S, int lindisp, const float* table) {
    DepthLadder L = make_ladder(near, far, S, lindisp, table);
    L.step = uniform_f(L.step); L.inv_near = uniform_f(L.inv_near); L.inv_far = uniform_f(L.inv_far);
    return L;
}

// t_s of torch.linspace(0,1,S): start + step*i below the midpoint, end - step*(S-1-i) above
// (ATen RangeFactories linspace, scalar form; its vectorised form differs from this in the last
// ulp on some CPUs -- see DESIGN.md "z ladder").
__device__ __forceinline__ float ladder_t(const DepthLadder& L, int s) {
    if (L.S == 1) return 0.0f;
    return (s < L.S / 2) ? __fmul_rn(L.step, (float)s) : __fsub_rn(1.0f, __fmul_rn(L.step, (float)(L.S - 1 - s)));
}

// un-jittered depth of sample s: near*(1-t) + far*t, or its disparity form (ray_utils.py:62,66)
// A caller-supplied table wins: torch.linspace's CPU kernel is vectorised and its last-ulp results
// depend on the host's SIMD width, and the encoding amplifies a 1-ulp depth difference by 2^(L-1), so
// the drop-in Python surface hands over the very ladder the reference would compute on that host.
__device__ __forceinline__ float ladder_z(const DepthLadder& L, int s) {
    if (L.table) return L.table[s];
    const float t = ladder_t(L, s);
    if (L.lindisp) {
        const float a = __fmul_rn(L.inv_near, __fsub_rn(1.0f, t));
        const float b = __fmul_rn(L.inv_far, t);
        return 1.0f / __fadd_rn(a, b);
    }
    return __fadd_rn(__fmul_rn(L.near, __fsub_rn(1.0f, t)), __fmul_rn(L.far, t));
}

// stratified jitter of sample s (ray_utils.py:71-79): interval [lower,upper] around z_s from the
// midpoints to its neighbours; z = lower + (upper-lower)*u.
__device__ __forceinline__ float ladder_z_jitter(const DepthLadder& L, int s, float u) {
    const float zc = ladder_z(L, s);
    const float lower = s > 0 ? __fmul_rn(0.5f, __fadd_rn(zc, ladder_z(L, s - 1))) : zc;
    const float upper = s < L.S - 1 ? __fmul_rn(0.5f, __fadd_rn(ladder_z(L, s + 1), zc)) : zc;
    return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));
}

// counter-based uniform [0,1) for the in-kernel jitter (only distribution-tested: the reference
// uses torch.rand, ray_utils.py:78, whose stream cannot be reproduced on a GPU)
__device__ __forceinline__ float counter_uniform(uint64_t seed, uint64_t ray, uint32_t s) {
    uint64_t x = seed ^ (ray * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)s << 40);
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (float)(x >> 40) * (1.0f / 16777216.0f);
}

// pts = o + d*z, product rounded before the add (ray_utils.py:82)
__device__ __forceinline__ float point_on_ray(float o, float d, float z) { return __fadd_rn(o, __fmul_rn(d, z)); }

// |d| as torch.norm(rays_d, dim=-1) (nerf_mlp.py:185)
__device__ __forceinline__ float ray_norm(const float d[3]) {
    return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
}

// Front-to-back compositing state of one ray (nerf_mlp.py:181-212): T is the exclusive running
// product of (1 - alpha + 1e-10).
struct Composite {
    float T, r, g, b, depth, acc;
    __device__ __forceinline__ void reset() { T = 1.0f; r = g = b = depth = acc = 0.0f; }
    // one sample: sigma raw (relu applied here, :193), colour c, depth z, dist = (z_next - z)*|d| or 1e10*|d| (:182-185)
    // The two halves of a step: the sample's opacity depends on nothing the ray has accumulated -- any lane may compute it -- and the
    // running state consumes it in order.  add() = add_alpha(alpha_of()): the same operations in the same order wherever they run.
    template <bool FAST>
    __device__ static __forceinline__ float alpha_of(float sigma, float dist) {
        const float x = __fmul_rn(-fmaxf(sigma, 0.0f), dist);
        const float e = FAST ? __expf(x) : expf(x);
        return __fsub_rn(1.0f, e);
    }
    template <bool FAST>
    __device__ __forceinline__ float add(float sigma, float cr, float cg, float cb, float z, float dist) {
        return add_alpha(alpha_of<FAST>(sigma, dist), cr, cg, cb, z);
    }
    __device__ __forceinline__ float add_alpha(float alpha, float cr, float cg, float cb, float z) {
        const float w = __fmul_rn(alpha, T);
        r = __fadd_rn(r, __fmul_rn(w, cr));
        g = __fadd_rn(g, __fmul_rn(w, cg));
        b = __fadd_rn(b, __fmul_rn(w, cb));
        depth = __fadd_rn(depth, __fmul_rn(w, z));
        acc = __fadd_rn(acc, w);
        T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f));
        return w;
    }
};

__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.0f + __expf(-x)); }

}  // namespace nrf

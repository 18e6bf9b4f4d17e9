// fused_impl.hpp -- the fused sample -> encode -> MLP -> composite renderer and the staged MLP
// forward, hand-written for gfx950 (CDNA4).  See mlp_core.hpp for the MFMA / weight-stream design.
//
// Work mapping of the renderer: a persistent workgroup (4 waves, one per SIMD: 64 sample columns each in the 16-bit modes, 32 in
// the fp32 / split-f16 modes -- "Workgroup geometry" below; 1 workgroup per CU because of its 142 KiB of LDS) walks ray tiles.
// Inside a wave LANE <-> RAY: the wave marches its rays front to back, so a ray's transmittance / colour / depth accumulators
// belong to ONE lane (parked in LDS between passes) and compositing never crosses lanes.  Early ray termination (ert_eps > 0)
// runs on render_queue_kernel: columns refill from a ray queue.
#pragma once
#include <atomic>
#include <cstdlib>

#include "kernels.hpp"
#include "nets.hpp"

namespace nrf {

constexpr int kBiasMaxFloats = 4096;                                   // 16 KiB bias table
constexpr int kLdsRing = kSlots * kChunkBytes;                         // 96 KiB (6 slots of 16 KiB)
constexpr int kLdsBytes = kLdsRing + kBiasMaxFloats * 4 + 64 + 4096;   // staged forward / training chain kernels: ring + bias table (+ slack)
constexpr int kLadderLds = 4096;                                        // the renderers keep the whole depth ladder in LDS (n_samples <= 4096: api.cpp:check_opts)
constexpr int kRenderThreads = 256;                                     // both renderers run 4 waves
constexpr int kLdsBytesQueue = kLdsRing + kBiasMaxFloats * 4 + 64 + kLadderLds * 4 + 14 * kRenderThreads * 4;   // + 14 floats of per-lane ray state
static_assert(kLdsBytesQueue <= 160 * 1024, "renderers: LDS over budget (use a 6-slot ring)");

struct NetArgs {
    const void* stream;
    const float* bias;
    uint32_t n_chunks;
    int n_bias;
    int n_layers;
    uint32_t ablate;
};

struct RenderKArgs {
    NetArgs net;
    RenderArgs a;
    int64_t n_tiles;
};

struct ForwardKArgs {
    NetArgs net;
    const float* x_enc;      // V1: (P, pe_dim)
    const float* pos;        // V2/V3: (P,3)
    const float* dir;        // V2/V3: (P,3)
    const float* dino;       // V3: (P,dino_dim)
    int64_t n;
    float* out4;             // V1: (P,4)
    float* rgb;              // V2/V3: (P,3)
    float* density;          // V2/V3: (P,1)
    int64_t n_tiles;
};

template <bool FAST>
__device__ __forceinline__ float sigmoid_sel(float x) { return FAST ? sigmoid_fast(x) : sigmoid_precise(x); }

// One of two register values by a (lane-dependent or uniform) flag, as a v_cndmask: left to itself LLVM turns
// `flag ? out4[1][k] : out4[0][k]` into a dynamically indexed load from a STACK copy of out4 (48 bytes of scratch per lane in the
// V2 / V3 renderers: VMEM traffic at the end of every pass, in front of the LDS-DMA queue)
__device__ __forceinline__ float pick_reg(float lo, float hi, bool take_hi) {
    asm volatile("" : "+v"(lo), "+v"(hi));
    return take_hi ? hi : lo;
}

__device__ __forceinline__ void load_bias_table(NRF_LDS float* bias, const float* src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) bias[i] = src[i];
    __syncthreads();
}

// local ray index -> (view of the batch, global ray id inside that view); pixel-tile sharding: SURVEY.md section 8e
__device__ __forceinline__ int64_t global_ray(const RenderArgs& a, int64_t i, int& cam) {
    cam = 0;
    if (!a.camera_mode) return i;
    if (a.n_cams > 1) {
        cam = (int)(i / a.rays_per_cam);
        i -= (int64_t)cam * a.rays_per_cam;
    }
    int64_t g = a.ray_begin + i;
    if (a.tile_rays < a.rays_per_cam) {
        const int64_t k = i / a.tile_rays;
        g = a.ray_begin + k * a.tile_stride + (i - k * a.tile_rays);
    }
    const int64_t last = (int64_t)a.cams[0].H * a.cams[0].W - 1;
    return g < last ? g : last;
}

// ---------------------------------------------------------------------------------------------
// fused renderer
// ---------------------------------------------------------------------------------------------
// Per-lane ray state (origin, direction, |d|, next depth, the compositor's six accumulators) is parked in LDS
// ([field][thread]: lane-linear, conflict-free) and pulled into registers only around the few instructions that use it:
// nothing per-ray is live across the MLP, whose register budget is full -- a compiler spill to scratch there is a VMEM
// op whose wait drains the LDS-DMA weight queue (the round-1 V2 / V3 builds carried ~80 spilled dwords).
enum { F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_NORM, F_Z, F_T, F_R, F_G, F_B, F_DEPTH, F_ACC, kFields };
constexpr int kLdsState = kLdsRing + kBiasMaxFloats * 4 + 64 + kLadderLds * 4;     // byte offset of the state rows

template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
__global__ void __launch_bounds__(WAVES * 64) render_kernel(const RenderKArgs P) {
    static_assert(NT == 1 || NT == 2, "a wave marches 32 or 64 sample columns");
    constexpr int COLS = 32 * NT;                  // sample columns of a wave: column q = c + 32 n sits on lanes c, c + 32 of operand tile n
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* bias = (NRF_LDS float*)(lds + kLdsRing);
    NRF_LDS float* st = (NRF_LDS float*)(lds + kLdsState);
    constexpr int nthreads = WAVES * 64;
    // ONE address register for this thread's column; fields sit at immediate offsets f*nthreads*4 (< 64 KiB, the DS offset
    // field).  Everything derived from the thread id is re-derived inside each pass from an opaque copy (tid_now): loop-invariant
    // per-lane values would otherwise be hoisted, spilled at the MLP's register peak and reloaded from scratch every pass.
    int tid_now = threadIdx.x;
    NRF_LDS float* st_me = st + tid_now;
    auto ST = [&](int f) -> NRF_LDS float& { return st_me[f * nthreads]; };
    typedef typename Mode::Act Act;
    constexpr int KT0 = pe_tiles(LP);

    int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const RenderArgs& a = P.a;
    const int S = a.n_samples;
    // The un-jittered depth ladder lives in LDS for the whole launch: the caller's table (the reference's own torch.linspace ladder) or,
    // without one, the in-kernel formula evaluated once per sample index.  Every later lookup is one ds_read: no global load in the
    // owner lane's composite loop, and none of the formula's wave-uniform constants (1/near, 1/far, the step) hoisted into VGPRs that
    // the network walk would spill.
    static_assert(WAVES * 64 <= kRenderThreads, "state rows are sized for 4 waves");
    NRF_LDS float* zl = (NRF_LDS float*)(bias + kBiasMaxFloats) + 16;
    {
        const DepthLadder lad = make_ladder(a.near, a.far, S, a.lindisp, nullptr);
        for (int i = threadIdx.x; i < S; i += blockDim.x) zl[i] = a.z_ladder ? a.z_ladder[i] : ladder_z(lad, i);
    }
    load_bias_table(bias, P.net.bias, P.net.n_bias);      // ends with __syncthreads()

    Pipe<WAVES, pinned_walk<Mode, NT>()> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, P.net.ablate);
    pipe.start();
#ifdef NRF_ABLATE_BUILD
    // timing experiment (pair with 2 = no barriers, which would re-align the waves): wave w starts w x 16 (512) or w x 64 (1024)
    // cycles late, so that the four waves no longer reach each LDS-DMA instruction in the same cycle
    if (P.net.ablate & (512 | 1024)) {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (int i = 0; i < w; ++i) {
            if (P.net.ablate & 512) asm volatile("s_nop 15"); else __builtin_amdgcn_s_sleep(1);
        }
    }
#endif

    auto z_base = [&](int s) -> float { return zl[s]; };
    // Samples per ray and MLP pass (host: pick_spw_log2).  The wave's COLS sample columns are RPW = COLS/SPW rays x SPW consecutive
    // samples: column q = ray (q mod RPW), sample (pass*SPW + q div RPW).  A ray is still composited front to back by ONE lane
    // (lane L < RPW owns ray L of the wave; its state rows are the ones of thread L), which fetches the other columns' network
    // outputs with ds_bpermute -- the per-ray sequence of operations, hence every bit of the result, depends neither on SPW nor
    // on NT.  The state rows of a lane >= RPW mirror ray (lane mod RPW): column q's origin / direction are read from thread q's rows.
    // What SPW buys: frames whose ray count does not fill whole rounds of 256-ray tiles over the CUs (400x400; an 80 000-ray
    // shard) are cut into 2x / 4x as many, shorter, work items.
    const int spw_log2 = a.spw_log2;
    const int rpw_log2 = (NT == 2 ? 6 : 5) - spw_log2;
    const int SPW = 1 << spw_log2, RPW = COLS >> spw_log2;
    const int64_t tile_rays = (int64_t)WAVES * RPW;
    const int n_pass = (S + SPW - 1) >> spw_log2;

    auto z_ray = [&](int64_t ray, int s) -> float {
        if (a.z_in) return a.z_in[ray * S + s];
        if (!a.perturb) return z_base(s);
        float u;
        if (a.t_rand) {
            u = a.t_rand[ray * S + s];
        } else {
            int ci;
            const int64_t g = global_ray(a, ray, ci);
            u = counter_uniform(a.seed + (uint64_t)ci * 0x51ED27ull, (uint64_t)g, (uint32_t)s);
        }
        // ladder_z_jitter (device_math.hpp; ray_utils.py:71-79) on the cached ladder
        const float zc = z_base(s);
        const float lower = s > 0 ? __fmul_rn(0.5f, __fadd_rn(zc, z_base(s - 1))) : zc;
        const float upper = s < S - 1 ? __fmul_rn(0.5f, __fadd_rn(z_base(s + 1), zc)) : zc;
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));
    };

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        // column q's ray (clamped: the columns past the last ray of a ragged tile repeat it and store nothing)
        auto column_ray = [&](int q) -> int64_t {
            const int64_t r = tile * tile_rays + wave * RPW + (q & (RPW - 1));
            return r < a.n_rays ? r : a.n_rays - 1;
        };
        {
            const int64_t rid = column_ray(lane);
            float o[3], d[3];
            if (a.camera_mode) {
                int ci;
                const int64_t g = global_ray(a, rid, ci);
                camera_ray(a.cams[ci], g, o, d);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { o[k] = a.rays_o[rid * 3 + k]; d[k] = a.rays_d[rid * 3 + k]; }
            }
            ST(F_OX) = o[0]; ST(F_OY) = o[1]; ST(F_OZ) = o[2];
            ST(F_DX) = d[0]; ST(F_DY) = d[1]; ST(F_DZ) = d[2];
            ST(F_NORM) = ray_norm(d);
            int s_first = 0;                         // opaque: z_0 (and the z_1 of its jitter interval) as compile-time constants of the ladder
            asm volatile("" : "+s"(s_first));        // formula were hoisted out of the tile loop into VGPRs and spilled across the network walk
            ST(F_Z) = z_ray(rid, s_first);           // depth of the ray's next sample to composite
            ST(F_T) = 1.0f; ST(F_R) = 0.0f; ST(F_G) = 0.0f; ST(F_B) = 0.0f; ST(F_DEPTH) = 0.0f; ST(F_ACC) = 0.0f;
        }

        for (int p = 0; p < n_pass; ++p) {
            tid_now = threadIdx.x;
            asm volatile("" : "+v"(tid_now));
            lane = tid_now & 63; c = lane & 31; h = lane >> 5;
            st_me = st + tid_now;

            // the state rows of column q = c + 32 n of this wave (NT == 1: the lane's own -- both lane halves mirror the ray)
            auto SQ = [&](int n, int f) -> NRF_LDS float& { return (NT == 1 ? st_me : st_me - lane + c + 32 * n)[f * nthreads]; };
            // view direction = raw rays_d (train.py:225); encoded on demand inside the network walk
            auto dirT = [&](Act (&dt)[1][NT]) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float d[3] = {SQ(n, F_DX), SQ(n, F_DY), SQ(n, F_DZ)};
                    Act t1[pe_tiles(LD)];
                    encode3<Mode, LD>(d, h, t1);
                    dt[0][n] = t1[0];
                }
            };
            // first-layer operand tiles of this step's samples (re-invoked by NetV3 for its second fusion pass: the blended feature-map
            // channels -- 16 DT fp32 registers per operand tile -- are held across the first pass, so that the gather, two rounds of
            // global loads whose latency a lone wave cannot cover, happens once per sample; the encoder's trig is recomputed)
            DinoHeld<Mode, Net::kDino ? Net::KT0 - KT0 : 1> held[NT];
            auto inputs = [&](const float (&w0)[NT], const float (&w1)[NT], Act (&x)[Net::KT0][NT], auto pass_) {
                constexpr int PASS = decltype(pass_)::value;
#ifdef NRF_ABLATE_BUILD
                if (P.net.ablate & 16) {      // timing experiment: no encoder (and no compositor below)
                    // hashed bit patterns in bf16 [0.008, 2): realistic toggling (all-zero operands let the clock rise) at ~1/6 of the encoder's VALU work
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int t = 0; t < Net::KT0; ++t) {
                            i32x4 w0v, w1v;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const unsigned hsh = (unsigned)(tid_now * 2654435761u) + (unsigned)(p * 40503 + (8 * t + q + 4 * n) * 7919);
                                w0v[q] = (int)(((hsh * 2246822519u) & 0x3FFF3FFFu) | 0x3C003C00u) ^ ((hsh & 1u) << 31);
                                w1v[q] = (int)(((hsh * 3266489917u) & 0x3FFF3FFFu) | 0x3C003C00u) ^ ((hsh & 2u) << 14);
                            }
                            Act e = {};
                            __builtin_memcpy(&e, &w0v, 16);
                            __builtin_memcpy((char*)&e + 16, &w1v, 16);
                            x[t][n] = e;
                        }
                    return;
                }
#endif
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    // depth of this column's sample (columns past the last sample repeat it; their outputs are not used)
                    float zc;
                    if (SPW == 1) {
                        zc = SQ(n, F_Z);
                    } else {
                        const int q = c + 32 * n;
                        const int sc = p * SPW + (q >> rpw_log2);
                        zc = z_ray(column_ray(q), sc < S ? sc : S - 1);
                    }
                    float pt[3];
                    pt[0] = point_on_ray(SQ(n, F_OX), SQ(n, F_DX), zc);
                    pt[1] = point_on_ray(SQ(n, F_OY), SQ(n, F_DY), zc);
                    pt[2] = point_on_ray(SQ(n, F_OZ), SQ(n, F_DZ), zc);
                    // V3, first pass: start the feature-map gather of this column BEFORE its positional encoding, blend behind it
                    DinoTaps tp;
                    DinoRaw<Net::kDino ? Net::KT0 - KT0 : 1> raw;
                    bool split_gather = false;
                    if constexpr (Net::kDino && PASS == 0) {
#ifdef NRF_ABLATE_BUILD
                        if (!(P.net.ablate & (32 | 256)))
#endif
                        {
                            tp = dino_taps(a.dino, pt);
                            raw.issue(a.dino.features, tp, h);
                            split_gather = true;
                        }
                    }
                    Act e1[KT0];
#ifdef NRF_ABLATE_BUILD
                    // timing experiments on V3 (results are wrong): 64 = the second fusion pass reuses un-gated first-layer tiles of the
                    // sample position instead of re-encoding; 32 = no feature-map gather (a constant map, no global loads)
                    if (PASS == 1 && (P.net.ablate & 64)) {
                        Act c1[pe_tiles(1)];
                        encode3<Mode, 1>(pt, h, c1, w0[n]);
#pragma unroll
                        for (int t = 0; t < KT0; ++t) e1[t] = c1[0];
                    } else
#endif
                    encode3<Mode, LP>(pt, h, e1, w0[n]);
#pragma unroll
                    for (int t = 0; t < KT0; ++t) x[t][n] = e1[t];
                    if constexpr (Net::kDino) {
                        constexpr int DT = Net::KT0 - KT0;
#ifdef NRF_ABLATE_BUILD
                        if (PASS == 0 && (P.net.ablate & 32)) {
                            DinoTaps tp;
#pragma unroll
                            for (int k = 0; k < 4; ++k) { tp.off[k] = 0; tp.w[k] = 0.25f; }
                            if (P.net.ablate & 128) held[n].gather(a.dino.features, tp, h);      // 128: keep the loads, one L2-hot address
                            else { tp.off[0] = tp.off[1] = tp.off[2] = tp.off[3] = -1; held[n].gather(a.dino.features, tp, h); }
                        } else
#endif
                        if constexpr (PASS == 0) {
                            if (split_gather) held[n].finish(raw, tp);
                            else held[n].gather(a.dino.features, dino_taps(a.dino, pt), h);      // (ablation 256: the round-2 order)
                        }
                        Act dt[DT];
                        held[n].template tiles<PASS>(w1[n], dt);
#pragma unroll
                        for (int t = 0; t < DT; ++t) x[KT0 + t][n] = dt[t];
                    }
                }
            };

            float out4[NT][4];
            Net::eval(pipe, bias, h, P.net.n_layers, inputs, dirT, out4);

            // ---- composite this pass's SPW samples of the lane's ray, front to back --------------------------------------
            // (the lane id once more from an opaque copy: what the compositor derives from it -- 64-bit ray ids, clamps, flags -- would
            // otherwise be computed before the network walk and carried, i.e. spilled, across it)
            tid_now = threadIdx.x;
            asm volatile("" : "+v"(tid_now));
            lane = tid_now & 63; c = lane & 31; h = lane >> 5;
            st_me = st + tid_now;
            const int64_t rid = column_ray(lane);
            const int64_t raw = tile * tile_rays + wave * RPW + (lane & (RPW - 1));
            const bool own_valid = lane < RPW && raw < a.n_rays;
            Composite comp;
            comp.T = ST(F_T); comp.r = ST(F_R); comp.g = ST(F_G); comp.b = ST(F_B); comp.depth = ST(F_DEPTH); comp.acc = ST(F_ACC);
            float zo = ST(F_Z);
            const float norm = ST(F_NORM);
            if (SPW == 1) {
                const int s = p;
#ifdef NRF_ABLATE_BUILD
                if (P.net.ablate & 16) { comp.r += out4[0][0] + out4[NT - 1][3]; } else
#endif
                {
                    const bool last = (s + 1 == S);
                    float v[4];
                    // NT == 2: lane L owns column L = tile h, column c -- and holds that tile's head rows itself
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = NT == 2 ? pick_reg(out4[0][k], out4[NT - 1][k], h != 0) : out4[0][k];
                    const float zn = last ? 0.0f : z_ray(rid, s + 1);
                    const float dist = last ? __fmul_rn(1e10f, norm) : __fmul_rn(__fsub_rn(zn, zo), norm);
                    const float w = comp.template add<Mode::FAST_EXP>(v[3], sigmoid_sel<Mode::FAST_EXP>(v[0]), sigmoid_sel<Mode::FAST_EXP>(v[1]),
                                                                      sigmoid_sel<Mode::FAST_EXP>(v[2]), zo, dist);
                    if (own_valid) {
                        if (a.weights) a.weights[rid * S + s] = w;
                        if (a.z_vals) a.z_vals[rid * S + s] = zo;
                    }
                    zo = zn;
                }
            } else {
                // SPW > 1.  Column phase, every lane in parallel: lane q holds the head rows of column q (NT == 2: tile q div 32; NT == 1:
                // both lane halves hold column c) and its state rows mirror that column's ray -- it turns its sample's network outputs
                // into (alpha, r, g, b, z): the transcendental part of the step (exp, three sigmoids: ~150 dependent VALU instructions in the
                // precise modes), which needs nothing of what the ray has accumulated.  Owner phase: lane L < RPW consumes its ray's SPW
                // samples in order, five ds_bpermute + the six-operation state update per sample.  Same operations on the same values in
                // the same order as the SPW = 1 march -- only the lane that runs the first half differs.  (Serial in the owner lane, the
                // whole step cost ~1500 cycles per sample: 20 us per pass at SPW = 32, 40 % on top of the MLP: profiles/r03_small_frames.txt.)
                const int q = NT == 2 ? lane : c;
                const int sq = p * SPW + (q >> rpw_log2);
                float ca = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, cz = 0.0f;
                if (sq < S) {
                    const bool last = (sq + 1 == S);
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = NT == 2 ? pick_reg(out4[0][k], out4[NT - 1][k], h != 0) : out4[0][k];
                    cz = z_ray(rid, sq);
                    const float zn = last ? 0.0f : z_ray(rid, sq + 1);
                    const float dist = last ? __fmul_rn(1e10f, norm) : __fmul_rn(__fsub_rn(zn, cz), norm);
                    ca = Composite::alpha_of<Mode::FAST_EXP>(v[3], dist);
                    cr = sigmoid_sel<Mode::FAST_EXP>(v[0]); cg = sigmoid_sel<Mode::FAST_EXP>(v[1]); cb = sigmoid_sel<Mode::FAST_EXP>(v[2]);
                }
                for (int j = 0; j < SPW; ++j) {
                    const int s = p * SPW + j;
                    if (s >= S) break;
                    const int qj = lane + (j << rpw_log2);                       // the column of sample s of this (owner) lane's ray
                    const int src = (NT == 2 ? (qj & 63) : ((lane & 32) | (qj & 31))) << 2;
                    auto from = [&](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x))); };
                    const float aj = from(ca), rj = from(cr), gj = from(cg), bj = from(cb), zj = from(cz);
                    const float w = comp.add_alpha(aj, rj, gj, bj, zj);
                    if (own_valid) {
                        if (a.weights) a.weights[rid * S + s] = w;
                        if (a.z_vals) a.z_vals[rid * S + s] = zj;
                    }
                }
            }
            ST(F_T) = comp.T; ST(F_R) = comp.r; ST(F_G) = comp.g; ST(F_B) = comp.b; ST(F_DEPTH) = comp.depth; ST(F_ACC) = comp.acc;
            ST(F_Z) = zo;
        }

        {
            const int64_t raw = tile * tile_rays + wave * RPW + (lane & (RPW - 1));
            if (lane < RPW && raw < a.n_rays) {
                float r = ST(F_R), g = ST(F_G), b = ST(F_B);
                if (a.white_bkgd) {                                      // nerf_mlp.py:209-212
                    const float bg = __fsub_rn(1.0f, ST(F_ACC));
                    r = __fadd_rn(r, bg); g = __fadd_rn(g, bg); b = __fadd_rn(b, bg);
                }
                if (a.interleaved) {
                    *(float4*)(a.rgb + raw * 4) = make_float4(r, g, b, ST(F_DEPTH));     // (R,4) rows [r,g,b,depth]: the gather buffer
                } else {
                    a.rgb[raw * 3 + 0] = r;
                    a.rgb[raw * 3 + 1] = g;
                    a.rgb[raw * 3 + 2] = b;
                    a.depth[raw] = ST(F_DEPTH);
                }
            }
        }
    }
    pipe.drain();
}

// ---------------------------------------------------------------------------------------------
// fused renderer with per-ray early termination: persistent lanes fed from a ray queue
// ---------------------------------------------------------------------------------------------
// When ert_eps > 0 the marching is no longer tile-synchronous.  Every sample column of a wave holds ONE ray and its own
// sample index; each MLP pass advances every live ray by one sample; a ray that has used its S samples or whose
// transmittance fell below ert_eps is written out and its column takes the next ray of the wave's strip.  Strips
// (kStrip * NT consecutive ray ids) come from one device-wide atomic counter, so workgroups drain the frame together.
// NT == 1 (32 columns): both lanes of a pair (c, c+32) hold the column's ray and run the identical compositor (the head tile
// carries [r,g,b,sigma] for both halves), so they agree on termination without exchanging anything; only the low lane stores.
// NT == 2 (64 columns): lane L owns column L = operand tile L div 32, column L mod 32.  Per-ray arithmetic is exactly that of
// render_kernel, so with ert_eps -> 0 the image is the same; with ert_eps > 0 each ray stops at ITS OWN T < eps.
constexpr int kStrip = 32;        // rays handed out per atomic and 32 columns: one wave-load, so that the frame's last strips spread over all waves (4 wave-loads left a 4-batch tail: +18 %)

template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
__global__ void __launch_bounds__(WAVES * 64) render_queue_kernel(const RenderKArgs P) {
    static_assert(NT == 1 || NT == 2, "a wave marches 32 or 64 sample columns");
    constexpr int STRIP = kStrip * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* bias = (NRF_LDS float*)(lds + kLdsRing);
    NRF_LDS int* flags = (NRF_LDS int*)(bias + kBiasMaxFloats);
    NRF_LDS float* zl = (NRF_LDS float*)(flags + 16);
    // Per-lane ray state lives in LDS ([field][thread]: lane-linear, conflict-free) and is only pulled into registers
    // around the few instructions that use it: nothing per-ray is live across the MLP, whose register budget is full
    // (a compiler spill to scratch there is a VMEM op whose wait drains the LDS-DMA weight queue).
    NRF_LDS float* st = zl + kLadderLds;                  // == lds + kLdsState
    constexpr int nthreads = WAVES * 64;
    // ONE address register for this thread's column; fields sit at immediate offsets f*nthreads*4 (< 64 KiB, the DS
    // offset field).  The empty asm keeps the compiler from folding the (> 64 KiB) region base into 14 separate
    // per-field address registers, which it then spilled to scratch -- every reload of those drained the LDS-DMA queue.
    // Everything derived from the thread id is re-derived inside each pass from an opaque copy (tid_now): loop-invariant
    // per-lane values would otherwise be hoisted, spilled at the MLP's register peak and reloaded from scratch every pass.
    int tid_now = threadIdx.x;
    NRF_LDS float* st_me = st + tid_now;
    auto ST = [&](int f) -> NRF_LDS float& { return st_me[f * nthreads]; };
    typedef typename Mode::Act Act;
    constexpr int KT0 = pe_tiles(LP);

    int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const RenderArgs& a = P.a;
    const int S = a.n_samples;
    static_assert(WAVES * 64 <= kRenderThreads, "state rows are sized for 4 waves");
    {   // the depth ladder in LDS: the caller's table or the in-kernel formula, once per sample index (render_kernel)
        const DepthLadder lad = make_ladder(a.near, a.far, S, a.lindisp, nullptr);
        for (int i = threadIdx.x; i < S; i += blockDim.x) zl[i] = a.z_ladder ? a.z_ladder[i] : ladder_z(lad, i);
    }
    ST(F_OX) = 0.f; ST(F_OY) = 0.f; ST(F_OZ) = 0.f; ST(F_DX) = 0.f; ST(F_DY) = 0.f; ST(F_DZ) = -1.f; ST(F_Z) = 1.f;
    load_bias_table(bias, P.net.bias, P.net.n_bias);      // ends with __syncthreads()

    Pipe<WAVES, pinned_walk<Mode, NT>()> pipe;     // a wave that has run dry keeps computing (on stale inputs, storing nothing): the workgroup moves in lockstep anyway, and the skip paths cost registers in every layer
    pipe.init(P.net.stream, P.net.n_chunks, lds, P.net.ablate);
    pipe.start();

    auto z_base = [&](int s) -> float { return zl[s]; };
    auto z_of = [&](int64_t ray, int s) -> float {
        if (a.z_in) return a.z_in[ray * S + s];
        if (!a.perturb) return z_base(s);
        float u;
        if (a.t_rand) {
            u = a.t_rand[ray * S + s];
        } else {
            int ci;
            const int64_t g = global_ray(a, ray, ci);
            u = counter_uniform(a.seed + (uint64_t)ci * 0x51ED27ull, (uint64_t)g, (uint32_t)s);
        }
        const float zc = z_base(s);
        const float lower = s > 0 ? __fmul_rn(0.5f, __fadd_rn(zc, z_base(s - 1))) : zc;
        const float upper = s < S - 1 ? __fmul_rn(0.5f, __fadd_rn(z_base(s + 1), zc)) : zc;
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));
    };

    // registers: only the ray id and its sample index (NT == 1: identical in both lanes of a pair)
    int ray = -1;
    int s = 0;
    // wave-uniform queue state
    int64_t pool_next = 0, pool_end = 0;
    bool exhausted = false;

    for (int pass = 0;; ++pass) {
        tid_now = threadIdx.x;
        asm volatile("" : "+v"(tid_now));
        lane = tid_now & 63; c = lane & 31; h = lane >> 5;
        st_me = st + tid_now;
        // ---- hand new rays to idle lane pairs -------------------------------------------------
        if (!pipe.skip) {
            const bool need = ray < 0;
            const uint64_t m = NT == 2 ? (uint64_t)__ballot(need) : (uint64_t)(__ballot(need) & 0xFFFFFFFFull);   // NT == 1: pairs are identical, the low half suffices
            const int cnt = __builtin_popcountll(m);
            if (cnt > 0 && !exhausted) {
                if (pool_next == pool_end) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(a.queue, (unsigned long long)STRIP);
                    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)base);
                    if ((int64_t)base >= a.n_rays) {
                        exhausted = true;
                    } else {
                        pool_next = (int64_t)base;
                        pool_end = (int64_t)base + STRIP < a.n_rays ? (int64_t)base + STRIP : a.n_rays;
                    }
                }
                if (pool_next < pool_end) {
                    const int64_t idx = pool_next + __builtin_popcountll(m & ((1ull << (NT == 2 ? lane : c)) - 1ull));
                    if (need && idx < pool_end) {
                        ray = (int)idx;
                        s = 0;
                        float o[3], d[3];
                        if (a.camera_mode) {
                            int ci;
                            const int64_t g = global_ray(a, idx, ci);
                            camera_ray(a.cams[ci], g, o, d);
                        } else {
#pragma unroll
                            for (int k = 0; k < 3; ++k) { o[k] = a.rays_o[idx * 3 + k]; d[k] = a.rays_d[idx * 3 + k]; }
                        }
                        ST(F_OX) = o[0]; ST(F_OY) = o[1]; ST(F_OZ) = o[2];
                        ST(F_DX) = d[0]; ST(F_DY) = d[1]; ST(F_DZ) = d[2];
                        ST(F_NORM) = ray_norm(d);
                        ST(F_Z) = z_of(idx, 0);
                        ST(F_T) = 1.0f; ST(F_R) = 0.0f; ST(F_G) = 0.0f; ST(F_B) = 0.0f; ST(F_DEPTH) = 0.0f; ST(F_ACC) = 0.0f;
                    }
                    pool_next = pool_next + cnt < pool_end ? pool_next + cnt : pool_end;
                }
            }
            // nothing left to do for this wave: keep the stream protocol, skip the math
            const int run_dry = (exhausted && pool_next == pool_end && !__any(ray >= 0)) ? 1 : 0;
            pipe.skip = (uint32_t)__builtin_amdgcn_readfirstlane(run_dry);     // provably wave-uniform: scalar branches only
        }

        // ---- one sample per live ray ----------------------------------------------------------
        // the state rows of column q = c + 32 n of this wave (NT == 1: the lane's own)
        auto SQ = [&](int n, int f) -> NRF_LDS float& { return (NT == 1 ? st_me : st_me - lane + c + 32 * n)[f * nthreads]; };
        auto dirT = [&](Act (&dt)[1][NT]) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float d[3] = {SQ(n, F_DX), SQ(n, F_DY), SQ(n, F_DZ)};
                Act t1[pe_tiles(LD)];
                encode3<Mode, LD>(d, h, t1);
                dt[0][n] = t1[0];
            }
        };
        DinoHeld<Mode, Net::kDino ? Net::KT0 - KT0 : 1> held[NT];        // render_kernel: the gathered channels are held across NetV3's first fusion pass
        auto inputs = [&](const float (&w0)[NT], const float (&w1)[NT], Act (&x)[Net::KT0][NT], auto pass_) {
            // (a wave that has run dry encodes its stale -- valid -- state like any other: an early return here made every operand
            // tile and the held channels values merged across a branch, 300 spilled registers in the V3 build)
            constexpr int PASS = decltype(pass_)::value;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float zc = SQ(n, F_Z);
                float p[3];
                p[0] = point_on_ray(SQ(n, F_OX), SQ(n, F_DX), zc);
                p[1] = point_on_ray(SQ(n, F_OY), SQ(n, F_DY), zc);
                p[2] = point_on_ray(SQ(n, F_OZ), SQ(n, F_DZ), zc);
                DinoTaps tp;
                DinoRaw<Net::kDino ? Net::KT0 - KT0 : 1> raw;
                if constexpr (Net::kDino && PASS == 0) {          // the gather starts before the encoding and is blended behind it (render_kernel)
                    tp = dino_taps(a.dino, p);
                    raw.issue(a.dino.features, tp, h);
                }
                Act e1[KT0];
                encode3<Mode, LP>(p, h, e1, w0[n]);
#pragma unroll
                for (int t = 0; t < KT0; ++t) x[t][n] = e1[t];
                if constexpr (Net::kDino) {
                    constexpr int DT = Net::KT0 - KT0;
                    if constexpr (PASS == 0) held[n].finish(raw, tp);
                    Act dt[DT];
                    held[n].template tiles<PASS>(w1[n], dt);
#pragma unroll
                    for (int t = 0; t < DT; ++t) x[KT0 + t][n] = dt[t];
                }
            }
        };
        float out4[NT][4];
        Net::eval(pipe, bias, h, P.net.n_layers, inputs, dirT, out4);

        if (!pipe.skip && ray >= 0) {
            const bool last = (s + 1 == S);
            const float zc = ST(F_Z);
            const float norm = ST(F_NORM);
            const float zn = last ? 0.0f : z_of(ray, s + 1);
            const float dist = last ? __fmul_rn(1e10f, norm) : __fmul_rn(__fsub_rn(zn, zc), norm);
            Composite comp;
            comp.T = ST(F_T); comp.r = ST(F_R); comp.g = ST(F_G); comp.b = ST(F_B); comp.depth = ST(F_DEPTH); comp.acc = ST(F_ACC);
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = NT == 2 ? pick_reg(out4[0][k], out4[NT - 1][k], h != 0) : out4[0][k];
            const float w = comp.template add<Mode::FAST_EXP>(v[3], sigmoid_sel<Mode::FAST_EXP>(v[0]),
                                                              sigmoid_sel<Mode::FAST_EXP>(v[1]), sigmoid_sel<Mode::FAST_EXP>(v[2]), zc, dist);
            const int64_t rr = ray;
            const bool stores = NT == 2 || h == 0;
            if (stores) {
                if (a.weights) a.weights[rr * S + s] = w;
                if (a.z_vals) a.z_vals[rr * S + s] = zc;
            }
            const bool fin = last || comp.T < a.ert_eps;
            if (fin) {
                if (stores) {
                    // samples skipped by early termination carry weight < ert_eps: report 0 and their depths
                    for (int s2 = s + 1; s2 < S; ++s2) {
                        if (a.weights) a.weights[rr * S + s2] = 0.0f;
                        if (a.z_vals) a.z_vals[rr * S + s2] = z_of(rr, s2);
                    }
                    float r = comp.r, g = comp.g, b = comp.b;
                    if (a.white_bkgd) {
                        const float bg = __fsub_rn(1.0f, comp.acc);
                        r = __fadd_rn(r, bg); g = __fadd_rn(g, bg); b = __fadd_rn(b, bg);
                    }
                    if (a.interleaved) {
                        *(float4*)(a.rgb + rr * 4) = make_float4(r, g, b, comp.depth);
                    } else {
                        a.rgb[rr * 3 + 0] = r;
                        a.rgb[rr * 3 + 1] = g;
                        a.rgb[rr * 3 + 2] = b;
                        a.depth[rr] = comp.depth;
                    }
                }
                ray = -1;
            } else {
                ST(F_T) = comp.T; ST(F_R) = comp.r; ST(F_G) = comp.g; ST(F_B) = comp.b; ST(F_DEPTH) = comp.depth; ST(F_ACC) = comp.acc;
                ST(F_Z) = zn;
                ++s;
            }
        }

        // ---- workgroup-wide vote every fourth pass: leave once every wave has run dry -----------
        if ((pass & 3) == 3) {
            if (lane == 0) flags[((pass >> 2) & 1) * WAVES + wave] = (int)pipe.skip;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const NRF_LDS int* fl = flags + ((pass >> 2) & 1) * WAVES;
            int all_done = 1;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) all_done &= fl[wv];
            if (all_done) break;
        }
    }
    pipe.drain();
}

// ---------------------------------------------------------------------------------------------
// staged MLP forward on explicit per-sample inputs (NeRFMLP.forward drop-in)
// ---------------------------------------------------------------------------------------------
template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
__global__ void __launch_bounds__(WAVES * 64) forward_kernel(const ForwardKArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NRF_LDS char* lds = (NRF_LDS char*)smem;
    NRF_LDS float* bias = (NRF_LDS float*)(lds + kLdsRing);
    typedef typename Mode::Act Act;
    constexpr int KT0 = pe_tiles(LP);
    constexpr int TILE = WAVES * 32 * NT;
    constexpr int PE = pe_dim(LP);

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    load_bias_table(bias, P.net.bias, P.net.n_bias);
    Pipe<WAVES, pinned_walk<Mode, NT>()> pipe;
    pipe.init(P.net.stream, P.net.n_chunks, lds, P.net.ablate);
    pipe.start();
    const int own = (NT == 2) ? h : 0;
    const bool owner = h < NT;

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        int64_t sid[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int64_t r = tile * TILE + wave * (32 * NT) + 32 * n + c;
            sid[n] = r < P.n ? r : P.n - 1;
        }
        auto dirT = [&](Act (&dt)[1][NT]) {
            if constexpr (Net::kNeedsDir) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float dd[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) dd[k] = P.dir[sid[n] * 3 + k];
                    Act t1[pe_tiles(LD)];
                    encode3<Mode, LD>(dd, h, t1);
                    dt[0][n] = t1[0];
                }
            }
        };
        auto inputs = [&](const float (&w0)[NT], const float (&w1)[NT], Act (&x)[Net::KT0][NT], auto) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if constexpr (Net::kNeedsDir) {
                    float p[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) p[k] = P.pos[sid[n] * 3 + k];
                    Act e1[KT0];
                    encode3<Mode, LP>(p, h, e1, w0[n]);
#pragma unroll
                    for (int t = 0; t < KT0; ++t) x[t][n] = e1[t];
                    if constexpr (Net::kDino) {
                        // features handed over per sample (NeRFMLP.forward's third argument): channel 32t+8g+4h+q
                        constexpr int DT = Net::KT0 - KT0;
                        const float* f = P.dino + sid[n] * (32 * DT);
#pragma unroll
                        for (int t = 0; t < DT; ++t) {
                            f32x16 e;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 v = *(const f32x4*)(f + 32 * t + 8 * g + 4 * h);
#pragma unroll
                                for (int q = 0; q < 4; ++q) e[4 * g + q] = v[q] * w1[n];
                            }
                            x[KT0 + t][n] = Mode::template to_act<false>(e);
                        }
                    }
                } else {
                    // gather the already-encoded features into operand order (feature_map.hpp)
                    const float* xin = P.x_enc + sid[n] * PE;
                    f32x16 e[KT0];
                    static_for<16 * KT0>([&](auto u_) {
                        constexpr int u = decltype(u_)::value;
                        constexpr int i0 = pe_ref_index(LP, u, 0), i1 = pe_ref_index(LP, u, 1);
                        float val = 0.0f;
                        if constexpr (i0 >= 0 && i1 >= 0) val = xin[h ? i1 : i0];
                        else if constexpr (i0 >= 0) val = h ? 0.0f : xin[i0];
                        else if constexpr (i1 >= 0) val = h ? xin[i1] : 0.0f;
                        e[u / 16][u % 16] = val;
                    });
#pragma unroll
                    for (int t = 0; t < KT0; ++t) x[t][n] = Mode::template to_act<false>(e[t]);
                }
            }
        };
        float out4[NT][4];
        Net::eval(pipe, bias, h, P.net.n_layers, inputs, dirT, out4);
        const int64_t own_raw = tile * TILE + wave * (32 * NT) + 32 * own + c;
        if (owner && own_raw < P.n) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = NT == 2 ? pick_reg(out4[0][k], out4[NT - 1][k], own != 0) : out4[0][k];
            const float r = sigmoid_sel<Mode::FAST_EXP>(v[0]), g = sigmoid_sel<Mode::FAST_EXP>(v[1]), b = sigmoid_sel<Mode::FAST_EXP>(v[2]);
            if constexpr (Net::kNeedsDir) {
                P.rgb[own_raw * 3 + 0] = r; P.rgb[own_raw * 3 + 1] = g; P.rgb[own_raw * 3 + 2] = b;
                P.density[own_raw] = fmaxf(v[3], 0.0f);                 // nerf_mlp.py:63
            } else {
                *(float4*)(P.out4 + own_raw * 4) = make_float4(r, g, b, v[3]);   // nerf_model.py:22-24
            }
        }
    }
    pipe.drain();
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
namespace {

// the >64 KiB dynamic-LDS opt-in is a per-device function attribute: set it once per (kernel, device)
template <class K>
int prepare(K kernel, int device, unsigned char (&done)[64], std::string& err, int lds_bytes = kLdsBytes) {
    if (device >= 0 && device < 64 && done[device]) return NRF_OK;
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) { err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return NRF_EHIP; }
    if (device >= 0 && device < 64) done[device] = 1;
    return NRF_OK;
}

NetArgs net_args(const DeviceNet& net, int mode) {
    NetArgs n;
    n.stream = net.stream[mode]; n.bias = net.bias; n.n_chunks = net.n_chunks[mode]; n.n_bias = net.n_bias; n.n_layers = net.arch.n_layers;
    static const uint32_t ablate = [] { const char* e = getenv("NRF_ABLATE"); return e ? (uint32_t)atoi(e) : 0u; }();
    n.ablate = ablate;   // timing experiments only: results are wrong when set
    return n;
}

// Samples per ray and pass (log2; render_kernel): the work items of a launch are tiles of WAVES * COLS/SPW rays marched in
// ceil(S/SPW) passes, dealt to the CUs in whole rounds -- pick the SPW = 1, 2, 4 ... COLS (a wave's columns: ONE ray per wave at the
// far end) with the least rounds x passes.  What a wider split costs is the owner lane's serial composite of its SPW samples
// per pass (~0.15 % of an MLP pass per sample: measured, profiles/r03_small_frames.txt); ties go to the smaller SPW.  Small frames live
// off the far end: 100 x 100 x 32 (BASELINE config 1) is 157 tiles x 8 passes at SPW = 4 -- one round, 61 % of the CUs -- and
// 1250 tiles x 1 pass = 5 rounds at SPW = 32; a 64 x 64 x 48 validation frame drops from 12 pass-times to 3.  Every choice is exact:
// a ray's sequence of operations does not depend on it.  NRF_SPW=0..6 pins it (A/B runs).
inline int pick_spw_log2(int64_t n_rays, int S, int waves, int cols_per_wave, int cu) {
    const char* env = getenv("NRF_SPW");                 // read per launch: tests walk through every split inside one process
    const int pinned = (env && *env) ? atoi(env) : -1;
    int max_l = 0;
    while ((2 << max_l) <= cols_per_wave) ++max_l;
    if (pinned >= 0) return pinned < max_l ? pinned : max_l;
    int best = 0;
    double best_t = 0.0;
    for (int l = 0; l <= max_l; ++l) {
        const int64_t tile = (int64_t)waves * (cols_per_wave >> l);
        const int64_t tiles = (n_rays + tile - 1) / tile;
        const int64_t rounds = (tiles + cu - 1) / cu;
        const double t = (double)rounds * (double)((S + (1 << l) - 1) >> l) * (1.0 + 0.0015 * (double)(1 << l));
        if (l == 0 || t < best_t * (1.0 - 1e-9)) { best_t = t; best = l; }
    }
    return best;
}

template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
int run_render_v(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) {
    auto kernel = render_kernel<Net, Mode, NT, WAVES, LP, LD>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err, kLdsBytesQueue);
    if (prepared != NRF_OK) return prepared;
    RenderKArgs k;
    k.net = net_args(net, mode);
    k.a = a;
    k.a.spw_log2 = pick_spw_log2(a.n_rays, a.n_samples, WAVES, 32 * NT, net.cu_count);
    const int64_t tile = (int64_t)WAVES * ((32 * NT) >> k.a.spw_log2);
    k.n_tiles = (a.n_rays + tile - 1) / tile;
    const int64_t grid = k.n_tiles < net.cu_count ? k.n_tiles : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytesQueue, s, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("render launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
int run_render_queue(const DeviceNet& net, int mode, RenderArgs a, hipStream_t s, std::string& err) {
    auto kernel = render_queue_kernel<Net, Mode, NT, WAVES, LP, LD>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err, kLdsBytesQueue);
    if (prepared != NRF_OK) return prepared;
    static std::atomic<unsigned> turn{0};
    a.queue = net.queues + (turn.fetch_add(1) % kQueueSlots);
    hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) { err = std::string("queue reset: ") + hipGetErrorString(e); return NRF_EHIP; }
    RenderKArgs k;
    k.net = net_args(net, mode);
    k.a = a;
    k.n_tiles = 0;
    const int64_t strips = (a.n_rays + kStrip * NT - 1) / (kStrip * NT);
    const int64_t blocks = (strips + WAVES - 1) / WAVES;
    const int64_t grid = blocks < net.cu_count ? blocks : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytesQueue, s, k);
    e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("render (ray queue) launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
int run_render(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) {
    if (a.ert_eps > 0.0f) {
        // per-ray early termination: the ray-queue kernel (one ray per lane pair, refilled from a device-wide queue)
        if (!net.queues || a.n_rays >= (int64_t)1 << 31) { err = "early ray termination: launch too large (>= 2^31 rays)"; return NRF_EINVAL; }
        return run_render_queue<Net, Mode, NT, WAVES, LP, LD>(net, mode, a, s, err);
    }
    return run_render_v<Net, Mode, NT, WAVES, LP, LD>(net, mode, a, s, err);
}

template <class Net, class Mode, int NT, int WAVES, int LP, int LD>
int run_forward(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) {
    auto kernel = forward_kernel<Net, Mode, NT, WAVES, LP, LD>;
    static unsigned char done[64] = {};
    const int prepared = prepare(kernel, net.device, done, err);
    if (prepared != NRF_OK) return prepared;
    k.net = net_args(net, mode);
    constexpr int TILE = WAVES * 32 * NT;
    k.n_tiles = (k.n + TILE - 1) / TILE;
    const int64_t grid = k.n_tiles < net.cu_count ? k.n_tiles : net.cu_count;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WAVES * 64), kLdsBytes, s, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("forward launch: ") + hipGetErrorString(e); return NRF_EHIP; }
    return NRF_OK;
}

bool check_net(const DeviceNet& net, int mode, std::string& err) {
    if (mode < 0 || mode >= kModes) { err = "unknown mma_mode"; return false; }
    if (net.n_bias > kBiasMaxFloats) { err = "bias table exceeds the LDS carve-out"; return false; }
    if (net.arch.dir_freq != 4 && net.arch.net != NRF_NET_V1) { err = "only dir_freq=4 is built"; return false; }
    return true;
}

}  // namespace

// Workgroup geometry per arithmetic mode and network family (NT sample tiles per wave, WAVES).  Every geometry marches 256
// sample columns per workgroup and weight pass.
//   * 16-bit modes: 4 waves x 64 columns, one wave per SIMD on the whole 512-entry register file, pinned walk.  A fragment
//     read from LDS feeds two MFMAs (half the LDS bytes per FLOP), finished operand images are parked in the AGPR half
//     (mlp_core.hpp NRF_PARK_ACT) and nothing of the walk spills (the 8 x 32 builds of V2 / V3 carried 40-50 spilled registers).
//     Against 8 x 32 on the same box: V2 +8 %, V1 +1...2 % as first built (profiles/r02_ab_wide16.txt), then +3.6 % (chained
//     layers), +2.7 % (AGPR operands), +0.8 % (spread LDS-DMA issue); V3 +3 % (its per-pass feature-map gather -- global loads,
//     twice per pass -- is latency one wave per SIMD cannot cover: profiles/r02_ab_v3_geometry.txt).
//   * fp32 and split-f16: 4 waves x 32 columns (their activations take 16 registers per tile).
// The images are bit-identical across geometries (tools/image_hash.py): a column's arithmetic does not depend on where it sits.
// Each family's translation unit is compiled twice (build.py): NRF_TU_HALF == 16 holds the two 16-bit modes (VGPR-form MFMAs,
// operand images parked in AGPRs), NRF_TU_HALF == 32 the fp32-class modes; fused_kernels.hip routes by mode.
#if !defined(NRF_TU_HALF)
// (fused_kernels.hip: routing only, no kernels)
#elif NRF_TU_HALF == 16
#define NRF_TU_NAME(f) f##_16
#define NRF_DISPATCH_MODE(FN, NET, LP, ...)                                                                 \
    switch (mode) {                                                                                         \
        case NRF_MMA_BF16: return FN<NET<ModeBF16, 2, LP>, ModeBF16, 2, 4, LP, 4>(__VA_ARGS__);             \
        default:           return FN<NET<ModeF16, 2, LP>, ModeF16, 2, 4, LP, 4>(__VA_ARGS__);               \
    }
// V2 / V3 (NT16, WAVES16: the 16-bit modes' geometry)
#define NRF_DISPATCH_MODE1(FN, NETT, LP, NT16, WAVES16, ...)                                                \
    switch (mode) {                                                                                         \
        case NRF_MMA_BF16: return FN<NETT(ModeBF16, NT16), ModeBF16, NT16, WAVES16, LP, 4>(__VA_ARGS__);    \
        default:           return FN<NETT(ModeF16, NT16), ModeF16, NT16, WAVES16, LP, 4>(__VA_ARGS__);      \
    }
#else
#define NRF_TU_NAME(f) f##_32
#define NRF_DISPATCH_MODE(FN, NET, LP, ...)                                                                 \
    switch (mode) {                                                                                         \
        case NRF_MMA_F16X3: return FN<NET<ModeF16X3, 1, LP>, ModeF16X3, 1, 4, LP, 4>(__VA_ARGS__);         \
        default:            return FN<NET<ModeF32, 1, LP>, ModeF32, 1, 4, LP, 4>(__VA_ARGS__);              \
    }
#define NRF_DISPATCH_MODE1(FN, NETT, LP, NT16, WAVES16, ...)                                                \
    switch (mode) {                                                                                         \
        case NRF_MMA_F16X3: return FN<NETT(ModeF16X3, 1), ModeF16X3, 1, 4, LP, 4>(__VA_ARGS__);           \
        default:            return FN<NETT(ModeF32, 1), ModeF32, 1, 4, LP, 4>(__VA_ARGS__);                 \
    }
#endif
#define NRF_NET_V2_10(M, NT) NetV2<M, NT, 10>
#define NRF_NET_V3_12_64(M, NT) NetV3<M, NT, 12, 2>
#define NRF_NET_V3_12_128(M, NT) NetV3<M, NT, 12, 4>

// per-family entry points, two translation units each (fused_v1.hip ... fused_v3w.hip x the two halves) so that hipcc builds them in parallel
#define NRF_DECLARE_FAMILY(fam)                                                                                            \
    int render_##fam##_16(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err);           \
    int render_##fam##_32(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err);           \
    int forward_##fam##_16(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err);               \
    int forward_##fam##_32(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err);               \
    inline int render_##fam(const DeviceNet& net, int mode, const RenderArgs& a, hipStream_t s, std::string& err) {        \
        return (mode == NRF_MMA_BF16 || mode == NRF_MMA_F16) ? render_##fam##_16(net, mode, a, s, err) : render_##fam##_32(net, mode, a, s, err); \
    }                                                                                                                      \
    inline int forward_##fam(const DeviceNet& net, int mode, ForwardKArgs k, hipStream_t s, std::string& err) {            \
        return (mode == NRF_MMA_BF16 || mode == NRF_MMA_F16) ? forward_##fam##_16(net, mode, k, s, err) : forward_##fam##_32(net, mode, k, s, err); \
    }
NRF_DECLARE_FAMILY(v1)
NRF_DECLARE_FAMILY(v2)
NRF_DECLARE_FAMILY(v3)
NRF_DECLARE_FAMILY(v3w)

}  // namespace nrf

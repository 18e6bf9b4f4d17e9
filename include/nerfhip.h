/* nerfhip.h -- C ABI of libnerfhip.so, the MI355X (gfx950) renderer for the
 * ray-marching hot path of ANKITSANJYAL/nerf-few-shot-limitations.
 *
 * The reference has no FFI of its own (it is 100 % Python, SURVEY.md section
 * 8b): the "interface each entry point replaces" is therefore the Python
 * callable named next to it (paths relative to the reference root).  The
 * Python package nerf_few_shot_limitations_amd binds this header with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative NRF_E* code; the
 *     message is kept per thread and read with nrf_last_error();
 *   - no C++ exception crosses the boundary;
 *   - the CALLER owns every buffer (device pointers unless stated "host");
 *     the library owns only the packed weight copies inside a nrf_model;
 *   - all work is enqueued on the caller's stream (void* = hipStream_t,
 *     NULL = the default stream); nothing synchronises, nothing allocates on
 *     a render / staged call;
 *   - all tensors are contiguous fp32, row-major, in the reference's layouts;
 *     ray id r = y*W + x (row-major pixel order) is the integer contract.
 */
#ifndef NERFHIP_H
#define NERFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRF_ABI_VERSION 5

/* error codes */
#define NRF_OK            0
#define NRF_EINVAL       -1   /* bad argument (shape, null pointer, unsupported size) */
#define NRF_EUNSUPPORTED -2   /* architecture / mode not built into this library */
#define NRF_EHIP         -3   /* a HIP runtime call failed */
#define NRF_ENOMEM       -4

/* network families on the path */
#define NRF_NET_V1 1  /* src/models/nerf_model.py:5-24  NeRFMLP(pos_dim=63,hidden,n_layers): PE(10) -> 8x(Linear+ReLU) -> sigma | sigmoid rgb */
#define NRF_NET_V2 2  /* src/models/nerf_mlp.py:41-84   PE(pos) -> DensityMLP -> ColorMLP(feature, PE(dir)): train.py:82-89 with use_dino=False */
#define NRF_NET_V3 3  /* src/models/nerf_mlp.py:86-158  NeRFWithDINO: + lora_dino.py:146-193 NeRFDINOFusion */

/* arithmetic of the MLP contraction (accumulation is always fp32) */
#define NRF_MMA_BF16 0  /* v_mfma_f32_32x32x16_bf16: weights + activations rounded to bf16 (8 mantissa bits: misses the 0.01 dB PSNR bar) */
#define NRF_MMA_F16  1  /* v_mfma_f32_32x32x16_f16 : same rate, 3 more mantissa bits: the throughput mode whose PSNR stays within 0.01 dB of
                           the fp32 render (bench.py's default).  f16-typed weight streams (this mode and NRF_MMA_F16X3) saturate at
                           +-65504 instead of overflowing to inf                                                                     */
#define NRF_MMA_F32  2  /* v_mfma_f32_32x32x2_f32  : exact fp32 fma chain (parity mode, 1/16 rate)           */
#define NRF_MMA_F16X3 3 /* split f16: x = hi + lo (22 bits), W_hi X_hi + W_hi X_lo + W_lo X_hi as three 32x32x16 MFMAs: fp32-class
                           results (meets the 1e-4 bar of the fp32 mode) at up to 1/3 of the 16-bit rate.  Inference entry
                           points only (render / mlp_forward); operands must lie within the f16 range (|x| <= 65504)       */

typedef struct nrf_model nrf_model;

typedef struct nrf_arch {
    int32_t net;        /* NRF_NET_* */
    int32_t pos_freq;   /* L of the position encoding (10 -> 63 features, 12 -> 75) */
    int32_t dir_freq;   /* L of the direction encoding (V2/V3; 4 -> 27 features)   */
    int32_t hidden;     /* hidden width (256)                                       */
    int32_t n_layers;   /* number of Linear+ReLU layers of the trunk (8)            */
    int32_t dino_dim;   /* feature-map channels (V3; 64)                            */
} nrf_arch;

/* One nn.Linear, HOST pointers, weight (out_f,in_f) row-major and bias (out_f),
 * in the order of the module's state_dict:
 *   V1: layers.0 .. layers.{n-1}, sigma_out, rgb_out                        (nerf_model.py:8-14)
 *   V2: density_mlp.density_layers.{0,2,..}, density_head, feature_head,
 *       color_mlp.color_layers.{0,2,4}                                      (nerf_mlp.py:46-57,72-79)
 *   V3: dino_fusion.fusion.{0,2}, dino_fusion.attention.{0,2},
 *       dino_fusion.output_proj, then the V2 list                           (lora_dino.py:153-169) */
typedef struct nrf_linear {
    const float* weight;
    const float* bias;
    int32_t out_f;
    int32_t in_f;
} nrf_linear;

/* DINO side channel of V3 (train.py:203-214): the feature map of ONE source
 * view, its camera and intrinsics. */
typedef struct nrf_dino {
    const float* features;   /* device, (1,Hp,Wp,C) fp32, channel-last (dino_feature_model.py:114-148) */
    int32_t Hp, Wp, C;
    float   inv_pose[16];    /* inverse of the source view's 4x4 camera-to-world, row-major (ray_utils.py:191) */
    float   focal;
    int32_t H, W;            /* image size the projection normalises by (ray_utils.py:199-204) */
} nrf_dino;

typedef struct nrf_render_opts {
    float    near, far;      /* train.py:192-193                                                      */
    int32_t  n_samples;      /* S                                                                     */
    int32_t  lindisp;        /* ray_utils.py:59-62                                                    */
    int32_t  perturb;        /* 1: stratified jitter (ray_utils.py:71-79)                             */
    const float* t_rand;     /* device (R,S) U[0,1) jitter, or NULL -> in-kernel counter RNG(rng_seed) */
    const float* z_ladder;   /* device (S) un-jittered depths z_0..z_{S-1}, or NULL -> computed in-kernel from near/far
                                (scalar torch.linspace formula).  torch's CPU linspace is vectorised and its last ulp
                                depends on the host: a caller that must match the reference bit for bit passes the
                                ladder the reference computes (ray_utils.py:58-66) */
    const float* z_in;       /* device (R,S) explicit, ascending per-ray depths (e.g. the sorted union of the coarse and the
                                importance samples, ray_utils.py:136-139): overrides near/far/perturb/lindisp sampling */
    uint64_t rng_seed;
    float    ert_eps;        /* early ray termination: stop a wave once every live ray has T < eps; 0 = off (reference behaviour) */
    int32_t  white_bkgd;     /* nerf_mlp.py:209-212                                                   */
    int32_t  mma_mode;       /* NRF_MMA_*                                                             */
    const nrf_dino* dino;    /* V3 only (host struct, copied at launch)                               */
    int32_t  out_rgbd;       /* 1: `rgb` points at (R,4) rows [r,g,b,depth] written with one 16-byte store per ray and `depth` is ignored
                                (may be NULL) -- the layout of the multi-GPU gather buffer; `rgb` must be 16-byte aligned, a
                                misaligned pointer is refused with NRF_EINVAL; 0: (R,3) + (R) */
} nrf_render_opts;

/* ---- model handle -------------------------------------------------------- */

/* Build the device-side packed weight streams (all NRF_MMA_* modes) for
 * `arch` from `n_linear` host Linear layers.  Replaces: module construction +
 * .to(device), train.py:82-89 / nerf_model.py:6-14. */
int nrf_model_create(nrf_model** out, int device, const nrf_arch* arch,
                     const nrf_linear* linears, int n_linear);
/* Re-pack after the weights changed (optimizer.step / load_state_dict). */
int nrf_model_update(nrf_model* m, const nrf_linear* linears, int n_linear, void* stream);
void nrf_model_destroy(nrf_model* m);
/* 2*MAC of the Linear layers per ray-sample. */
int64_t nrf_model_flops_per_sample(const nrf_model* m);

/* ---- the fused path -------------------------------------------------------- */

/* sample -> encode -> MLP -> composite for explicit rays.
 * Replaces NeRFDINOTrainer.render_rays, train.py:188-242.
 * rays_o, rays_d: (R,3).  Outputs: rgb (R,3), depth (R); weights (R,S) and
 * z_vals (R,S) may be NULL. */
int nrf_render_rays(const nrf_model* m, const float* rays_o, const float* rays_d, int64_t n_rays,
                    const nrf_render_opts* opts,
                    float* rgb, float* depth, float* weights, float* z_vals, void* stream);

/* Same, with the rays generated in-kernel from a pinhole camera
 * (ray_sampler.py:4-30) for the ray-id range [ray_begin, ray_end) of an HxW
 * image; output row i holds ray ray_begin+i.  c2w = first 3 rows of the pose,
 * row-major, host.  Replaces get_rays + the eval chunk loop, train.py:305-319 /
 * evaluate.py:65-81. */
int nrf_render_camera(const nrf_model* m, int H, int W, float focal, const float c2w[12],
                      int64_t ray_begin, int64_t ray_end, const nrf_render_opts* opts,
                      float* rgb, float* depth, float* weights, float* z_vals, void* stream);

/* Pixel-tile sharded, multi-view form (no reference counterpart; the batch
 * of views is the reference's loop over test views, train.py:304).  The image's rays are cut into
 * tiles of tile_rays consecutive ray ids; ONE launch renders, for each of the n_cams (<= 8) poses
 * c2w[c*12 .. c*12+11] of the same HxW/focal sensor, tiles first_tile, first_tile+tile_step, ...
 * (n_tiles of them, e.g. first_tile=rank, tile_step=world).  Output row
 * (c*n_tiles + k)*tile_rays + j holds ray (first_tile + k*tile_step)*tile_rays + j of view c; rows
 * whose ray id falls beyond H*W repeat the last ray (padding, so every rank's buffer has the same
 * shape for the gather).  Per-ray arithmetic is identical to nrf_render_camera: shards reassemble
 * bit-exactly. */
int nrf_render_cameras_tiles(const nrf_model* m, int H, int W, float focal, const float* c2w, int n_cams,
                             int64_t tile_rays, int64_t first_tile, int64_t tile_step, int64_t n_tiles,
                             const nrf_render_opts* opts,
                             float* rgb, float* depth, float* weights, float* z_vals, void* stream);

/* ---- staged entry points (one reference leaf each; used by the drop-in
 *      Python surface and by the stage-wise parity tests) ------------------- */

/* ray_sampler.py:4-30 == ray_utils.py:4-37.  Rays [ray_begin,ray_end) -> (n,3),(n,3). */
int nrf_get_rays(int H, int W, float focal, const float c2w[12], int64_t ray_begin, int64_t ray_end,
                 float* rays_o, float* rays_d, void* stream);
/* ray_utils.py:39-84 == ray_sampler.py:32-61.  pts (R,S,3) and/or z_vals (R,S) (either may be NULL). */
int nrf_sample_along_rays(const float* rays_o, const float* rays_d, int64_t n_rays,
                          float near, float far, int n_samples, int lindisp, int perturb,
                          const float* t_rand, const float* z_ladder, uint64_t rng_seed,
                          float* pts, float* z_vals, void* stream);
/* positional_encoding.py:20-33 == nerf_mlp.py:17-33.  x (n,dim) -> out (n, dim*(2L+include_input)).
 * freq_bands: NULL = 2^k, k = 0..L-1 (log_sampling=True, every caller of the reference), or a device table of L
 * frequencies (log_sampling=False, positional_encoding.py:18: torch.linspace(1, 2^(L-1), L) as the caller computed it). */
int nrf_encode(const float* x, int64_t n, int dim, int num_freqs, int include_input, const float* freq_bands, float* out, void* stream);
/* V1: nerf_model.py:16-24, x_enc (P, 3*(2*pos_freq+1)) -> out4 (P,4) = [rgb, sigma]. */
int nrf_mlp_forward_v1(const nrf_model* m, int mma_mode, const float* x_enc, int64_t n, float* out4, void* stream);
/* V2/V3: NeRFMLP.forward(positions, directions, dino_features), train.py:229 / nerf_mlp.py:134-158:
 * positions (P,3), directions (P,3), dino (P,dino_dim) or NULL (V2) -> rgb (P,3), density (P,1). */
int nrf_mlp_forward(const nrf_model* m, int mma_mode, const float* positions, const float* directions,
                    const float* dino, int64_t n, float* rgb, float* density, void* stream);
/* nerf_mlp.py:165-215 (VolumeRenderer.forward, eval path) and volume_renderer.py:4-43:
 * rgb (R,S,*) with element stride rgb_stride (3, or 4 for the [r,g,b,sigma] layout),
 * sigma (R,S,*) with stride sigma_stride (1 or 4).  out_depth / out_weights may be NULL. */
int nrf_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride,
                  const float* z_vals, const float* rays_d, int64_t n_rays, int n_samples, int white_bkgd,
                  float* out_rgb, float* out_depth, float* out_weights, void* stream);
/* ray_utils.py:86-143 (intent; the reference function raises):
 * z_vals, weights (R,S) -> samples (R,Ni) and sorted union (R,S+Ni); u: ray r reads u[r * u_ray_stride + j] -- (R,Ni) rows with
 * u_ray_stride = Ni, or ONE row shared by all rays with u_ray_stride = 0 (the caller's torch.linspace(0,1,Ni): its last ulp
 * is host dependent and a nearly empty bin amplifies one ulp of u to 2e-4 of depth) -- or NULL -> linspace(0,1,Ni) in-kernel.
 * The two reductions (:107-109) follow PyTorch's CPU kernels -- weights.sum in ATen's cascade order (8-float vectors),
 * torch.cumsum accumulated in double and rounded per knot -- so the `denom < 1e-5` guard (:131) takes the same branch as the
 * reference's fp32 CPU run would. */
int nrf_sample_pdf(const float* z_vals, const float* weights, int64_t n_rays, int n_samples, int n_importance,
                   const float* u, int64_t u_ray_stride, float* samples, float* z_union, void* stream);
/* ray_utils.py:176-210 + dino_feature_model.py:114-148: points (N,3) -> features (N,C); xy (N,2) may be NULL. */
int nrf_project_fetch(const nrf_dino* dino, const float* points, int64_t n, float* feats, float* xy, void* stream);

/* dino_feature_model.py:114-148 on its own (== lora_dino.py:110-144, multi_scale_dino.py:156-183): bilinear grid_sample
 * (zeros padding, align_corners=False) of a (1,Hp,Wp,C) channel-last map at n normalised image points (n,2) -> (n,C). */
int nrf_sample_features(const float* features, int Hp, int Wp, int C, const float* points_2d, int64_t n, float* feats, void* stream);

/* ---- host-only introspection (no GPU needed; used by the CPU test-suite to replay the
 *      kernel's MFMA walk over the packed stream) -------------------------------- */
/* Packs `linears` exactly as nrf_model_create would for `mma_mode`.  stream_out / bias_out may be
 * NULL to query the sizes (bytes of the fragment stream, floats of the bias table). */
int nrf_debug_pack(const nrf_arch* arch, const nrf_linear* linears, int n_linear, int mma_mode,
                   uint8_t* stream_out, int64_t stream_cap, int64_t* stream_bytes,
                   float* bias_out, int64_t bias_cap, int64_t* n_bias);

/* Host-only, like nrf_debug_pack, for the training path: the transposed (backward-chain) fragment stream, and the
 * saved-tensor / weight-gradient plan serialised as int32:
 *   n_slots, slot_tiles[n_slots], n_jobs, then per job: x_slot, dz_slot, KT, MT, x_first,
 *   row_w[32*MT], row_b[32*MT] (flat-parameter offsets of the weight row / of the bias, -1 = none), col[32*KT];
 *   finally n_mask_planes (ReLU-mask bit planes of 1 KiB per 32 samples appended to the context).
 * Used by the CPU tests to replay the backward pass through a numpy model of the MFMA lane maps. */
int nrf_debug_pack_backward(const nrf_arch* arch, const nrf_linear* linears, int n_linear, int mma_mode,
                            uint8_t* stream_out, int64_t stream_cap, int64_t* stream_bytes);
int nrf_debug_train_plan(const nrf_arch* arch, const nrf_linear* linears, int n_linear,
                         int32_t* out, int64_t cap, int64_t* n_ints);

/* ---- misc ------------------------------------------------------------------ */
const char* nrf_last_error(void);
int nrf_abi_version(void);
/* sizeof() of the ABI structs as this library was compiled (0: nrf_arch, 1: nrf_linear, 2: nrf_dino,
 * 3: nrf_render_opts; -1 otherwise): lets a binding check its struct declarations before the first call. */
int nrf_abi_sizeof(int which);
/* name / average duration bookkeeping is the caller's business: the library never times anything. */

/* ------------------------------------------------------------------------
 * Training path: what loss.backward() and
 * optimizer.step() of src/training/train.py:282-288 run through.  Built for
 * NRF_NET_V1 (nerf_model.NeRFMLP, the network of train_minimal.py:28),
 * NRF_NET_V2 (train.py's model with use_dino=False) and NRF_NET_V3 (use_dino=True).
 *
 * Parameters and gradients travel as ONE flat fp32 device vector: the Linears
 * in state_dict order, each weight (out_f*in_f, row-major) followed by its
 * bias (out_f) -- nrf_param_count() floats.  The Python module keeps its
 * nn.Parameters as views into such a vector, so torch.optim.Adam (or
 * nrf_adam_step) updates it in place and nrf_model_update_device re-packs
 * the MFMA operand streams from it without leaving the device.
 * ------------------------------------------------------------------------ */
int64_t nrf_param_count(const nrf_model* m);

/* Replaces nrf_model_update for device-resident parameters: re-packs the
 * forward (and, once training has been used, backward) streams of the modes in
 * mode_mask (bit NRF_MMA_*) from flat_params, plus the bias table.  Enqueued on
 * `stream`; a single mode is one kernel launch.  The streams of the modes NOT in the
 * mask keep the previous parameters (a later render / backward in such a mode needs
 * its own update; nrf_mlp_backward* refuses stale backward weights). */
int nrf_model_update_device(nrf_model* m, const float* flat_params, int mode_mask, void* stream);

/* Bytes of saved tensors ("context") a forward_train/backward pair needs for n
 * samples; the caller allocates it (device) and keeps it until backward ran.  It holds
 * the saved operand tiles of every layer, the ReLU bit planes, (V3) the softmax gate and
 * the weight-gradient partial sums: about 9 KiB per sample for the 8x256 network in the
 * 16-bit modes plus ~64 MiB of partial sums. */
int64_t nrf_train_context_bytes(nrf_model* m, int mma_mode, int64_t n);

/* nerf_model.py:16-24 forward with grad enabled: out4 as nrf_mlp_forward_v1,
 * and every layer's activations saved into ctx. */
int nrf_mlp_forward_train_v1(nrf_model* m, int mma_mode, const float* x_enc, int64_t n, float* out4,
                             void* ctx, int64_t ctx_bytes, void* stream);

/* Its backward: g_out4 = dL/d out4 (n,4); flat_grad (nrf_param_count floats)
 * += dL/d parameters.  out4 is the tensor forward_train wrote.  No gradient
 * with respect to x_enc is produced (the reference never needs one). */
int nrf_mlp_backward_v1(nrf_model* m, int mma_mode, const float* out4, const float* g_out4, int64_t n,
                        void* ctx, int64_t ctx_bytes, float* flat_grad, void* stream);

/* The same pair for NRF_NET_V2 (train.py:229 `rgb, density = self.nerf_model(positions, directions, None)` with grad
 * enabled; nerf_mlp.py:134-158 without the DINO branch).  rgb (n,3) / density (n,1) are written by the forward and
 * read again by the backward (sigmoid', relu'); g_rgb / g_density are dL/d of them.  No gradient with respect to
 * positions, directions or the DINO features.  NRF_NET_V3 (nerf_mlp.py:134-158 with lora_dino.py:171-193: the fusion
 * block runs twice on the same weights, gated by a 2-way softmax) takes the per-sample features as `dino`; up to 8 trunk
 * layers. */
int nrf_mlp_forward_train(nrf_model* m, int mma_mode, const float* positions, const float* directions,
                          const float* dino /* NRF_NET_V3: (n, dino_dim) per-sample features, else NULL */, int64_t n,
                          float* rgb, float* density, void* ctx, int64_t ctx_bytes, void* stream);
int nrf_mlp_backward(nrf_model* m, int mma_mode, const float* rgb, const float* density,
                     const float* g_rgb, const float* g_density, int64_t n,
                     void* ctx, int64_t ctx_bytes, float* flat_grad, void* stream);

/* Backward of nrf_composite (autograd through nerf_mlp.py:181-212): given
 * dL/d rgb_map (n_rays,3), optionally dL/d depth (n_rays) and dL/d weights
 * (n_rays,S), writes dL/d rgb (strided like the inputs) and dL/d sigma. */
int nrf_composite_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride,
                           const float* z_vals, const float* rays_d, int64_t n_rays, int n_samples, int white_bkgd,
                           const float* g_rgb, const float* g_depth, const float* g_weights,
                           float* d_rgb, int d_rgb_stride, float* d_sigma, int d_sigma_stride, void* stream);

/* `rgb_weight * nn.MSELoss()(pred, target)` (train.py:36-44) and its gradient in one launch: loss[0] = weight * mean((pred -
 * target)^2) over n values, g_pred = d loss / d pred.  n <= 2^22 (ray batches). */
int nrf_mse_grad(const float* pred, const float* target, int64_t n, float weight, float* g_pred, float* loss, void* stream);

/* nrf_composite + nrf_mse_grad + nrf_composite_backward in ONE launch (the three steps between the network's forward and its
 * backward in the reference's train_step, train.py:236,36-44,285): a ray's loss gradient needs only its own prediction and
 * target -- d loss / d pred = 2 weight (pred - target) / (3 n_rays).  d_rgb / d_sigma are bit-equal to the three-call sequence;
 * pred (n_rays,3) may be NULL.
 *   ray_loss : n_rays floats, receives each ray's squared error; the loss is weight * sum(ray_loss) / (3 n_rays)
 *              (nrf_adam_step_loss adds them up in a fixed order as a side job of its launch);
 *   zero_buf : optional, zero_n floats cleared by the same launch (the flat gradient vector nrf_mlp_backward* adds into). */
int nrf_composite_mse_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride,
                               const float* z_vals, const float* rays_d, int64_t n_rays, int n_samples, int white_bkgd,
                               const float* target, float weight, float* pred,
                               float* d_rgb, int d_rgb_stride, float* d_sigma, int d_sigma_stride,
                               float* ray_loss, float* zero_buf, int64_t zero_n, void* stream);

/* torch.optim.Adam's update (train.py:113-118; no amsgrad) on flat vectors;
 * step counts from 1. */
int nrf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
/* The same update, and as a side job of the launch loss[0] = loss_weight * sum(ray_loss[0..n_rays)) / (3 n_rays), summed in a
 * fixed order (ray_loss: nrf_composite_mse_backward's per-ray squared errors). */
int nrf_adam_step_loss(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       const float* ray_loss, int64_t n_rays, float loss_weight, float* loss, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFHIP_H */

// kernels.hpp -- host-visible launch interface between api.cpp and the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/nerfhip.h"
#include "device_math.hpp"

namespace nrf {

constexpr int kMaxCams = 8;
constexpr int kModes = 4;           // NRF_MMA_BF16, _F16, _F32, _F16X3 (the training path is built for the first three)

// device-side image of one nrf_model
struct DeviceNet {
    nrf_arch arch;
    const void* stream[kModes];   // packed fragment streams, indexed by NRF_MMA_*
    uint32_t n_chunks[kModes];
    const float* bias;          // bias table (fp32, shared by all modes)
    int n_bias;
    int64_t flops_per_sample;
    int device;
    int cu_count;
    unsigned long long* queues;   // kQueueSlots counters; launches cycle through them
};

constexpr int kQueueSlots = 64;

struct DinoDev {                // V3 side channel, by value in the kernel arguments
    const float* features;
    int Hp, Wp, C;
    float inv_pose[12];         // first 3 rows of inverse(pose)
    float focal;
    int H, W;
};

struct RenderArgs {
    // rays: explicit (rays_o/rays_d) or camera (cam + ray_begin)
    const float* rays_o;
    const float* rays_d;
    int camera_mode;
    int n_cams;             // camera mode: a batch of up to kMaxCams views of one HxW sensor rendered by ONE launch
    int64_t rays_per_cam;   // local rays [c*rays_per_cam, (c+1)*rays_per_cam) belong to cams[c]
    Camera cams[8];
    int64_t ray_begin;
    int64_t n_rays;
    int64_t tile_rays;      // camera mode: local ray i is global ray ray_begin + (i / tile_rays) * tile_stride + i % tile_rays,
    int64_t tile_stride;    // clamped to the image (tile_rays >= n_rays: one contiguous range)
    // sampling
    float near, far;
    int n_samples, lindisp, perturb;
    const float* t_rand;
    const float* z_ladder;
    const float* z_in;
    uint64_t seed;
    // compositing
    float ert_eps;
    int white_bkgd;
    // outputs
    int interleaved;        // 1: rgb points at (R,4) rows [r,g,b,depth] (one 16-B store per ray), depth is unused
    float* rgb;
    float* depth;
    float* weights;
    float* z_vals;
    DinoDev dino;
    unsigned long long* queue;   // ERT (ray-queue) kernel: device counter of rays handed out, zeroed before the launch
    int spw_log2;                // render_kernel: log2 of the samples per ray and MLP pass (set by the launcher)
};

int launch_render(const DeviceNet& net, int mma_mode, const RenderArgs& a, hipStream_t s, std::string& err);
int launch_forward_v1(const DeviceNet& net, int mma_mode, const float* x_enc, int64_t n, float* out4, hipStream_t s, std::string& err);
int launch_forward(const DeviceNet& net, int mma_mode, const float* pos, const float* dir, const float* dino, int64_t n,
                   float* rgb, float* density, hipStream_t s, std::string& err);

// ---- training path (train_v1.hip) -------------------------------------------------
constexpr int kMaxSlots = 40;
constexpr int kMaxJobs = 24;
constexpr int kMaxMaskSlots = 20;
constexpr int kMapStride = 3 * 320;      // per weight-gradient job: row_w[320] | row_b[320] | col[320]

struct TrainDev {
    const void* bstream[3];     // backward-chain (transposed) streams by NRF_MMA_*
    uint32_t n_bchunks[3];
    const int32_t* maps;        // device, n_jobs * kMapStride
    int n_slots;
    int slot_tiles[kMaxSlots];  // feature tiles per saved-tensor slot
    int n_mask_slots;           // ReLU-mask bit planes (one per masked layer)
    int aux_floats;             // extra fp32 values saved per sample (V3: the softmax gate)
    int cu_count;               // sizes the weight-gradient partial sums (one block per workgroup)
    int n_jobs;
    int job_x_slot[kMaxJobs], job_dz_slot[kMaxJobs], job_KT[kMaxJobs], job_MT[kMaxJobs], job_x_first[kMaxJobs];
    int64_t n_params;
};

int64_t train_ctx_bytes(const TrainDev& t, int mma_mode, int64_t n);
int launch_train_forward(const DeviceNet& net, const TrainDev& t, int mma_mode, const float* x_enc, int64_t n, float* out4, void* ctx,
                         hipStream_t s, std::string& err);
// dZ chain + weight gradients: grad (flat, n_params floats) += dL/dparams
int launch_train_backward(const DeviceNet& net, const TrainDev& t, int mma_mode, const float* out4, const float* g_out4, int64_t n,
                          void* ctx, float* grad, hipStream_t s, std::string& err);
int launch_train_forward_v2(const DeviceNet& net, const TrainDev& t, int mma_mode, const float* pos, const float* dir, int64_t n, float* rgb,
                            float* density, void* ctx, hipStream_t s, std::string& err);
int launch_train_backward_v2(const DeviceNet& net, const TrainDev& t, int mma_mode, const float* rgb, const float* density,
                             const float* g_rgb, const float* g_density, int64_t n, void* ctx, float* grad, hipStream_t s, std::string& err);
int launch_train_forward_v3(const DeviceNet& net, const TrainDev& t, int mma_mode, const float* pos, const float* dir, const float* dino, int64_t n,
                            float* rgb, float* density, void* ctx, hipStream_t s, std::string& err);
int launch_train_backward_v3(const DeviceNet& net, const TrainDev& t, int mma_mode, const float* rgb, const float* density,
                             const float* g_rgb, const float* g_density, int64_t n, void* ctx, float* grad, hipStream_t s, std::string& err);
int launch_repack(const float* flat, const int32_t* src, int64_t n_elems, int mma_mode, void* out, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step,
                const float* ray_loss, int64_t n_rays, float loss_weight, float* loss, hipStream_t s);
int launch_mse_grad(const float* pred, const float* target, int64_t n, float weight, float* g_pred, float* loss, hipStream_t s);
int launch_composite_mse_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z, const float* rays_d,
                                  int64_t n_rays, int S, int white_bkgd, const float* target, float weight, float* pred, float* d_rgb,
                                  int d_rgb_stride, float* d_sigma, int d_sigma_stride, float* ray_loss, float* zero_buf,
                                  int64_t zero_n, hipStream_t s);
int launch_repack3(const float* flat, const int32_t* const src[3], const int64_t n_elems[3], const int modes[3], void* const out[3], hipStream_t s);
int launch_composite_backward(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z, const float* rays_d,
                              int64_t n_rays, int S, int white_bkgd, const float* g_rgb, const float* g_depth, const float* g_w,
                              float* d_rgb, int d_rgb_stride, float* d_sigma, int d_sigma_stride, hipStream_t s);

// staged kernels (staged_kernels.hip)
int launch_get_rays(const Camera& cam, int64_t ray_begin, int64_t n, float* rays_o, float* rays_d, hipStream_t s);
int launch_sample(const float* rays_o, const float* rays_d, int64_t n_rays, float near, float far, int S, int lindisp,
                  int perturb, const float* t_rand, const float* z_ladder, uint64_t seed, float* pts, float* z_vals, hipStream_t s);
int launch_encode(const float* x, int64_t n, int dim, int L, int include_input, const float* freq_bands, float* out, hipStream_t s);
int launch_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z, const float* rays_d,
                     int64_t n_rays, int S, int white_bkgd, float* out_rgb, float* out_depth, float* out_w, hipStream_t s);
int launch_sample_pdf(const float* z, const float* w, int64_t n_rays, int S, int Ni, const float* u, int64_t u_ray_stride, float* samples,
                      float* z_union, hipStream_t s);
int launch_project_fetch(const DinoDev& d, const float* points, int64_t n, float* feats, float* xy, hipStream_t s);
int launch_sample_features(const float* features, int Hp, int Wp, int C, const float* points_2d, int64_t n, float* feats, hipStream_t s);

}  // namespace nrf

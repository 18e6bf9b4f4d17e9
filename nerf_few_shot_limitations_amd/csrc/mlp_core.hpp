// mlp_core.hpp -- the MFMA machinery of the fused NeRF MLP (gfx950 only).
//
// Geometry (DESIGN.md "MLP kernel"):
//   * one workgroup = 4 waves = one CU; a wave owns 32*NT samples; features sit
//     on the MFMA ROW axis, samples on the LANE (column) axis:
//          H_next[feature][sample] = W[feature][k] * H[k][sample]
//     so a layer's 32x32 fp32 accumulator tile (column on the lane, rows in its
//     16 registers) is, after ReLU + down-conversion, directly the B operand of
//     the next layer -- activations never leave registers;
//   * the weights (A operand) are pre-packed on the host into 1-KiB "fragments"
//     (64 lanes x 16 B, exactly one ds_read_b128 per lane) in the order the
//     kernel consumes them and streamed L2 -> LDS with global_load_lds into a
//     ring of NSLOT 16-KiB chunks, NSLOT-1 chunks ahead of the MFMAs; all four
//     waves share every fragment; one raw s_barrier per chunk (32*NT MFMAs per
//     wave) with a counted vmcnt keeps the prefetch in flight across it.
//
// K order inside a fragment (cdna_hip_programming.md section 3, "An accumulator
// tile as the next MFMA's operand"): for lane (i = lane&31, h = lane>>5)
//   16-bit modes: fragment (m, t, s) element j  = W[32m+i][32t + 16s + 8(j>>2) + 4h + (j&3)]
//   fp32 mode   : fragment (m, t, g) element e  = W[32m+i][32t + 8g + 4h + e]
// which is what nrf::pack_* in packing.cpp writes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace nrf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NRF_LDS __attribute__((address_space(3)))
#define NRF_GLB __attribute__((address_space(1)))

constexpr int kFragBytes = 1024;                     // 64 lanes x 16 B
constexpr int kChunkFrags = 16;
constexpr int kChunkBytes = kFragBytes * kChunkFrags;  // 16 KiB
constexpr int kSlots = 8;                              // ring depth (128 KiB)

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------
// weight streamer
// ---------------------------------------------------------------------------
// WAVES = waves of the workgroup that share the stream (4: one per SIMD, or 8: two per SIMD)
template <int WAVES>
struct Pipe {
    static constexpr int kFragsPerWave = kChunkFrags / WAVES;    // glds instructions per wave per chunk
    const NRF_GLB char* src;   // packed stream + wave*4 KiB + lane*16
    NRF_LDS char* ring;        // ring base (LDS)
    NRF_LDS char* cur;         // current chunk + lane*16
    uint32_t n_chunks;         // chunks per MLP pass (the stream wraps)
    uint32_t issue_chunk;
    uint32_t issue_slot;
    uint32_t read_slot;
    uint32_t wave_off;         // wave * 4 KiB (wave-uniform)
    uint32_t lane_off;         // lane * 16

    __device__ __forceinline__ void init(const void* stream, uint32_t chunks, NRF_LDS char* ring_base) {
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane_off = (threadIdx.x & 63) * 16;
        wave_off = wave * (kFragsPerWave * kFragBytes);
        src = (const NRF_GLB char*)stream + wave_off + lane_off;
        ring = ring_base;
        n_chunks = chunks;
        issue_chunk = 0; issue_slot = 0; read_slot = 0;
        cur = ring_base + lane_off;
    }
    __device__ __forceinline__ void issue_one() {
        const NRF_GLB char* g = src + (size_t)issue_chunk * kChunkBytes;
        NRF_LDS char* l = ring + issue_slot * kChunkBytes + wave_off;
#pragma unroll
        for (int i = 0; i < kFragsPerWave; ++i)
            __builtin_amdgcn_global_load_lds((const NRF_GLB void*)(g + i * kFragBytes), (NRF_LDS void*)(l + i * kFragBytes), 16, 0, 0);
        issue_chunk = (issue_chunk + 1 == n_chunks) ? 0u : issue_chunk + 1;
        issue_slot = (issue_slot + 1 == (uint32_t)kSlots) ? 0u : issue_slot + 1;
    }
    // fill the ring: kSlots-1 chunks in flight
    __device__ __forceinline__ void start() {
        for (int k = 0; k < kSlots - 1; ++k) issue_one();
    }
    // make the next chunk readable, free the previous one, keep the ring full
    __device__ __forceinline__ void acquire() {
        // own part of the chunk has landed once at most (kSlots-2) younger chunks' loads are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFragsPerWave * (kSlots - 2)) : "memory");
        __builtin_amdgcn_s_barrier();     // every wave's part landed; every wave is done with the previous chunk
        __builtin_amdgcn_sched_barrier(0);
        issue_one();                      // refill the slot that was just released
        cur = ring + read_slot * kChunkBytes + lane_off;
        read_slot = (read_slot + 1 == (uint32_t)kSlots) ? 0u : read_slot + 1;
    }
    // all LDS-DMA must have landed before the workgroup gives its LDS back
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ---------------------------------------------------------------------------
// arithmetic modes
// ---------------------------------------------------------------------------
struct ModeBF16 {
    static constexpr int SUB = 2;            // fragments per (m-tile, k-tile)
    static constexpr bool FAST_TRIG = true;  // v_sin on exactly reduced turns: error far below bf16 resolution
    static constexpr bool FAST_EXP = true;   // v_exp based exp/sigmoid in the compositor
    typedef bf16x8 frag_t;
    struct Act { bf16x8 f[2]; };
    __device__ static __forceinline__ void mma(f32x16& acc, const frag_t& a, const Act& b, int s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b.f[s], acc, 0, 0, 0);
    }
    template <bool RELU>
    __device__ static __forceinline__ Act to_act(const f32x16& v) {
        Act o;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = v[8 * s + j];
                o.f[s][j] = (__bf16)(RELU ? fmaxf(x, 0.0f) : x);
            }
        return o;
    }
};

struct ModeF16 {
    static constexpr int SUB = 2;
    static constexpr bool FAST_TRIG = true;
    static constexpr bool FAST_EXP = false;
    typedef f16x8 frag_t;
    struct Act { f16x8 f[2]; };
    __device__ static __forceinline__ void mma(f32x16& acc, const frag_t& a, const Act& b, int s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b.f[s], acc, 0, 0, 0);
    }
    template <bool RELU>
    __device__ static __forceinline__ Act to_act(const f32x16& v) {
        Act o;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = v[8 * s + j];
                o.f[s][j] = (_Float16)(RELU ? fmaxf(x, 0.0f) : x);
            }
        return o;
    }
};

struct ModeF32 {
    static constexpr int SUB = 4;
    static constexpr bool FAST_TRIG = false;
    static constexpr bool FAST_EXP = false;
    typedef f32x4 frag_t;
    struct Act { float r[16]; };
    __device__ static __forceinline__ void mma(f32x16& acc, const frag_t& a, const Act& b, int g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b.r[4 * g + e], acc, 0, 0, 0);
    }
    template <bool RELU>
    __device__ static __forceinline__ Act to_act(const f32x16& v) {
        Act o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o.r[r] = RELU ? fmaxf(v[r], 0.0f) : v[r];
        return o;
    }
};

// accumulator rows of this lane: reg r <-> row (r&3) + 8*(r>>2) + 4*(lane>>5).  The bias of a
// layer initialises the accumulator: 4 x 16-B LDS reads (broadcast inside each lane half).
__device__ __forceinline__ void load_bias(f32x16& acc, const NRF_LDS float* bias_rows /* + 32*m */, int h) {
    const NRF_LDS f32x4* p = (const NRF_LDS f32x4*)(bias_rows + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = p[2 * g];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = v[e];
    }
}

// One Linear layer: out[m][n] (32 features x 32 samples each) = W * in + bias, for MT output
// tiles, KT input tiles, NT sample tiles.  Consumes MT*KT*SUB fragments from the pipe (the host
// pads every layer to whole chunks, so a layer always starts on a chunk boundary).
// `fin(m, acc)` receives each finished tile's raw accumulators.
template <class Mode, int KT, int MT, int NT, class P, class Fin>
__device__ __forceinline__ void dense(P& pipe, const NRF_LDS float* bias, int h,
                                      const typename Mode::Act (&in)[KT][NT], Fin&& fin) {
    static_for<MT>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        f32x16 acc[NT];
        load_bias(acc[0], bias + 32 * m, h);
#pragma unroll
        for (int n = 1; n < NT; ++n) acc[n] = acc[0];
        static_for<KT>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            static_for<Mode::SUB>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                constexpr int f = (m * KT + t) * Mode::SUB + s;
                if constexpr (f % kChunkFrags == 0) pipe.acquire();
                const typename Mode::frag_t a =
                    *(const NRF_LDS typename Mode::frag_t*)(pipe.cur + (f % kChunkFrags) * kFragBytes);
#pragma unroll
                for (int n = 0; n < NT; ++n) Mode::mma(acc[n], a, in[t][n], s);
            });
        });
        fin(m_, acc);
    });
}

// layer with an activation, producing the next layer's operand tiles
template <class Mode, int KT, int MT, int NT, bool RELU, class P>
__device__ __forceinline__ void dense_act(P& pipe, const NRF_LDS float* bias, int h,
                                          const typename Mode::Act (&in)[KT][NT], typename Mode::Act (&out)[MT][NT]) {
    dense<Mode, KT, MT, NT>(pipe, bias, h, in, [&](auto m_, f32x16(&acc)[NT]) {
        constexpr int m = decltype(m_)::value;
#pragma unroll
        for (int n = 0; n < NT; ++n) out[m][n] = Mode::template to_act<RELU>(acc[n]);
    });
}

// head layer: one output tile, raw accumulators back to the caller
template <class Mode, int KT, int NT, class P>
__device__ __forceinline__ void dense_head(P& pipe, const NRF_LDS float* bias, int h,
                                           const typename Mode::Act (&in)[KT][NT], f32x16 (&out)[NT]) {
    dense<Mode, KT, 1, NT>(pipe, bias, h, in, [&](auto, f32x16(&acc)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n) out[n] = acc[n];
    });
}

constexpr __host__ __device__ int chunks_for(int frags) { return (frags + kChunkFrags - 1) / kChunkFrags; }

}  // namespace nrf

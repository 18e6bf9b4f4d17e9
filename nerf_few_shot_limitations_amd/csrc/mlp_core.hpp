// mlp_core.hpp -- the MFMA machinery of the fused NeRF MLP (gfx950 only).
//
// Geometry (DESIGN.md "MLP kernel"):
//   * one workgroup = one CU = 256 sample columns per weight pass: WAVES waves x NT tiles of 32 columns (8 x 1: two waves per
//     SIMD; 4 x 2 or 4 x 1: one wave per SIMD -- fused_impl.hpp, "Workgroup geometry"); features sit on the MFMA ROW axis,
//     samples on the LANE (column) axis:
//          H_next[feature][sample] = W[feature][k] * H[k][sample]
//     so a layer's 32x32 fp32 accumulator tile (column on the lane, rows in its
//     16 registers) is, after ReLU + down-conversion, directly the B operand of
//     the next layer -- activations never leave registers;
//   * the weights (A operand) are pre-packed on the host into 1-KiB "fragments"
//     (64 lanes x 16 B, exactly one ds_read_b128 per lane) in the order the
//     kernel consumes them and streamed L2 -> LDS with global_load_lds into a
//     ring of NSLOT 16-KiB chunks, NSLOT-2 chunks ahead of the MFMAs; all
//     waves share every fragment (a wave feeds it to NT MFMAs); one raw s_barrier per chunk
//     (16*NT MFMAs per wave) with a counted vmcnt keeps the prefetch in flight across it.
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

#include "stream_config.hpp"

namespace nrf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// two fp32 -> one packed 16-bit pair (v_cvt_pk_*), optional ReLU as a signed-integer max with 0 on the pair
// (v_pk_max_i16: a negative float has its sign bit set, i.e. is a negative int16) -- one VALU op per two
// activations instead of an fp32 max (+ canonicalisation) per element
template <class V2, bool RELU>
__device__ __forceinline__ int pack_pair(float a, float b) {
    const f32x2 ab = {a, b};
    i16x2 q = __builtin_bit_cast(i16x2, __builtin_convertvector(ab, V2));
    if (RELU) {
        const i16x2 zero = {0, 0};
        q = __builtin_elementwise_max(q, zero);
    }
    return __builtin_bit_cast(int, q);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NRF_LDS __attribute__((address_space(3)))
#define NRF_GLB __attribute__((address_space(1)))

constexpr int kFragBytes = 1024;                     // 64 lanes x 16 B
constexpr int kChunkFrags = NRF_CHUNK_FRAGS;
constexpr int kChunkBytes = kFragBytes * kChunkFrags;  // 16 KiB
constexpr int kSlots = NRF_SLOTS;                      // ring depth (128 KiB)

// An epilogue slice of the pinned walk ends in an empty, opaque asm on the words it produced (see ModeF16X3::to_act_pair).
#define NRF_PIN_ACT1(a) asm volatile("" : "+v"(a))
#define NRF_PIN_ACT2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// ... and, when the translation unit is built with VGPR-form MFMAs (-mllvm -amdgpu-mfma-vgpr-form=1, -DNRF_ACT_AGPR), a finished
// 128-bit operand image is moved to the AGPR half of the register file, where the next layer's MFMAs read it directly
#ifdef NRF_ACT_AGPR
#define NRF_PARK_ACT(v) asm volatile("" : "+a"(v))
#else
#define NRF_PARK_ACT(v)
#endif

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
// WAVES = waves of the workgroup that share the stream (4: one per SIMD, or 8: two per SIMD).
//
// Ring protocol.  Chunk X of the (periodic) stream lives in slot X mod kSlots.  acquire(X):
//   1. s_waitcnt vmcnt(kFragsPerWave*(kAhead-1)): this wave's own part of chunk X has landed (the kAhead-1
//      younger chunks X+1..X+kAhead-1 may still be in flight);
//   2. raw s_barrier: every wave's part has landed, and every wave has issued all its reads of chunk X-2;
//   3. issue chunk X+kAhead into slot (X+kAhead) mod kSlots == slot of chunk X-2 (kAhead = kSlots-2).
// acquire(X) is called a few fragments BEFORE the reads of chunk X-1 are finished (dense() below), so that the
// first fragments of chunk X are already in flight while the last MFMAs of chunk X-1 run: that is why the
// slot recycled at the barrier is the one two chunks back, not one.
// ASM_DMA: issue the LDS-DMA as inline asm (the one-wave-per-SIMD modes: see dense_pinned()).
template <int WAVES, bool ASM_DMA = false>
struct Pipe {
    static constexpr int kFragsPerWave = kChunkFrags / WAVES;    // glds instructions per wave per chunk
    static constexpr int kAhead = kSlots - 2;                    // chunks in flight / landed ahead of the reader
    const NRF_GLB char* src;   // packed stream + wave*kFragsPerWave KiB + lane*16
    NRF_LDS char* ring;        // ring base (LDS)
    NRF_LDS char* base[2];     // chunk bases (+ lane*16) by parity of the chunk's index inside the layer
    uint32_t n_chunks;         // chunks per MLP pass (the stream wraps)
    uint32_t issue_chunk;
    uint32_t issue_slot;
    uint32_t read_slot;
    uint32_t wave_off;         // wave * kFragsPerWave KiB (wave-uniform)
    uint32_t lane_off;         // lane * 16
    uint32_t ablate;           // timing experiments only (builds with -DNRF_ABLATE_BUILD, env NRF_ABLATE): 1 = stop streaming after the first fill, 2 = no barriers
    uint32_t skip;             // wave-uniform flag of the ray-queue kernel: this wave has run dry (it keeps computing, stores nothing)

    __device__ __forceinline__ void init(const void* stream, uint32_t chunks, NRF_LDS char* ring_base, uint32_t ablate_flags = 0) {
        ablate = ablate_flags;
        skip = 0;
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane_off = (threadIdx.x & 63) * 16;
        wave_off = wave * (kFragsPerWave * kFragBytes);
        src = (const NRF_GLB char*)stream + wave_off + lane_off;
        ring = ring_base;
        n_chunks = chunks;
        issue_chunk = 0; issue_slot = 0; read_slot = 0;
        base[0] = base[1] = ring_base + lane_off;
    }
    // One chunk: this wave's kFragsPerWave consecutive fragments.  ONE address pair and ONE LDS base (M0) serve all of them: the
    // instruction's immediate offset moves the global and the LDS address alike.  (Per-fragment address arithmetic and M0 traffic
    // made this ~45 scalar / vector instructions per chunk; at one wave per SIMD that was a ~120-cycle hole in the MFMA stream
    // behind every chunk barrier.)
    __device__ __forceinline__ void issue_one() {
        const NRF_GLB char* g = src + (size_t)issue_chunk * kChunkBytes;
        NRF_LDS char* l = ring + issue_slot * kChunkBytes + wave_off;
        if constexpr (ASM_DMA) {
            // hipcc then tracks neither the DMA's vmcnt (waited for by hand in acquire()) nor -- the point -- an LDS access
            // through a FLAT-encoded instruction, whose "pending flat" state makes its waitcnt pass answer the fragment reads
            // that follow with lgkmcnt(0) instead of counted waits.
            static_assert(kFragsPerWave == 2 || kFragsPerWave == 4, "LDS-DMA asm written for 2 or 4 fragments per wave and chunk");
            const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)l);
            if constexpr (kFragsPerWave == 4)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                             : : "v"(g), "s"(dst) : "memory", "m0");
            else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                             : : "v"(g), "s"(dst) : "memory", "m0");
        } else {
            static_for<kFragsPerWave>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                __builtin_amdgcn_global_load_lds((const NRF_GLB void*)g, (NRF_LDS void*)l, 16, i * kFragBytes, 0);
            });
        }
        issue_chunk = (issue_chunk + 1 == n_chunks) ? 0u : issue_chunk + 1;
        issue_slot = (issue_slot + 1 == (uint32_t)kSlots) ? 0u : issue_slot + 1;
    }
    // fill the ring: kAhead chunks in flight
    __device__ __forceinline__ void start() {
        for (int k = 0; k < kAhead; ++k) issue_one();
    }
    // make the next chunk of the stream readable through base[parity]
    __device__ __forceinline__ void acquire(int parity) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFragsPerWave * (kAhead - 1)) : "memory");
#ifdef NRF_ABLATE_BUILD
        if (!(ablate & 2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!(ablate & 1)) issue_one();
#else
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_one();
#endif
        base[parity] = ring + read_slot * kChunkBytes + lane_off;
        read_slot = (read_slot + 1 == (uint32_t)kSlots) ? 0u : read_slot + 1;
    }
    // The pinned walk's form of acquire(): the barrier and the first of this wave's kFragsPerWave LDS-DMA instructions now, the
    // others one per fragment step (issue_part<k>() rides with the read of fragment k of the chunk).  Back to back behind the
    // barrier, the four 1-KiB DMA instructions held the wave's issue port for ~100 cycles with an empty MFMA queue (one wave per
    // SIMD); spread out, each sits in the shadow of its step's MFMAs (+0.8 % bf16, +1.7 % split-f16).  What the stream still costs
    // (-6 % by ablation, proportional to the number of DMA instructions) is not the issue port: the four waves, released
    // together by the chunk barrier, reach every DMA instruction at the same moment.  Giving each wave its own steps
    // (a wave-id test per step) costs more than it saves: -9 % (profiles/r02_ab_dma_issue.txt).
    const NRF_GLB char* dma_g;
    uint32_t dma_m0;
    __device__ __forceinline__ void acquire_begin(int parity) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFragsPerWave * (kAhead - 1)) : "memory");
#ifdef NRF_ABLATE_BUILD
        if (!(ablate & 2))
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dma_g = src + (size_t)issue_chunk * kChunkBytes;
        dma_m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ring + issue_slot * kChunkBytes + wave_off));
        issue_chunk = (issue_chunk + 1 == n_chunks) ? 0u : issue_chunk + 1;
        issue_slot = (issue_slot + 1 == (uint32_t)kSlots) ? 0u : issue_slot + 1;
        base[parity] = ring + read_slot * kChunkBytes + lane_off;
        read_slot = (read_slot + 1 == (uint32_t)kSlots) ? 0u : read_slot + 1;
        issue_part<0>();
    }
    template <int K>
    __device__ __forceinline__ void issue_part() {
        static_assert(ASM_DMA && K < kFragsPerWave, "issue_part: the pinned walk's pipe");
#ifdef NRF_ABLATE_BUILD
        if ((ablate & 1) || ((ablate & 4) && K >= 1) || ((ablate & 8) && K >= 2)) return;      // 4 / 8: a quarter / half of the DMA instructions (vmcnt no longer counts right: pair with 2)
#endif
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                     : : "v"(dma_g), "s"(dma_m0), "n"(K * kFragBytes) : "memory", "m0");
    }
    // all LDS-DMA must have landed before the workgroup gives its LDS back
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ---------------------------------------------------------------------------
// arithmetic modes
// ---------------------------------------------------------------------------
struct ModeBF16 {
    static constexpr int SUB = 2;            // fragments per (m-tile, k-tile)
    static constexpr bool kPinned = false;   // pinned by geometry (pinned_walk): the inference kernels run NT = 2; the training kernels (NT = 1, two waves per SIMD) keep dense()
    static constexpr int TRIG = 1;           // v_sin on exactly reduced turns: error far below bf16 resolution (nets.hpp:encode3)
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
        for (int s = 0; s < 2; ++s) {
            i32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack_pair<bf16x2, RELU>(v[8 * s + 2 * j], v[8 * s + 2 * j + 1]);
            o.f[s] = __builtin_bit_cast(bf16x8, w);
        }
        return o;
    }
    template <bool RELU>
    __device__ static __forceinline__ void to_act_pair(const f32x16& v, int j, Act& o) {     // dense_pinned's epilogue slice
        int nw = pack_pair<bf16x2, RELU>(v[2 * j], v[2 * j + 1]);
        NRF_PIN_ACT1(nw);
        i32x4 w = __builtin_bit_cast(i32x4, o.f[j >> 2]);
        w[j & 3] = nw;
        if ((j & 3) == 3) NRF_PARK_ACT(w);
        o.f[j >> 2] = __builtin_bit_cast(bf16x8, w);
    }
};

struct ModeF16 {
    static constexpr int SUB = 2;
    static constexpr bool kPinned = false;
    static constexpr int TRIG = 1;
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
        for (int s = 0; s < 2; ++s) {
            i32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack_pair<f16x2, RELU>(v[8 * s + 2 * j], v[8 * s + 2 * j + 1]);
            o.f[s] = __builtin_bit_cast(f16x8, w);
        }
        return o;
    }
    template <bool RELU>
    __device__ static __forceinline__ void to_act_pair(const f32x16& v, int j, Act& o) {     // dense_pinned's epilogue slice
        int nw = pack_pair<f16x2, RELU>(v[2 * j], v[2 * j + 1]);
        NRF_PIN_ACT1(nw);
        i32x4 w = __builtin_bit_cast(i32x4, o.f[j >> 2]);
        w[j & 3] = nw;
        if ((j & 3) == 3) NRF_PARK_ACT(w);
        o.f[j >> 2] = __builtin_bit_cast(f16x8, w);
    }
};

struct ModeF32 {
    static constexpr int SUB = 4;
    static constexpr bool kPinned = false;   // 64-cycle MFMAs: the epilogue is 1 % of a tile
    static constexpr int TRIG = 0;           // ocml sincosf on the exact argument
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


// Split-precision f16 ("f16x3"): every operand is carried as hi + lo with hi = f16(x), lo = f16(x - hi) -- 22 significant
// bits -- and a product is three MFMAs at the 32x32x16 rate: W_hi*X_hi + W_hi*X_lo + W_lo*X_hi (the dropped lo*lo term is
// 2^-22 relative; every f16 x f16 product is exact in the fp32 accumulator).  This is the parity-grade FAST mode: fp32-class
// results (1e-4 bar of BASELINE.json) at up to 1/3 of the 16-bit MFMA rate instead of the fp32 MFMA's 1/16.
// The low parts of small values are f16 subnormals; gfx950 keeps them in the conversion and in the matrix core
// (tools/f16_subnormal_probe.hip).  Operands are clamped to the f16 range (+-65504): activations beyond it would need bf16x3.
// Stream: SUB = 4 fragments per (m-tile, k-tile) = [s0 hi, s0 lo, s1 hi, s1 lo] (packing.cpp); an even fragment (W_hi) feeds
// two MFMAs (X_hi, X_lo), an odd one (W_lo) one (X_hi).  Activations take 16 registers per 32-feature tile, as in the fp32
// mode, so the geometry is the fp32 mode's: 4 waves x 32 samples, one wave per SIMD.
struct ModeF16X3 {
    static constexpr int SUB = 4;
    static constexpr bool kPinned = true;    // one wave per SIMD: the step order and the epilogue slices are pinned (dense_pinned)
    static constexpr int TRIG = 2;           // polynomial sin on exactly reduced turns: <= 2e-7 abs (nets.hpp:encode3)
    static constexpr bool FAST_EXP = false;
    typedef f16x8 frag_t;
    struct Act { f16x8 hi[2], lo[2]; };
    __device__ static __forceinline__ void mma(f32x16& acc, const frag_t& a, const Act& b, int q) {
        const int s = q >> 1;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b.hi[s], acc, 0, 0, 0);
        if (!(q & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b.lo[s], acc, 0, 0, 0);
    }
    template <bool RELU>
    __device__ static __forceinline__ Act to_act(const f32x16& v) {
        Act o;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            i32x4 wh, wl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // v_med3_f32: ReLU and the f16 range clamp in one instruction
                const float a = __builtin_amdgcn_fmed3f(v[8 * s + 2 * j], RELU ? 0.0f : -65504.0f, 65504.0f);
                const float b = __builtin_amdgcn_fmed3f(v[8 * s + 2 * j + 1], RELU ? 0.0f : -65504.0f, 65504.0f);
                const f32x2 ab = {a, b};
                const f16x2 h = __builtin_convertvector(ab, f16x2);
                const f32x2 hf = __builtin_convertvector(h, f32x2);
                const f32x2 rest = {__fsub_rn(a, hf[0]), __fsub_rn(b, hf[1])};      // exact
                wh[j] = __builtin_bit_cast(int, h);
                wl[j] = __builtin_bit_cast(int, __builtin_convertvector(rest, f16x2));
            }
            o.hi[s] = __builtin_bit_cast(f16x8, wh);
            o.lo[s] = __builtin_bit_cast(f16x8, wl);
        }
        return o;
    }
    // one register pair (accumulator registers 2j, 2j+1) of a tile -> word j of its hi / lo operand images: the epilogue in
    // eight slices, so that dense_pinned() can hand one slice to each of the next tile's first MFMA steps
    template <bool RELU>
    __device__ static __forceinline__ void to_act_pair(const f32x16& v, int j, Act& o) {
        const int s = j >> 2, w = j & 3;
        const float a = __builtin_amdgcn_fmed3f(v[2 * j], RELU ? 0.0f : -65504.0f, 65504.0f);
        const float b = __builtin_amdgcn_fmed3f(v[2 * j + 1], RELU ? 0.0f : -65504.0f, 65504.0f);
        const f32x2 ab = {a, b};
        const f16x2 hh = __builtin_convertvector(ab, f16x2);
        const f32x2 hf = __builtin_convertvector(hh, f32x2);
        const f32x2 rest = {__fsub_rn(a, hf[0]), __fsub_rn(b, hf[1])};
        int nh = __builtin_bit_cast(int, hh), nl = __builtin_bit_cast(int, __builtin_convertvector(rest, f16x2));
        // the words are first USED by the next layer: without this (empty, opaque) statement LLVM sinks the whole slice down to
        // that use, across the step fences, and the epilogues of several tiles pile up between two MFMAs again
        NRF_PIN_ACT2(nh, nl);
        i32x4 wh = __builtin_bit_cast(i32x4, o.hi[s]), wl = __builtin_bit_cast(i32x4, o.lo[s]);
        wh[w] = nh;
        wl[w] = nl;
        o.hi[s] = __builtin_bit_cast(f16x8, wh);
        o.lo[s] = __builtin_bit_cast(f16x8, wl);
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

// One Linear layer: out[m][n] (32 features x 32 samples each) = W * in + bias, for MT output tiles, KT input
// tiles, NT sample tiles.  Consumes MT*KT*SUB fragments of the stream (the host pads every layer to whole
// chunks, so a layer always starts on a chunk boundary).  `fin(m, acc)` receives each finished tile's raw
// accumulators.  Fragment reads are software-pipelined kPrefetch deep (LDS latency would otherwise sit in front
// of every MFMA) and the next chunk is acquired kPrefetch fragments before the current one is used up, so
// the read stream never stops inside a layer.
constexpr __host__ __device__ int chunks_for_frags(int frags) { return (frags + kChunkFrags - 1) / kChunkFrags; }

// Which layer walk a geometry uses: one wave per SIMD -- the fp32-class split mode, and any mode at NT = 2 sample tiles per wave
// (64 columns: 128 + 128 operand registers and 64 accumulators, i.e. the whole 512-entry file) -- takes dense_pinned() and issues
// its LDS-DMA as inline asm; two waves per SIMD (16-bit modes at NT = 1) and the fp32 MFMA mode take dense().
template <class Mode, int NT>
constexpr __host__ __device__ bool pinned_walk() { return Mode::kPinned || NT > 1; }

#ifndef NRF_PREFETCH
#define NRF_PREFETCH 3
#endif
#ifndef NRF_EPILOGUE_AT
#define NRF_EPILOGUE_AT 3
#endif
constexpr int kPrefetch = NRF_PREFETCH;
constexpr int kEpilogueAt = NRF_EPILOGUE_AT;   // the epilogue of tile m-1 runs after this many fragments of tile m have been issued

template <class Mode, int KT, int MT, int NT, class P, class Fin>
__device__ __forceinline__ void dense(P& pipe, const NRF_LDS float* bias, int h,
                                      const typename Mode::Act (&in)[KT][NT], Fin&& fin) {
    constexpr int PER_M = KT * Mode::SUB;
    constexpr int NF = MT * PER_M;
    constexpr int PF = NF < kPrefetch ? NF : kPrefetch;
    constexpr int EPI = PER_M > kEpilogueAt ? kEpilogueAt : PER_M - 1;
    typedef typename Mode::frag_t frag_t;
    frag_t fr[PF];
    auto read = [&](auto g_) -> frag_t {
        constexpr int g = decltype(g_)::value;
        if constexpr (g % kChunkFrags == 0) pipe.acquire((g / kChunkFrags) & 1);
        return *(const NRF_LDS frag_t*)(pipe.base[(g / kChunkFrags) & 1] + (g % kChunkFrags) * kFragBytes);
    };
    static_for<PF>([&](auto i_) { fr[decltype(i_)::value] = read(i_); });
    // two accumulator sets: while tile m accumulates into acc[m&1], the finished tile m-1 sits in acc[(m-1)&1]
    // until its MFMAs have drained (no s_nop bubble), is handed to fin(), and the set is re-armed with the
    // bias of tile m+1 -- so neither the epilogue nor the bias reads sit between two MFMAs
    f32x16 acc[2][NT];
    load_bias(acc[0][0], bias, h);
#pragma unroll
    for (int n = 1; n < NT; ++n) acc[0][n] = acc[0][0];
    static_for<NF>([&](auto f_) {
        constexpr int f = decltype(f_)::value;
        constexpr int m = f / PER_M, rem = f % PER_M, t = rem / Mode::SUB, s = rem % Mode::SUB;
        const frag_t a = fr[f % PF];
#pragma unroll
        for (int n = 0; n < NT; ++n) Mode::mma(acc[m & 1][n], a, in[t][n], s);
        if constexpr (f + PF < NF) fr[f % PF] = read(std::integral_constant<int, f + PF>{});
        if constexpr (rem == EPI) {
            if constexpr (m > 0) fin(std::integral_constant<int, m - 1>{}, acc[(m - 1) & 1]);
            if constexpr (m + 1 < MT) {
                load_bias(acc[(m + 1) & 1][0], bias + 32 * (m + 1), h);
#pragma unroll
                for (int n = 1; n < NT; ++n) acc[(m + 1) & 1][n] = acc[(m + 1) & 1][0];
            }
        }
    });
    fin(std::integral_constant<int, MT - 1>{}, acc[(MT - 1) & 1]);
}

// The same layer for the geometries that run ONE wave per SIMD (pinned_walk<Mode, NT>()).  There nothing covers what the wave itself does not
// overlap, and hipcc's schedule does two things that cost a quarter of the MFMA pipe: it sinks every fragment read to just in
// front of its MFMA, and it gathers the epilogues of several tiles into single blocks of ~280 VALU instructions between two
// MFMAs.  Here every fragment step is fenced (__builtin_amdgcn_sched_barrier): step f = the MFMAs of fragment f, the read of
// fragment f + PF, and ONE slice (a register pair) of the previous tile's epilogue, `fin(m, acc, j)`, j = 0..7 -- so the VALU work
// rides in the shadow of the MFMAs (an MFMA holds the issue port for 8 of its 32 cycles).  The bias of the tile after next is
// fetched in the step behind the last slice (its accumulator set is free from then on).  With the LDS-DMA issued as inline asm
// (Pipe<..., ASM_DMA>) hipcc's waitcnt pass sees a pure in-order ds_read stream and emits counted lgkmcnt waits.
//
// Chained layers.  What is left exposed is the layer boundary: the eight slices of the LAST tile (nothing behind them to hide in),
// the first fragment reads and the first bias fetch of the next layer (LDS latency with an empty MFMA queue).  A layer can
// therefore hand a Carry to the next one (COUT / CIN):
//   kCarryPre : the next layer's first PF fragments (read during this layer's last PF steps -- the stream is continuous, the
//               next layer starts on the next chunk) and its tile-0 bias (fetched where a tile m+1 bias would be);
//   kCarryTail: additionally the last tile's RAW accumulators; the next layer converts them, `tail(j)`, in the slice steps of
//               its tile 0 -- the operand tile they complete is first read (KT-1)*SUB steps in, long after.
enum { kCarryNone = 0, kCarryPre = 1, kCarryTail = 2 };

template <class Mode, int NT>
struct Carry {
    f32x16 tail[NT];
    f32x16 bias0;
    typename Mode::frag_t fr[kPrefetch];
};

template <class Mode, int KT, int MT, int NT, int CIN, int COUT, int TAIL_TILE, class P, class FinSlice, class TailSlice>
__device__ __forceinline__ void dense_pinned(P& pipe, const NRF_LDS float* bias, const NRF_LDS float* bias_next, int h,
                                             const typename Mode::Act (&in)[KT][NT], FinSlice&& fin, TailSlice&& tail, Carry<Mode, NT>& carry) {
    constexpr int PER_M = KT * Mode::SUB;
    constexpr int NF = MT * PER_M;
    constexpr int PF = kPrefetch;
    static_assert(NF >= PF, "layer shorter than the fragment prefetch");
    constexpr int EPI = PER_M > kEpilogueAt + 1 ? kEpilogueAt : 1;      // first slice step: tile m-1's last MFMAs have drained
    constexpr int ROOM = PER_M - EPI - 1;                               // steps that may carry slices (one more is the bias step)
    static_assert(ROOM >= 1, "layer too short for the sliced epilogue");
    constexpr int SPS = (8 + ROOM - 1) / ROOM;                          // slices per step
    constexpr int NSTEP = (8 + SPS - 1) / SPS;
    constexpr int BIAS_AT = EPI + NSTEP;
    static_assert(BIAS_AT < PER_M, "no step left for the bias fetch");
    static_assert(CIN != kCarryTail || EPI + NSTEP <= TAIL_TILE * Mode::SUB, "the carried tile is read before its slices are done");
    typedef typename Mode::frag_t frag_t;
    frag_t fr[PF], nxt[PF];
    // a layer's last chunk must see all of its DMA parts issued (they ride with the reads of its first fragments)
    static_assert((NF % kChunkFrags == 0 ? kChunkFrags : NF % kChunkFrags) >= P::kFragsPerWave, "last chunk of the layer too short for the spread LDS-DMA issue");
    auto read = [&](auto g_) -> frag_t {
        constexpr int g = decltype(g_)::value;
        if constexpr (g % kChunkFrags == 0) pipe.acquire_begin((g / kChunkFrags) & 1);
        else if constexpr (g % kChunkFrags < P::kFragsPerWave) pipe.template issue_part<g % kChunkFrags>();
        return *(const NRF_LDS frag_t*)(pipe.base[(g / kChunkFrags) & 1] + (g % kChunkFrags) * kFragBytes);
    };
    f32x16 acc[2][NT];
    if constexpr (CIN != kCarryNone) {
        static_for<PF>([&](auto i_) { fr[decltype(i_)::value] = carry.fr[decltype(i_)::value]; });
        acc[0][0] = carry.bias0;
    } else {
        static_for<PF>([&](auto i_) { fr[decltype(i_)::value] = read(i_); });
        load_bias(acc[0][0], bias, h);
    }
#pragma unroll
    for (int n = 1; n < NT; ++n) acc[0][n] = acc[0][0];
    f32x16 bias0_next;
    __builtin_amdgcn_sched_barrier(0);
    static_for<NF>([&](auto f_) {
        constexpr int f = decltype(f_)::value;
        constexpr int m = f / PER_M, rem = f % PER_M, t = rem / Mode::SUB, s = rem % Mode::SUB;
        const frag_t a = fr[f % PF];
#pragma unroll
        for (int n = 0; n < NT; ++n) Mode::mma(acc[m & 1][n], a, in[t][n], s);
        if constexpr (f + PF < NF) fr[f % PF] = read(std::integral_constant<int, f + PF>{});
        else if constexpr (COUT != kCarryNone) nxt[f + PF - NF] = read(std::integral_constant<int, f + PF - NF>{});   // the next layer's fragment: its chunk 0, parity 0
        if constexpr ((m > 0 || CIN == kCarryTail) && rem >= EPI && rem < EPI + NSTEP) {
            static_for<SPS>([&](auto k_) {
                constexpr int j = (rem - EPI) * SPS + decltype(k_)::value;
                if constexpr (j < 8) {
                    if constexpr (m > 0) fin(std::integral_constant<int, m - 1>{}, acc[(m - 1) & 1], std::integral_constant<int, j>{});
                    else tail(carry.tail, std::integral_constant<int, j>{});
                }
            });
        }
        if constexpr (rem == BIAS_AT) {
            if constexpr (m + 1 < MT) {
                load_bias(acc[(m + 1) & 1][0], bias + 32 * (m + 1), h);
#pragma unroll
                for (int n = 1; n < NT; ++n) acc[(m + 1) & 1][n] = acc[(m + 1) & 1][0];
            } else if constexpr (COUT != kCarryNone) {
                load_bias(bias0_next, bias_next, h);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (COUT != kCarryNone) {
        static_for<PF>([&](auto i_) { carry.fr[decltype(i_)::value] = nxt[decltype(i_)::value]; });
        carry.bias0 = bias0_next;
    }
    if constexpr (COUT == kCarryTail) {
#pragma unroll
        for (int n = 0; n < NT; ++n) carry.tail[n] = acc[(MT - 1) & 1][n];
    } else {
        static_for<8>([&](auto j_) { fin(std::integral_constant<int, MT - 1>{}, acc[(MT - 1) & 1], j_); });
    }
}

// layer with an activation, producing the next layer's operand tiles.  Chained form: `carry` links it to its neighbours in a
// pinned walk (ignored -- every layer complete in itself -- by the two-waves-per-SIMD and fp32 walks).  TAIL_RELU / TAIL_TILE: the
// activation of the layer that left its last tile in `carry`, and the operand tile of `in` that tile completes.
template <class Mode, int KT, int MT, int NT, bool RELU, int CIN, int COUT, bool TAIL_RELU = true, int TAIL_TILE = KT - 1, class P>
__device__ __forceinline__ void dense_act_chain(P& pipe, const NRF_LDS float* bias, const NRF_LDS float* bias_next, int h,
                                                typename Mode::Act (&in)[KT][NT], typename Mode::Act (&out)[MT][NT], Carry<Mode, NT>& carry) {
    if constexpr (pinned_walk<Mode, NT>()) {
        dense_pinned<Mode, KT, MT, NT, CIN, COUT, TAIL_TILE>(pipe, bias, bias_next, h, in,
            [&](auto m_, f32x16(&acc)[NT], auto j_) {
                constexpr int m = decltype(m_)::value;
#pragma unroll
                for (int n = 0; n < NT; ++n) Mode::template to_act_pair<RELU>(acc[n], decltype(j_)::value, out[m][n]);
            },
            [&](f32x16(&tl)[NT], auto j_) {
#pragma unroll
                for (int n = 0; n < NT; ++n) Mode::template to_act_pair<TAIL_RELU>(tl[n], decltype(j_)::value, in[TAIL_TILE][n]);
            }, carry);
    } else {
        dense<Mode, KT, MT, NT>(pipe, bias, h, in, [&](auto m_, f32x16(&acc)[NT]) {
            constexpr int m = decltype(m_)::value;
#pragma unroll
            for (int n = 0; n < NT; ++n) out[m][n] = Mode::template to_act<RELU>(acc[n]);
        });
    }
}

template <class Mode, int KT, int MT, int NT, bool RELU, class P>
__device__ __forceinline__ void dense_act(P& pipe, const NRF_LDS float* bias, int h,
                                          const typename Mode::Act (&in)[KT][NT], typename Mode::Act (&out)[MT][NT]) {
    typedef typename Mode::Act Act;
    Carry<Mode, NT> none;      // unchained: `in` is only read
    dense_act_chain<Mode, KT, MT, NT, RELU, kCarryNone, kCarryNone>(pipe, bias, bias, h, const_cast<Act(&)[KT][NT]>(in), out, none);
}

// head layer: one output tile, raw accumulators back to the caller
template <class Mode, int KT, int NT, int CIN, int COUT, bool TAIL_RELU = true, int TAIL_TILE = KT - 1, class P>
__device__ __forceinline__ void dense_head_chain(P& pipe, const NRF_LDS float* bias, const NRF_LDS float* bias_next, int h,
                                                 typename Mode::Act (&in)[KT][NT], f32x16 (&out)[NT], Carry<Mode, NT>& carry) {
    static_assert(COUT != kCarryTail, "a head hands its tile to the caller");
    if constexpr (pinned_walk<Mode, NT>()) {
        dense_pinned<Mode, KT, 1, NT, CIN, COUT, TAIL_TILE>(pipe, bias, bias_next, h, in,
            [&](auto, f32x16(&acc)[NT], auto j_) {
                constexpr int j = decltype(j_)::value;
#pragma unroll
                for (int n = 0; n < NT; ++n) { out[n][2 * j] = acc[n][2 * j]; out[n][2 * j + 1] = acc[n][2 * j + 1]; }
            },
            [&](f32x16(&tl)[NT], auto j_) {
#pragma unroll
                for (int n = 0; n < NT; ++n) Mode::template to_act_pair<TAIL_RELU>(tl[n], decltype(j_)::value, in[TAIL_TILE][n]);
            }, carry);
    } else {
        dense<Mode, KT, 1, NT>(pipe, bias, h, in, [&](auto, f32x16(&acc)[NT]) {
#pragma unroll
            for (int n = 0; n < NT; ++n) out[n] = acc[n];
        });
    }
}

template <class Mode, int KT, int NT, class P>
__device__ __forceinline__ void dense_head(P& pipe, const NRF_LDS float* bias, int h,
                                           const typename Mode::Act (&in)[KT][NT], f32x16 (&out)[NT]) {
    typedef typename Mode::Act Act;
    Carry<Mode, NT> none;
    dense_head_chain<Mode, KT, NT, kCarryNone, kCarryNone>(pipe, bias, bias, h, const_cast<Act(&)[KT][NT]>(in), out, none);
}

constexpr __host__ __device__ int chunks_for(int frags) { return (frags + kChunkFrags - 1) / kChunkFrags; }

}  // namespace nrf

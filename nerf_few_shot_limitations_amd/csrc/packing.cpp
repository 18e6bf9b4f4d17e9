// packing.cpp -- host-side weight packer (see packing.hpp / mlp_core.hpp).
#include "packing.hpp"

#include <cmath>
#include <cstring>

#include "feature_map.hpp"
#include "stream_config.hpp"

namespace nrf {

namespace {
constexpr int kFragBytes = 1024, kChunkFrags = NRF_CHUNK_FRAGS;

uint32_t f32_bits(float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; }
}  // namespace

uint16_t f32_to_bf16(float x) {                      // round to nearest even, NaN stays NaN
    uint32_t u = f32_bits(x);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

uint16_t f32_to_f16(float x) {                       // round to nearest even, subnormals kept, overflow -> inf
    const uint32_t u = f32_bits(x);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t e = (u >> 23) & 0xFFu;
    uint32_t m = u & 0x007FFFFFu;
    if (e == 0xFFu) return (uint16_t)(sign | 0x7C00u | (m ? 0x0200u : 0u));
    int32_t ne = (int32_t)e - 127 + 15;
    if (ne >= 31) return (uint16_t)(sign | 0x7C00u);
    if (ne <= 0) {
        if (ne < -10) return (uint16_t)sign;
        m |= 0x00800000u;
        const int shift = 14 - ne;                   // 14..24
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) ++hm;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = ((uint32_t)ne << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;   // may carry into the exponent: still correct
    return (uint16_t)(sign | h);
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) { u = sign; }
        else {                                         // subnormal: value = m * 2^-24
            const float v = (float)m * 5.9604644775390625e-8f;
            std::memcpy(&u, &v, 4);
            u |= sign;
        }
    } else if (e == 31) {
        u = sign | 0x7F800000u | (m << 13);
    } else {
        u = sign | ((e - 15 + 127) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

int expected_linears(const nrf_arch& a) {
    switch (a.net) {
        case NRF_NET_V1: return a.n_layers + 2;
        case NRF_NET_V2: return a.n_layers + 5;
        case NRF_NET_V3: return a.n_layers + 10;
        default: return 0;
    }
}

namespace {

std::vector<int> identity_cols(int n) {
    std::vector<int> c(n);
    for (int i = 0; i < n; ++i) c[i] = i;
    return c;
}

// columns of a Linear fed by PE(L): kernel K index -> reference feature index (+ base)
void pe_cols(std::vector<int>& col, int k0, int L, int base) {
    const int KT = pe_tiles(L);
    for (int k = 0; k < 32 * KT; ++k) {
        const int idx = pe_ref_index(L, k_slot(k), k_half(k));
        col[k0 + k] = idx < 0 ? -1 : base + idx;
    }
}

std::vector<std::pair<int, int>> rows_of(int lin, int n_rows, int MT) {
    std::vector<std::pair<int, int>> r(32 * MT, {-1, 0});
    for (int i = 0; i < n_rows; ++i) r[i] = {lin, i};
    return r;
}

bool check(const std::vector<HostLinear>& lin, int i, int out_f, int in_f, const char* name, std::string& err) {
    if (lin[i].out_f != out_f || lin[i].in_f != in_f) {
        err = std::string("linear '") + name + "' (index " + std::to_string(i) + ") has shape (" +
              std::to_string(lin[i].out_f) + "," + std::to_string(lin[i].in_f) + "), expected (" +
              std::to_string(out_f) + "," + std::to_string(in_f) + ")";
        return false;
    }
    return true;
}

}  // namespace

bool make_plan(const nrf_arch& a, const std::vector<HostLinear>& lin, NetPlan& plan, std::string& err) {
    plan = NetPlan();
    if (a.hidden != 256) { err = "only hidden=256 is built (every experiments/*.yaml uses 256)"; return false; }
    if (a.n_layers < 1 || a.n_layers > 16) { err = "n_layers must be in 1..16"; return false; }
    if (a.pos_freq < 1 || a.pos_freq > 15) { err = "pos_freq must be in 1..15"; return false; }
    if ((int)lin.size() != expected_linears(a)) {
        err = "expected " + std::to_string(expected_linears(a)) + " Linear layers, got " + std::to_string(lin.size());
        return false;
    }
    const int H = a.hidden, HT = H / 32;
    const int pe = pe_dim(a.pos_freq), peT = pe_tiles(a.pos_freq);
    auto add = [&](LayerPlan&& L) { plan.layers.push_back(std::move(L)); };

    if (a.net == NRF_NET_V1) {
        const int n = a.n_layers;
        for (int i = 0; i < n; ++i)
            if (!check(lin, i, H, i == 0 ? pe : H, "layers.N", err)) return false;
        if (!check(lin, n, 1, H, "sigma_out", err) || !check(lin, n + 1, 3, H, "rgb_out", err)) return false;
        LayerPlan L0; L0.KT = peT; L0.MT = HT; L0.col.assign(32 * peT, -1); pe_cols(L0.col, 0, a.pos_freq, 0);
        L0.row = rows_of(0, H, HT); add(std::move(L0));
        for (int i = 1; i < n; ++i) {
            LayerPlan L; L.KT = HT; L.MT = HT; L.col = identity_cols(H); L.row = rows_of(i, H, HT); add(std::move(L));
        }
        // head tile: rows 0..2 = rgb_out, row 3 = sigma_out, rows 4..7 repeat them so that BOTH lane
        // halves find [r,g,b,sigma] in accumulator registers 0..3
        LayerPlan Hd; Hd.KT = HT; Hd.MT = 1; Hd.col = identity_cols(H); Hd.row.assign(32, {-1, 0});
        for (int rep = 0; rep < 2; ++rep) {
            for (int c = 0; c < 3; ++c) Hd.row[4 * rep + c] = {n + 1, c};
            Hd.row[4 * rep + 3] = {n, 0};
        }
        add(std::move(Hd));
    } else if (a.net == NRF_NET_V2) {
        if (a.dir_freq < 1 || 3 * a.dir_freq + 2 > 16) { err = "dir_freq must be in 1..4 (one operand tile)"; return false; }
        const int n = a.n_layers, de = pe_dim(a.dir_freq);
        for (int i = 0; i < n; ++i)
            if (!check(lin, i, H, i == 0 ? pe : H, "density_layers.N", err)) return false;
        if (!check(lin, n, 1, H, "density_head", err) || !check(lin, n + 1, H, H, "feature_head", err) ||
            !check(lin, n + 2, H / 2, H + de, "color_layers.0", err) || !check(lin, n + 3, H / 4, H / 2, "color_layers.2", err) ||
            !check(lin, n + 4, 3, H / 4, "color_layers.4", err))
            return false;
        LayerPlan L0; L0.KT = peT; L0.MT = HT; L0.col.assign(32 * peT, -1); pe_cols(L0.col, 0, a.pos_freq, 0);
        L0.row = rows_of(0, H, HT); add(std::move(L0));
        for (int i = 1; i < n; ++i) {
            LayerPlan L; L.KT = HT; L.MT = HT; L.col = identity_cols(H); L.row = rows_of(i, H, HT); add(std::move(L));
        }
        LayerPlan Dh; Dh.KT = HT; Dh.MT = 1; Dh.col = identity_cols(H); Dh.row.assign(32, {-1, 0});
        Dh.row[0] = {n, 0}; Dh.row[4] = {n, 0}; add(std::move(Dh));                    // density in reg 0 of both halves
        LayerPlan Fh; Fh.KT = HT; Fh.MT = HT; Fh.col = identity_cols(H); Fh.row = rows_of(n + 1, H, HT); add(std::move(Fh));
        LayerPlan C0; C0.KT = HT + 1; C0.MT = H / 64; C0.col.assign(32 * (HT + 1), -1);
        for (int k = 0; k < H; ++k) C0.col[k] = k;
        pe_cols(C0.col, H, a.dir_freq, H);
        C0.row = rows_of(n + 2, H / 2, H / 64); add(std::move(C0));
        LayerPlan C2; C2.KT = H / 64; C2.MT = H / 128; C2.col = identity_cols(H / 2); C2.row = rows_of(n + 3, H / 4, H / 128); add(std::move(C2));
        LayerPlan C4; C4.KT = H / 128; C4.MT = 1; C4.col = identity_cols(H / 4); C4.row.assign(32, {-1, 0});
        for (int rep = 0; rep < 2; ++rep)
            for (int c = 0; c < 3; ++c) C4.row[4 * rep + c] = {n + 4, c};
        add(std::move(C4));
    } else if (a.net == NRF_NET_V3) {
        // lora_dino.py:171-193 + nerf_mlp.py:144-158.  Stream order = the order nets.hpp:NetV3 walks:
        // fusion.0, fusion.2, attention.0, attention.2, fusion.0, fusion.2 (the SAME weights, second pass),
        // output_proj, trunk, density_head, feature_head, colour layers.
        if (a.dir_freq < 1 || 3 * a.dir_freq + 2 > 16) { err = "dir_freq must be in 1..4 (one operand tile)"; return false; }
        if (a.dino_dim != 64 && a.dino_dim != 128) { err = "dino_dim must be 64 (single-scale) or 128 (multi-scale)"; return false; }
        const int n = a.n_layers, de = pe_dim(a.dir_freq), DT = a.dino_dim / 32;
        if (!check(lin, 0, H, pe + a.dino_dim, "dino_fusion.fusion.0", err) || !check(lin, 1, H, H, "dino_fusion.fusion.2", err) ||
            !check(lin, 2, H / 4, H, "dino_fusion.attention.0", err) || !check(lin, 3, 2, H / 4, "dino_fusion.attention.2", err) ||
            !check(lin, 4, H, H, "dino_fusion.output_proj", err))
            return false;
        const int b = 5;
        for (int i = 0; i < n; ++i)
            if (!check(lin, b + i, H, H, "density_layers.N", err)) return false;
        if (!check(lin, b + n, 1, H, "density_head", err) || !check(lin, b + n + 1, H, H, "feature_head", err) ||
            !check(lin, b + n + 2, H / 2, H + de, "color_layers.0", err) || !check(lin, b + n + 3, H / 4, H / 2, "color_layers.2", err) ||
            !check(lin, b + n + 4, 3, H / 4, "color_layers.4", err))
            return false;
        auto fusion0 = [&]() {
            LayerPlan L; L.KT = peT + DT; L.MT = HT; L.col.assign(32 * (peT + DT), -1);
            pe_cols(L.col, 0, a.pos_freq, 0);
            for (int k = 0; k < a.dino_dim; ++k) L.col[32 * peT + k] = pe + k;          // channel k of the fetched feature
            L.row = rows_of(0, H, HT);
            return L;
        };
        auto square = [&](int li) { LayerPlan L; L.KT = HT; L.MT = HT; L.col = identity_cols(H); L.row = rows_of(li, H, HT); return L; };
        add(fusion0()); add(square(1));
        LayerPlan A0; A0.KT = HT; A0.MT = H / 128; A0.col = identity_cols(H); A0.row = rows_of(2, H / 4, H / 128); add(std::move(A0));
        LayerPlan A2; A2.KT = H / 128; A2.MT = 1; A2.col = identity_cols(H / 4); A2.row.assign(32, {-1, 0});
        for (int rep = 0; rep < 2; ++rep) { A2.row[4 * rep + 0] = {3, 0}; A2.row[4 * rep + 1] = {3, 1}; }   // logits in regs 0,1 of both lane halves
        add(std::move(A2));
        add(fusion0()); add(square(1));
        add(square(4));
        for (int i = 0; i < n; ++i) add(square(b + i));
        LayerPlan Dh; Dh.KT = HT; Dh.MT = 1; Dh.col = identity_cols(H); Dh.row.assign(32, {-1, 0});
        Dh.row[0] = {b + n, 0}; Dh.row[4] = {b + n, 0}; add(std::move(Dh));
        add(square(b + n + 1));
        LayerPlan C0; C0.KT = HT + 1; C0.MT = H / 64; C0.col.assign(32 * (HT + 1), -1);
        for (int k = 0; k < H; ++k) C0.col[k] = k;
        pe_cols(C0.col, H, a.dir_freq, H);
        C0.row = rows_of(b + n + 2, H / 2, H / 64); add(std::move(C0));
        LayerPlan C2; C2.KT = H / 64; C2.MT = H / 128; C2.col = identity_cols(H / 2); C2.row = rows_of(b + n + 3, H / 4, H / 128); add(std::move(C2));
        LayerPlan C4; C4.KT = H / 128; C4.MT = 1; C4.col = identity_cols(H / 4); C4.row.assign(32, {-1, 0});
        for (int rep = 0; rep < 2; ++rep)
            for (int c = 0; c < 3; ++c) C4.row[4 * rep + c] = {b + n + 4, c};
        add(std::move(C4));
    } else {
        err = "unknown network family";
        return false;
    }

    int off = 0;
    for (auto& L : plan.layers) { L.bias_off = off; off += 32 * L.MT; }
    plan.n_bias = off;
    int64_t mac = 0;
    for (const auto& l : lin) mac += (int64_t)l.out_f * l.in_f;
    if (a.net == NRF_NET_V3) mac += (int64_t)lin[0].out_f * lin[0].in_f + (int64_t)lin[1].out_f * lin[1].in_f;   // fusion runs twice
    plan.flops_per_sample = 2 * mac;
    return true;
}

std::vector<float> pack_bias(const NetPlan& plan, const std::vector<HostLinear>& lin) {
    std::vector<float> b(plan.n_bias, 0.0f);
    for (const auto& L : plan.layers)
        for (int r = 0; r < 32 * L.MT; ++r)
            if (L.row[r].first >= 0) b[L.bias_off + r] = lin[L.row[r].first].b[L.row[r].second];
    return b;
}

ParamLayout param_layout(const std::vector<HostLinear>& lin) {
    ParamLayout lay;
    int64_t off = 0;
    for (const auto& l : lin) {
        lay.w_off.push_back(off); off += (int64_t)l.out_f * l.in_f;
        lay.b_off.push_back(off); off += l.out_f;
        lay.in_f.push_back(l.in_f);
    }
    lay.total = off;
    return lay;
}

namespace {
// flat offset of element (output row r, K index k) of a packed layer, -1 = zero
int64_t element_source(const LayerPlan& L, const ParamLayout& lay, int r, int k) {
    if (L.transposed) {
        const auto& ks = L.krow[k];
        if (ks.first < 0 || L.rcol[r] < 0) return -1;
        return lay.w_off[ks.first] + (int64_t)ks.second * lay.in_f[ks.first] + L.rcol[r];
    }
    const auto& rs = L.row[r];
    if (rs.first < 0 || L.col[k] < 0) return -1;
    return lay.w_off[rs.first] + (int64_t)rs.second * lay.in_f[rs.first] + L.col[k];
}
}  // namespace

std::vector<int32_t> stream_sources(const NetPlan& plan, const ParamLayout& lay, int kind) {
    const bool f32 = kind == kStreamF32, x3 = kind == kStreamX3;
    const int SUB = (f32 || x3) ? 4 : 2, n_el = f32 ? 4 : 8;
    size_t total_frags = 0;
    for (const auto& L : plan.layers) {
        const size_t f = (size_t)L.MT * L.KT * SUB;
        total_frags += (f + kChunkFrags - 1) / kChunkFrags * kChunkFrags;
    }
    std::vector<int32_t> src(total_frags * 64 * n_el, -1);
    size_t frag = 0;
    for (const auto& L : plan.layers) {
        for (int m = 0; m < L.MT; ++m)
            for (int t = 0; t < L.KT; ++t)
                for (int s = 0; s < SUB; ++s, ++frag)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, h = lane >> 5;
                        for (int e = 0; e < n_el; ++e) {
                            // split mode: fragments 2s' (hi parts) and 2s'+1 (lo parts) hold the same K indices as the 16-bit
                            // fragment s' (mlp_core.hpp:ModeF16X3)
                            const int s16 = x3 ? (s >> 1) : s;
                            const int k = f32 ? (32 * t + 8 * s + 4 * h + e)
                                              : (32 * t + 16 * s16 + 8 * (e >> 2) + 4 * h + (e & 3));
                            src[(frag * 64 + lane) * n_el + e] = (int32_t)element_source(L, lay, 32 * m + i, k);
                        }
                    }
        frag = (frag + kChunkFrags - 1) / kChunkFrags * kChunkFrags;   // every layer starts on a chunk boundary
    }
    return src;
}

std::vector<int32_t> bias_sources(const NetPlan& plan, const ParamLayout& lay) {
    std::vector<int32_t> b(plan.n_bias, -1);
    for (const auto& L : plan.layers) {
        if (L.transposed) continue;
        for (int r = 0; r < 32 * L.MT; ++r)
            if (L.row[r].first >= 0) b[L.bias_off + r] = (int32_t)(lay.b_off[L.row[r].first] + L.row[r].second);
    }
    return b;
}

PackedStream pack_stream(const NetPlan& plan, const std::vector<HostLinear>& lin, int mode) {
    const bool f32 = (mode == NRF_MMA_F32), x3 = (mode == NRF_MMA_F16X3);
    const ParamLayout lay = param_layout(lin);
    std::vector<float> flat((size_t)lay.total);
    for (size_t i = 0; i < lin.size(); ++i) {
        std::memcpy(flat.data() + lay.w_off[i], lin[i].w.data(), lin[i].w.size() * sizeof(float));
        std::memcpy(flat.data() + lay.b_off[i], lin[i].b.data(), lin[i].b.size() * sizeof(float));
    }
    const std::vector<int32_t> src = stream_sources(plan, lay, stream_kind(mode));
    PackedStream out;
    const size_t per_frag = f32 ? 256 : 512;
    out.n_chunks = (uint32_t)(src.size() / per_frag / kChunkFrags);
    out.bytes.assign(src.size() / per_frag * kFragBytes, 0);
    for (size_t i = 0; i < src.size(); ++i) {
        float v = src[i] >= 0 ? flat[src[i]] : 0.0f;
        // f16-typed streams saturate at the largest finite f16: an out-of-range weight would otherwise become inf (and the split
        // mode's low part -inf, their products NaN).  No trained NeRF comes near it; a diverged checkpoint renders saturated, not NaN.
        if (!f32 && mode != NRF_MMA_BF16) v = std::fmin(std::fmax(v, -65504.0f), 65504.0f);
        if (f32) {
            std::memcpy(out.bytes.data() + i * 4, &v, 4);
        } else if (x3) {
            // element i sits in fragment i/512; every layer starts on a chunk (16 fragments) boundary and holds its hi/lo
            // fragments alternately, so the parity of the fragment index tells the part
            const uint16_t hi = f32_to_f16(v);
            uint16_t q = hi;
            if ((i / per_frag) & 1) q = f32_to_f16(v - f16_to_f32(hi));      // v - hi is exact in fp32
            std::memcpy(out.bytes.data() + i * 2, &q, 2);
        } else {
            const uint16_t q = (mode == NRF_MMA_BF16) ? f32_to_bf16(v) : f32_to_f16(v);
            std::memcpy(out.bytes.data() + i * 2, &q, 2);
        }
    }
    return out;
}

// ---------------------------------------------------------------------------
// training path
// ---------------------------------------------------------------------------
// Saved-tensor slots: train_impl.hpp (V1), train_v2_impl.hpp (V2).
bool make_backward_plan(const nrf_arch& a, const std::vector<HostLinear>& lin, NetPlan& plan, std::string& err) {
    plan = NetPlan();
    if (a.net != NRF_NET_V1 && a.net != NRF_NET_V2 && a.net != NRF_NET_V3) { err = "unknown network family"; return false; }
    if (a.hidden != 256 || (int)lin.size() != expected_linears(a)) { err = "backward plan: unexpected architecture"; return false; }
    const int n = a.n_layers, H = a.hidden, HT = H / 32;
    const int b = a.net == NRF_NET_V3 ? 5 : 0;                     // index of the first trunk Linear
    auto first_cols = [&](int count, int MT) { std::vector<int> r(32 * MT, -1); for (int i = 0; i < count; ++i) r[i] = i; return r; };
    auto transposed = [&](int li, int out_rows, int KT, int in_cols, int MT) {      // lin[li]^T: K = its rows, outputs = its first in_cols columns
        LayerPlan L; L.transposed = true; L.KT = KT; L.MT = MT; L.krow.assign(32 * KT, {-1, 0});
        for (int k = 0; k < out_rows; ++k) L.krow[k] = {li, k};
        L.rcol = first_cols(in_cols, MT);
        return L;
    };
    if (a.net == NRF_NET_V1) {
        LayerPlan L; L.transposed = true; L.KT = 1; L.MT = HT; L.krow.assign(32, {-1, 0});        // head^T: K = [rgb_out rows 0..2, sigma_out row 0]
        for (int c = 0; c < 3; ++c) L.krow[c] = {n + 1, c};
        L.krow[3] = {n, 0};
        L.rcol = first_cols(H, HT);
        plan.layers.push_back(std::move(L));
    } else {
        plan.layers.push_back(transposed(b + n + 4, 3, 1, H / 4, H / 128));         // color_layers.4^T: 3 -> 64
        plan.layers.push_back(transposed(b + n + 3, H / 4, H / 128, H / 2, H / 64));   // color_layers.2^T: 64 -> 128
        plan.layers.push_back(transposed(b + n + 2, H / 2, H / 64, H, HT));         // color_layers.0^T, feature columns only: 128 -> 256
        LayerPlan FD = transposed(b + n + 1, H, HT + 1, H, HT);                     // [feature_head | density_head]^T: 256 + 1 -> 256
        FD.krow[H] = {b + n, 0};
        plan.layers.push_back(std::move(FD));
    }
    for (int l = n - 1; l >= 1; --l) plan.layers.push_back(transposed(b + l, H, HT, H, HT));   // trunk layer l^T: dZ_l -> dH_{l-1}
    if (a.net == NRF_NET_V3) {
        // below the trunk (train_v3_impl.hpp): trunk layer 0^T, output_proj^T, fusion.2^T, fusion.0^T (second pass: dX is
        // needed for the gate), attention.2^T, attention.0^T, fusion.2^T again (first pass; its fusion.0 needs no dX)
        NetPlan fwd;
        if (!make_plan(a, lin, fwd, err)) return false;
        const LayerPlan& F0 = fwd.layers[0];                                       // K order of [PE(pos) | dino] in the kernel
        plan.layers.push_back(transposed(b, H, HT, H, HT));
        plan.layers.push_back(transposed(4, H, HT, H, HT));
        plan.layers.push_back(transposed(1, H, HT, H, HT));
        LayerPlan F0T = transposed(0, H, HT, 0, F0.KT);
        for (int r = 0; r < 32 * F0.KT; ++r) F0T.rcol[r] = F0.col[r];
        plan.layers.push_back(std::move(F0T));
        plan.layers.push_back(transposed(3, 2, 1, H / 4, H / 128));                  // attention.2^T: 2 -> 64
        plan.layers.push_back(transposed(2, H / 4, H / 128, H, HT));                 // attention.0^T: 64 -> 256
        plan.layers.push_back(transposed(1, H, HT, H, HT));
    }
    int off = 0;
    for (auto& L : plan.layers) { L.bias_off = off; off += 32 * L.MT; }
    plan.n_bias = off;
    return true;
}

bool make_train_plan(const nrf_arch& a, const NetPlan& fwd, const ParamLayout& lay, TrainPlan& tp, std::string& err) {
    tp = TrainPlan();
    if (a.net != NRF_NET_V1 && a.net != NRF_NET_V2 && a.net != NRF_NET_V3) { err = "unknown network family"; return false; }
    const int n = a.n_layers;
    const int expect = a.net == NRF_NET_V1 ? n + 1 : (a.net == NRF_NET_V2 ? n + 5 : n + 12);
    if ((int)fwd.layers.size() != expect) { err = "train plan: forward plan has an unexpected layer count"; return false; }
    // one job per forward-plan layer (a window of <= 8 input tiles of it)
    auto add_job = [&](const LayerPlan& L, int x_slot, int dz_slot, int x_first, int KT, bool with_bias) {
        GradJobPlan J;
        J.x_slot = x_slot; J.dz_slot = dz_slot; J.KT = KT; J.MT = L.MT; J.x_first = x_first;
        J.row_w.assign(32 * L.MT, -1); J.row_b.assign(32 * L.MT, -1);
        std::vector<std::pair<int, int>> seen;
        for (int r = 0; r < 32 * L.MT; ++r) {
            const auto& rs = L.row[r];
            if (rs.first < 0) continue;
            bool dup = false;                   // head tiles repeat their rows for the second lane half: count once
            for (const auto& q : seen) dup = dup || q == rs;
            if (dup) continue;
            seen.push_back(rs);
            J.row_w[r] = (int32_t)(lay.w_off[rs.first] + (int64_t)rs.second * lay.in_f[rs.first]);
            if (with_bias) J.row_b[r] = (int32_t)(lay.b_off[rs.first] + rs.second);
        }
        J.col.assign(L.col.begin() + 32 * x_first, L.col.begin() + 32 * (x_first + KT));
        tp.jobs.push_back(std::move(J));
    };
    if (a.net == NRF_NET_V1) {
        tp.slot_tiles.assign(2 * n + 2, 8);
        tp.slot_tiles[0] = fwd.layers[0].KT;
        tp.slot_tiles[2 * n + 1] = 1;
        tp.n_mask_slots = n;                                       // plane l: ReLU of layers.{l}
        for (int l = 0; l <= n; ++l) add_job(fwd.layers[l], l, l < n ? n + 1 + l : 2 * n + 1, 0, fwd.layers[l].KT, true);
        return true;
    }
    if (a.net == NRF_NET_V3) {
        // slots and planes: train_v3_impl.hpp.  Forward plan layers: 0 F0, 1 F1, 2 A0, 3 A2, 4 F0 (second pass), 5 F1, 6 proj,
        // 7.. trunk, then density_head, feature_head, colour layers.
        const int KT0 = fwd.layers[0].KT, D = 11 + n;
        tp.slot_tiles.assign(23 + 2 * n, 8);
        tp.slot_tiles[0] = KT0; tp.slot_tiles[3] = 2; tp.slot_tiles[4] = KT0;
        tp.slot_tiles[8 + n] = 9; tp.slot_tiles[9 + n] = 4; tp.slot_tiles[10 + n] = 2;
        tp.slot_tiles[D + 2] = 2; tp.slot_tiles[D + 3] = 1;
        tp.slot_tiles[D + 7 + n] = 1; tp.slot_tiles[D + 9 + n] = 4; tp.slot_tiles[D + 10 + n] = 2; tp.slot_tiles[D + 11 + n] = 1;
        tp.n_mask_slots = 7 + n;
        tp.aux_floats = 2;                                           // the gate (w0, w1) per sample
        add_job(fwd.layers[0], 0, D + 0, 0, KT0, true);              // fusion.0, first pass
        add_job(fwd.layers[4], 4, D + 4, 0, KT0, true);              // fusion.0, second pass (same weights: the sums add up)
        add_job(fwd.layers[1], 1, D + 1, 0, 8, true);                // fusion.2, first pass
        add_job(fwd.layers[5], 5, D + 5, 0, 8, true);                // fusion.2, second pass
        add_job(fwd.layers[2], 2, D + 2, 0, 8, true);                // attention.0
        add_job(fwd.layers[3], 3, D + 3, 0, 2, true);                // attention.2
        add_job(fwd.layers[6], 6, D + 6, 0, 8, true);                // output_proj
        for (int l = 0; l < n; ++l) add_job(fwd.layers[7 + l], 7 + l, D + 7 + l, 0, 8, true);
        add_job(fwd.layers[7 + n], 7 + n, D + 7 + n, 0, 8, true);    // density_head
        add_job(fwd.layers[8 + n], 7 + n, D + 8 + n, 0, 8, true);    // feature_head
        add_job(fwd.layers[9 + n], 8 + n, D + 9 + n, 0, 8, true);    // color_layers.0, feature columns
        add_job(fwd.layers[9 + n], 8 + n, D + 9 + n, 8, 1, false);   // color_layers.0, direction columns
        add_job(fwd.layers[10 + n], 9 + n, D + 10 + n, 0, 4, true);  // color_layers.2
        add_job(fwd.layers[11 + n], 10 + n, D + 11 + n, 0, 2, true); // color_layers.4
        return true;
    }
    tp.slot_tiles.assign(2 * n + 9, 8);
    tp.slot_tiles[0] = fwd.layers[0].KT;
    tp.slot_tiles[n + 1] = 9; tp.slot_tiles[n + 2] = 4; tp.slot_tiles[n + 3] = 2;
    tp.n_mask_slots = n + 2;                                       // trunk planes 0..n-1, colour layer 0 (n), colour layer 2 (n+1)
    tp.slot_tiles[2 * n + 4] = 1; tp.slot_tiles[2 * n + 6] = 4; tp.slot_tiles[2 * n + 7] = 2; tp.slot_tiles[2 * n + 8] = 1;
    for (int l = 0; l < n; ++l) add_job(fwd.layers[l], l, n + 4 + l, 0, fwd.layers[l].KT, true);
    add_job(fwd.layers[n], n, 2 * n + 4, 0, 8, true);              // density_head
    add_job(fwd.layers[n + 1], n, 2 * n + 5, 0, 8, true);          // feature_head
    add_job(fwd.layers[n + 2], n + 1, 2 * n + 6, 0, 8, true);      // color_layers.0, feature columns
    add_job(fwd.layers[n + 2], n + 1, 2 * n + 6, 8, 1, false);     // color_layers.0, direction-encoding columns (bias counted above)
    add_job(fwd.layers[n + 3], n + 2, 2 * n + 7, 0, 4, true);      // color_layers.2
    add_job(fwd.layers[n + 4], n + 3, 2 * n + 8, 0, 2, true);      // color_layers.4
    return true;
}

}  // namespace nrf

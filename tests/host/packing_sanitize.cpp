// Host-only harness for the weight packer (csrc/packing.cpp is plain C++): every plan / stream / source table of every network family
// and arithmetic mode, built under AddressSanitizer + UBSan by tests/test_packing_sanitize.py (sanitizers run on the CPU build only).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../nerf_few_shot_limitations_amd/csrc/packing.hpp"

using namespace nrf;

static std::vector<HostLinear> linears(const std::vector<std::pair<int, int>>& shapes) {
    std::vector<HostLinear> out;
    unsigned s = 12345u;
    for (auto& sh : shapes) {
        HostLinear l;
        l.out_f = sh.first; l.in_f = sh.second;
        l.w.resize((size_t)l.out_f * l.in_f); l.b.resize(l.out_f);
        for (auto& v : l.w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        for (auto& v : l.b) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        out.push_back(l);
    }
    return out;
}

static int run(const nrf_arch& a, const std::vector<std::pair<int, int>>& shapes, const char* tag) {
    std::string err;
    std::vector<HostLinear> lin = linears(shapes);
    if ((int)lin.size() != expected_linears(a)) { std::printf("%s: linear count %zu != %d\n", tag, lin.size(), expected_linears(a)); return 1; }
    NetPlan plan, bplan;
    if (!make_plan(a, lin, plan, err)) { std::printf("%s: make_plan: %s\n", tag, err.c_str()); return 1; }
    const ParamLayout lay = param_layout(lin);
    size_t bytes = 0;
    for (int mode = 0; mode < 4; ++mode) {
        const PackedStream ps = pack_stream(plan, lin, mode);
        bytes += ps.bytes.size();
        const std::vector<int32_t> src = stream_sources(plan, lay, stream_kind(mode));
        for (int32_t v : src) if (v >= lay.total || v < -1) { std::printf("%s: source out of range\n", tag); return 1; }
    }
    const std::vector<float> bias = pack_bias(plan, lin);
    const std::vector<int32_t> bsrc = bias_sources(plan, lay);
    if (bias.size() != bsrc.size()) { std::printf("%s: bias table sizes differ\n", tag); return 1; }
    if (!make_backward_plan(a, lin, bplan, err)) { std::printf("%s: make_backward_plan: %s\n", tag, err.c_str()); return 1; }
    for (int mode = 0; mode < 3; ++mode) bytes += pack_stream(bplan, lin, mode).bytes.size();
    for (int kind = 0; kind < 2; ++kind) (void)stream_sources(bplan, lay, kind);
    TrainPlan tp;
    if (!make_train_plan(a, plan, lay, tp, err)) { std::printf("%s: make_train_plan: %s\n", tag, err.c_str()); return 1; }
    std::printf("%s ok: %zu layers, %d bias floats, %zu jobs, %zu stream bytes\n", tag, plan.layers.size(), plan.n_bias, tp.jobs.size(), bytes);
    return 0;
}

int main() {
    int rc = 0;
    for (int n = 1; n <= 8; n += (n == 1 ? 1 : 3)) {                       // trunk depths 1, 2, 5, 8
        {   // V1: layers.0..n-1, sigma_out, rgb_out
            nrf_arch a{NRF_NET_V1, 10, 0, 256, n, 0};
            std::vector<std::pair<int, int>> sh;
            for (int i = 0; i < n; ++i) sh.push_back({256, i == 0 ? 63 : 256});
            sh.push_back({1, 256}); sh.push_back({3, 256});
            rc |= run(a, sh, "v1");
        }
        {   // V2
            nrf_arch a{NRF_NET_V2, 10, 4, 256, n, 0};
            std::vector<std::pair<int, int>> sh;
            for (int i = 0; i < n; ++i) sh.push_back({256, i == 0 ? 63 : 256});
            sh.push_back({1, 256}); sh.push_back({256, 256});
            sh.push_back({128, 256 + 27}); sh.push_back({64, 128}); sh.push_back({3, 64});
            rc |= run(a, sh, "v2");
        }
        for (int dd = 64; dd <= 128; dd += 64) {   // V3, dino_dim 64 / 128
            nrf_arch a{NRF_NET_V3, 12, 4, 256, n, dd};
            std::vector<std::pair<int, int>> sh = {{256, 75 + dd}, {256, 256}, {64, 256}, {2, 64}, {256, 256}};
            for (int i = 0; i < n; ++i) sh.push_back({256, 256});
            sh.push_back({1, 256}); sh.push_back({256, 256});
            sh.push_back({128, 256 + 27}); sh.push_back({64, 128}); sh.push_back({3, 64});
            rc |= run(a, sh, dd == 64 ? "v3" : "v3w");
        }
    }
    // malformed architectures must be refused, not crash
    std::string err;
    NetPlan plan;
    nrf_arch bad{NRF_NET_V2, 10, 4, 256, 8, 0};
    std::vector<HostLinear> few = linears({{256, 63}, {1, 256}});
    if (make_plan(bad, few, plan, err)) { std::printf("malformed arch accepted\n"); rc = 1; }
    nrf_arch bad2{7, 10, 4, 256, 8, 0};
    if (make_plan(bad2, few, plan, err)) { std::printf("unknown family accepted\n"); rc = 1; }
    std::printf(rc ? "FAILED\n" : "sanitize ok\n");
    return rc;
}

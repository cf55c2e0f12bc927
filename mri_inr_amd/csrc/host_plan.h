// Host-side plans of round 5 (plain C++, no HIP: unit-tested on the CPU by tests/test_host_plan.py):
//   * how a synchronous host-pointer call cuts itself into chunks over the handle's two streams (msiren_forward_tiles_impl);
//   * the layout of the one-launch prologue's packed weight stream (pack_prologue_f16x3 / latent_mods_f16x3_kernel): where each
//     section starts, how many k-steps it has, which ring slot a k-step lands in.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace msiren {

struct HostChunk {
    int64_t lo, n;   // tiles [lo, lo + n) of the call
    int stream;      // which of the handle's two streams
    int trunk;       // 0 = the launcher's own rule, 1 = register-resident (room beside it), 2 = weight-stationary
    bool beside;     // its prologue runs beside the previous chunk's trunk (shallow weight ring)
};

// Chunk 0 is small (its upload is short: the device starts early; its trunk runs while the next chunk's tiles arrive), the
// others are `piece` tiles, the last one absorbs a remainder below 128 tiles.  Every chunk but the last takes the
// register-resident trunk (the next chunk's prologue runs beside it), the last one the weight-stationary trunk.
inline std::vector<HostChunk> pipelined_host_plan(int64_t B, int64_t first, int64_t piece, int first_stream) {
    std::vector<HostChunk> plan;
    if (B <= 0) return plan;
    first = std::max<int64_t>(1, std::min<int64_t>(first, std::max<int64_t>(16, B / 3)));
    first = std::min<int64_t>(first, B);
    piece = std::max<int64_t>(1, piece);
    plan.push_back({0, first, first_stream & 1, first == B ? 2 : 1, false});
    for (int64_t lo = first; lo < B;) {
        int64_t n = std::min<int64_t>(piece, B - lo);
        if (B - lo - n < 128) n = B - lo;  // (no tiny last chunk)
        const bool last = lo + n == B;
        plan.push_back({lo, n, (int)((plan.size() & 1) ^ (first_stream & 1)), last ? 2 : 1, true});
        lo += n;
    }
    return plan;
}

// k-steps of one wave's stream, by section (H = 128 NPH, Z = 128 NPZ, L layers; with / without the encoder's and the Modulator's weights)
struct EmStreamLayout {
    int c3, fc, zp, hl;        // k-steps of conv3 (this wave's K half), Linear(64, Z), the latent stage, the hidden chain
    int zp_start, hl_start, total;
    int z_pass(int l, int ph, int nph, int kz) const { return zp_start + (l * nph + ph) * kz; }
    int h_pass(int l, int ph, int nph, int kh) const { return hl_start + ((l - 1) * nph + ph) * kh; }
};
inline EmStreamLayout em_stream_layout(int nph, int npz, int L, bool enc, bool mod, int c3_ksteps_half = 32, int fc_ksteps = 4) {
    EmStreamLayout s{};
    const int kz = 4 * npz, kh = 4 * nph;
    s.c3 = enc ? c3_ksteps_half : 0;
    s.fc = enc ? npz * fc_ksteps : 0;
    s.zp = mod ? L * nph * kz : 0;
    s.hl = mod && L > 1 ? (L - 1) * nph * kh : 0;
    s.zp_start = s.c3 + s.fc;
    s.hl_start = s.zp_start + s.zp;
    s.total = s.hl_start + s.hl;
    return s;
}
// Ring slots are compile-time constants in the kernel: a stage whose repeat count is a run-time value must advance the stream by a
// multiple of the ring depth per repetition, and both entry points (stream start, latent stage) must agree on the slot of every k-step.
constexpr bool em_ring_depth_ok(int nph, int npz, int depth, int c3_ksteps_half = 32) {
    const int kz = 4 * npz, kh = 4 * nph;
    return depth > 0 && c3_ksteps_half % depth == 0 && (nph * kz) % depth == 0 && (nph * kh) % depth == 0;
}

}  // namespace msiren

// libmsiren.so, host side: the reference's state_dict (host copies in msiren_ctx::tensors) -> the kernels' weight layouts in HBM.
// Host arithmetic (fp64 scaling, hi / lo splits, fragment orders) + blocking uploads; called from msiren_commit_weights.  The layouts
// are documented at the kernels that consume them; DESIGN.md section 3 has the table.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "host_ctx.h"
#include "host_plan.h"
#include "encoder_modulator_f16x3.hip.h"   // (templates only: EM_* constants, em_u4)
#include "siren_trunk_f16x3n.hip.h"        // F16Lds
#include "siren_trunk_x1n.hip.h"           // X1nLds
#include "siren_trunk_x1w.hip.h"           // X1wLds
#include "trunk_instances.h"

namespace mh {

void declare_expected(msiren_ctx* h) {
    auto& e = h->expected;
    const size_t H = h->H, Z = h->Z, L = h->L;
    e["grid"] = (size_t)h->P * 2;
    for (size_t l = 0; l < L; ++l) {
        const std::string p = "net.layers." + std::to_string(l);
        e[p + ".weight"] = H * (l == 0 ? 2 : H);
        if (h->cfg.use_bias) e[p + ".bias"] = H;
        const std::string m = "modulator.layers." + std::to_string(l) + ".0";
        e[m + ".weight"] = H * (l == 0 ? Z : H + Z);
        e[m + ".bias"] = H;
    }
    e["net.last_layer.weight"] = H;
    if (h->cfg.use_bias) e["net.last_layer.bias"] = 1;
    const std::string en = "encoder.encoder.encoder.";
    e[en + "0.weight"] = 16 * 1 * 3 * 3;
    e[en + "0.bias"] = 16;
    e[en + "2.weight"] = 32 * 16 * 3 * 3;
    e[en + "2.bias"] = 32;
    e[en + "4.weight"] = 64 * 32 * 8 * 8;
    e[en + "4.bias"] = 64;
    e[en + "7.weight"] = Z * 64;
    e[en + "7.bias"] = Z;
}

const std::vector<float>* get(msiren_ctx* h, const std::string& k) {
    auto it = h->tensors.find(k);
    return it == h->tensors.end() ? nullptr : &it->second;
}

// ---- trunk packing --------------------------------------------------------------------------
// Everything is scaled by w0/(2*pi) in double before rounding to fp32, so that the kernel's
// accumulator is the sine argument in revolutions (see siren_trunk_f32.hip.h).
int pack_trunk(msiren_ctx* h) {
    const int H = h->H, HP = h->HP, L = h->L;
    const int TT = HP / 128, QN = HP / 8;
    const double two_pi = 6.283185307179586476925286766559;
    const double c0 = (double)h->cfg.w0_initial / two_pi, c = (double)h->cfg.w0 / two_pi;
    std::string missing;
    auto need = [&](const std::string& k) -> const std::vector<float>* {
        const auto* v = get(h, k);
        if (!v) missing += (missing.empty() ? "" : ", ") + k;
        return v;
    };
    std::vector<const std::vector<float>*> W(L), Bv(L);
    for (int l = 0; l < L; ++l) {
        W[l] = need("net.layers." + std::to_string(l) + ".weight");
        Bv[l] = h->cfg.use_bias ? need("net.layers." + std::to_string(l) + ".bias") : nullptr;
    }
    const auto* Wo = need("net.last_layer.weight");
    const auto* Bo = h->cfg.use_bias ? need("net.last_layer.bias") : nullptr;
    if (!missing.empty())
        return fail(MSIREN_E_STATE, "Missing key(s) in state_dict: %s", missing.c_str());

    std::vector<float>& grid = h->grid_host;  // the layer-0 tables of the 16-bit trunks are built from it as well
    if (const auto* g = get(h, "grid")) {
        grid = *g;
    } else {  // the reference registers it as a buffer (modulated_siren.py:427-433); rebuild it if a checkpoint lacks it
        grid.resize((size_t)h->P * 2);
        const int S = h->S;
        std::vector<float> lin(S);
        const float step = S > 1 ? (1.0f - (-1.0f)) / (float)(S - 1) : 0.f;
        for (int i = 0; i < S; ++i) lin[i] = (i < S / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(S - 1 - i));
        for (int a = 0; a < S; ++a)
            for (int b2 = 0; b2 < S; ++b2) {
                grid[((size_t)a * S + b2) * 2 + 0] = lin[a];
                grid[((size_t)a * S + b2) * 2 + 1] = lin[b2];
            }
    }

    std::vector<float> l0((size_t)HP * 4, 0.f);
    for (int f = 0; f < H; ++f) {
        l0[(size_t)f * 4 + 0] = (float)((double)(*W[0])[(size_t)f * 2 + 0] * c0);
        l0[(size_t)f * 4 + 1] = (float)((double)(*W[0])[(size_t)f * 2 + 1] * c0);
        l0[(size_t)f * 4 + 2] = Bv[0] ? (float)((double)(*Bv[0])[f] * c0) : 0.f;
    }
    const int nh = L > 1 ? L - 1 : 0;
    std::vector<float> wp((size_t)std::max(nh, 1) * 4 * QN * TT * 256, 0.f);
    std::vector<float> bias((size_t)std::max(nh, 1) * HP, 0.f);
    for (int l = 1; l < L; ++l) {
        const std::vector<float>& w = *W[l];
        for (int wave = 0; wave < 4; ++wave)
            for (int q = 0; q < QN; ++q)
                for (int tt = 0; tt < TT; ++tt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int f = wave * 32 * TT + 32 * tt + (lane & 31);
                        float* dst = &wp[(((((size_t)(l - 1) * 4 + wave) * QN + q) * TT + tt) * 64 + lane) * 4];
                        for (int j = 0; j < 4; ++j) {
                            const int k = 8 * q + 4 * (lane >> 5) + j;
                            dst[j] = (f < H && k < H) ? (float)((double)w[(size_t)f * H + k] * c) : 0.f;
                        }
                    }
        if (Bv[l])
            for (int f = 0; f < H; ++f) bias[(size_t)(l - 1) * HP + f] = (float)((double)(*Bv[l])[f] * c);
    }
    std::vector<float> wout(HP, 0.f);
    for (int f = 0; f < H; ++f) wout[f] = (float)((double)(*Wo)[f] * c);
    h->bout = Bo ? (float)((double)(*Bo)[0] * c) : 0.f;
    // Morlet: exp(-0.5 p^2) with p = r * 2pi / w  ->  exp2(cg * r^2)
    const double log2e = 1.4426950408889634;
    h->cg0 = (float)(-0.5 * log2e * (two_pi / h->cfg.w0_initial) * (two_pi / h->cfg.w0_initial));
    h->cg = (float)(-0.5 * log2e * (two_pi / h->cfg.w0) * (two_pi / h->cfg.w0));

    int rc;
    if ((rc = upload(&h->d_grid, grid))) return rc;
    if ((rc = upload(&h->d_l0, l0))) return rc;
    if ((rc = upload(&h->d_wp, wp))) return rc;
    if ((rc = upload(&h->d_bias, bias))) return rc;
    if ((rc = upload(&h->d_wout, wout))) return rc;
    return 0;
}

// ---- split-fp16 trunk packing ------------------------------------------------------------------
// Chunk (layer l, feature tile t) = [16 k-steps][hi|lo][64 lanes][8 x f16]; lane (r = lane&31, h = lane>>5),
// element j of k-step s multiplies feature  kf = 32*(s>>1) + 16*(s&1) + 8*(j>>2) + 4*h + (j&3)  of the
// previous layer -- the order in which the previous layer's accumulator registers hold them.
// Weights are scaled by w0/2pi and by 2^e (e per layer, max|W| -> [8192, 16384)) before the split.
uint16_t f32_to_f16_rne(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
float f16_to_f32(uint16_t u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
}

int pack_trunk_f16x3(msiren_ctx* h) {
    h->f16x3_ready = false;
    const int H = h->H, L = h->L;
    if (h->cfg.precision != MSIREN_PREC_F16X3 || H != 256 || L < 2 || msiren::F16Lds<3>::total(L) > 160 * 1024) return 0;
    const double two_pi = 6.283185307179586476925286766559;
    const double c = (double)h->cfg.w0 / two_pi;
    std::vector<uint16_t> wpn((size_t)(L - 1) * 8 * 16 * 2 * 64 * 8);
    std::vector<float> bias((size_t)(L - 1) * 256, 0.f), wout(256, 0.f);
    for (int l = 1; l < L; ++l) {
        const std::vector<float>& w = *get(h, "net.layers." + std::to_string(l) + ".weight");
        double mx = 0.0;
        for (float v : w) mx = std::max(mx, std::fabs((double)v * c));
        // 16x16x32 kernel (siren_trunk_f16x3n.hip.h): chunk (l, t) = [8 k-steps][2 sub-tiles][hi|lo][64 lanes][8 x f16];
        // lane (r = lane & 15, q = lane >> 4), element j of k-step s of sub-tile u: output feature 32 t + 16 u + r,
        // input feature 32 s + 16 (j >> 2) + 4 q + (j & 3).  Scale 2^a with rms|W'| ~ 0.1 (a is undone on the
        // activation side, through the previous layer's modulation row, so the accumulator is the sine argument).
        {
            double sq = 0.0;
            for (float v : w) sq += ((double)v * c) * ((double)v * c);
            const double rmsw = std::sqrt(sq / (double)w.size());
            int a = 0;
            if (rmsw > 0.0) a = (int)std::lround(std::log2(0.1 / rmsw));
            if (mx > 0.0) a = std::min(a, (int)std::floor(std::log2(32768.0 / mx)));  // stay inside fp16
            a = std::max(-14, std::min(a, 30));
            const double scn = std::ldexp(c, a);
            h->mscale16[l - 1] = (float)std::ldexp(1.0, -a);  // row l-1 of the modulation table
            for (int t = 0; t < 8; ++t)
                for (int s2 = 0; s2 < 8; ++s2)
                    for (int u = 0; u < 2; ++u)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int f = 32 * t + 16 * u + (lane & 15);
                                const int k = 32 * s2 + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                                const float ws = (float)((double)w[(size_t)f * H + k] * scn);
                                const uint16_t hi = f32_to_f16_rne(ws);
                                const uint16_t lo = f32_to_f16_rne(ws - f16_to_f32(hi));
                                const size_t base = (((((size_t)(l - 1) * 8 + t) * 8 + s2) * 2 + u) * 2) * 64 * 8;
                                wpn[base + (size_t)lane * 8 + j] = hi;
                                wpn[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
                            }
        }
        if (const auto* b = h->cfg.use_bias ? get(h, "net.layers." + std::to_string(l) + ".bias") : nullptr)
            for (int f = 0; f < H; ++f) bias[(size_t)(l - 1) * 256 + f] = (float)((double)(*b)[f] * c);
    }
    h->mscale16[L - 1] = 1.0f;  // the last hidden layer's output meets last_layer unscaled
    const auto* Wo = get(h, "net.last_layer.weight");
    for (int f = 0; f < H; ++f) wout[f] = (float)((double)(*Wo)[f] * c);
    if (h->d_wp16n) HIPCHK(hipFree(h->d_wp16n));
    h->d_wp16n = nullptr;
    HIPCHK(hipMalloc(&h->d_wp16n, wpn.size() * 2));
    HIPCHK(hipMemcpy(h->d_wp16n, wpn.data(), wpn.size() * 2, hipMemcpyHostToDevice));
    int rc;
    if ((rc = upload(&h->d_bias16, bias))) return rc;
    if ((rc = upload(&h->d_wout16, wout))) return rc;
    {   // layer-0 activation table S0T[f/4][p][f%4] = act0(w0_initial * (W0 x_p + b0)), fp64 -> fp32
        const auto& W0 = *get(h, "net.layers.0.weight");
        const auto* B0 = h->cfg.use_bias ? get(h, "net.layers.0.bias") : nullptr;
        const std::vector<float>& grid = h->grid_host;  // pack_trunk ran first
        if (grid.size() != (size_t)h->P * 2) return fail(MSIREN_E_STATE, "grid buffer missing");
        std::vector<float> tab((size_t)64 * h->P * 4);
        const bool morlet = h->cfg.activation == MSIREN_ACT_MORLET;
        for (int f = 0; f < 256; ++f)
            for (int pidx = 0; pidx < h->P; ++pidx) {
                // the pre-activation is formed in fp32 like F.linear does, the activation in fp64
                const float pre = std::fmaf(grid[(size_t)pidx * 2 + 1], W0[(size_t)f * 2 + 1],
                                            std::fmaf(grid[(size_t)pidx * 2], W0[(size_t)f * 2], B0 ? (*B0)[f] : 0.f));
                double a = std::sin((double)h->cfg.w0_initial * (double)pre);
                if (morlet) a *= std::exp(-0.5 * (double)pre * (double)pre);
                tab[((size_t)(f / 4) * h->P + pidx) * 4 + (f & 3)] = (float)a;
            }
        if ((rc = upload(&h->d_s0t, tab))) return rc;
    }
    h->f16x3_ready = true;
    return 0;
}

// ---- single-product 16-bit trunk packing (H = 512) ---------------------------------------------
uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int pack_trunk_x1(msiren_ctx* h) {
    h->x1_ready = false;
    const int H = h->H, L = h->L;
    const bool bf = h->cfg.precision == MSIREN_PREC_BF16;
    if (!(h->cfg.precision == MSIREN_PREC_BF16 || h->cfg.precision == MSIREN_PREC_F16)) return 0;
    // (the kernel launch_trunk_x1_kernel will pick: weight-stationary from 3 layers on, depths 2..11; register-resident 2..10)
    const int lds_need = L >= 3 ? msiren::X1wLds::total(L) : msiren::X1nLds<3>::total(L);
    if (H != 512 || L < 2 || L > 65 || lds_need > 160 * 1024)
        return fail(MSIREN_E_INVALID, "precision bf16/f16 (single-product trunk) needs dim_hidden = 512 and 2 <= num_layers <= 11: its tables must fit the 160 KB LDS; got H=%d L=%d", H, L);
    const double two_pi = 6.283185307179586476925286766559;
    const double c = (double)h->cfg.w0 / two_pi;
    std::vector<uint16_t> wpn((size_t)(L - 1) * 16 * 16 * 2 * 64 * 8), wout(512, 0);  // chunk (l, t) = [16 k-steps][2 sub-tiles][64 lanes][8]
    std::vector<float> bias32((size_t)(L - 1) * 512, 0.f);
    std::vector<uint16_t> wpw(wpn.size());
    for (int l = 1; l < L; ++l) {
        const std::vector<float>& w = *get(h, "net.layers." + std::to_string(l) + ".weight");
        int e = 0;
        if (!bf) {  // fp16: scale max|W| into [8192, 16384); bf16 has fp32's exponent range
            double mx = 0.0;
            for (float v : w) mx = std::max(mx, std::fabs((double)v * c));
            if (mx > 0.0) e = std::max(-14, std::min((int)std::floor(std::log2(16384.0 / mx)), 30));
        }
        const double sc = std::ldexp(c, e);
        h->winvx1[l - 1] = (float)std::ldexp(1.0, -e);
        // lane (r = lane & 15, q = lane >> 4), element j of k-step s of sub-tile u: output feature 32 t + 16 u + r, input
        // feature 32 s + 16 (j >> 2) + 4 q + (j & 3) (the order of siren_trunk_f16x3n.hip.h)
        for (int t = 0; t < 16; ++t)
            for (int s = 0; s < 16; ++s)
                for (int u = 0; u < 2; ++u)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int f = 32 * t + 16 * u + (lane & 15);
                            const int k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                            const float ws = (float)((double)w[(size_t)f * H + k] * sc);
                            wpn[(((((size_t)(l - 1) * 16 + t) * 16 + s) * 2 + u) * 64 + lane) * 8 + j] = bf ? f32_to_bf16_rne(ws) : f32_to_f16_rne(ws);
                        }
        // weight-stationary kernel: block ((l - 1) * 2 + n) * 4 + wave = [16 k-steps][4 tiles][64 lanes][8]: output feature
        // 256 n + 64 wave + 16 t + r, input feature as above
        for (int n = 0; n < 2; ++n)
            for (int wv = 0; wv < 4; ++wv)
                for (int s = 0; s < 16; ++s)
                    for (int t = 0; t < 4; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int f = 256 * n + 64 * wv + 16 * t + (lane & 15);
                                const int k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
                                const float ws = (float)((double)w[(size_t)f * H + k] * sc);
                                wpw[((((((size_t)(l - 1) * 2 + n) * 4 + wv) * 16 + s) * 4 + t) * 64 + lane) * 8 + j] = bf ? f32_to_bf16_rne(ws) : f32_to_f16_rne(ws);
                            }
        if (const auto* b = h->cfg.use_bias ? get(h, "net.layers." + std::to_string(l) + ".bias") : nullptr)
            for (int f = 0; f < H; ++f) bias32[(size_t)(l - 1) * 512 + f] = (float)((double)(*b)[f] * sc);  // (x 2^e: the accumulator is scaled like the weights)
    }
    const auto* Wo = get(h, "net.last_layer.weight");
    for (int f = 0; f < H; ++f) wout[f] = f32_to_f16_rne((float)((double)(*Wo)[f] * c));
    const auto& W0 = *get(h, "net.layers.0.weight");
    const auto* B0 = h->cfg.use_bias ? get(h, "net.layers.0.bias") : nullptr;
    const std::vector<float>* g = &h->grid_host;  // pack_trunk ran first
    if (g->size() != (size_t)h->P * 2) return fail(MSIREN_E_STATE, "grid buffer missing");
    std::vector<float> tab((size_t)128 * h->P * 4);  // layer-0 activation table S0T[f/4][p][f%4] = act0(w0_initial * (W0 x_p + b0))
    const bool morlet = h->cfg.activation == MSIREN_ACT_MORLET;
    for (int f = 0; f < 512; ++f)
        for (int pidx = 0; pidx < h->P; ++pidx) {
            const float pre = std::fmaf((*g)[(size_t)pidx * 2 + 1], W0[(size_t)f * 2 + 1],
                                        std::fmaf((*g)[(size_t)pidx * 2], W0[(size_t)f * 2], B0 ? (*B0)[f] : 0.f));
            double a = std::sin((double)h->cfg.w0_initial * (double)pre);
            if (morlet) a *= std::exp(-0.5 * (double)pre * (double)pre);
            tab[((size_t)(f / 4) * h->P + pidx) * 4 + (f & 3)] = (float)a;
        }
    auto up16 = [&](void** dst, const std::vector<uint16_t>& v) -> int {
        if (*dst) HIPCHK(hipFree(*dst));
        *dst = nullptr;
        HIPCHK(hipMalloc(dst, v.size() * 2));
        HIPCHK(hipMemcpy(*dst, v.data(), v.size() * 2, hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up16(&h->d_wpx1n, wpn)) || (rc = up16(&h->d_wpx1w, wpw)) || (rc = up16(&h->d_woutx1, wout))) return rc;
    if ((rc = upload(&h->d_bias32x1, bias32)) || (rc = upload(&h->d_s0t512, tab))) return rc;
    h->x1_ready = true;
    return 0;
}

// ---- modulator / encoder packing ------------------------------------------------------------
int pack_modulator(msiren_ctx* h) {
    const int H = h->H, Z = h->Z, L = h->L;
    // transposed: Wt[l][k][f], k over [hidden(H) ; latent(Z)] (layer 0: latent only), so that a
    // thread per output feature reads consecutive addresses
    size_t total = 0;
    for (int l = 0; l < L; ++l) total += (size_t)(l == 0 ? Z : H + Z) * H;
    std::vector<float> wt(total), bb((size_t)L * H);
    size_t off = 0;
    for (int l = 0; l < L; ++l) {
        const auto* w = get(h, "modulator.layers." + std::to_string(l) + ".0.weight");
        const auto* b = get(h, "modulator.layers." + std::to_string(l) + ".0.bias");
        if (!w || !b) return 1;  // not present: latent/tiles entry points stay unavailable
        const int K = (l == 0 ? Z : H + Z);
        for (int f = 0; f < H; ++f)
            for (int k = 0; k < K; ++k) wt[off + (size_t)k * H + f] = (*w)[(size_t)f * K + k];
        for (int f = 0; f < H; ++f) bb[(size_t)l * H + f] = (*b)[f];
        off += (size_t)K * H;
    }
    std::vector<float> rm(total);
    off = 0;
    for (int l = 0; l < L; ++l) {
        const auto* w = get(h, "modulator.layers." + std::to_string(l) + ".0.weight");
        std::copy(w->begin(), w->end(), rm.begin() + off);
        off += w->size();
    }
    int rc;
    if ((rc = upload(&h->d_modw, wt))) return rc;
    if ((rc = upload(&h->d_modw_rm, rm))) return rc;
    if ((rc = upload(&h->d_modb, bb))) return rc;
    return 0;
}

int pack_encoder(msiren_ctx* h) {
    const std::string en = "encoder.encoder.encoder.";
    const char* keys[8] = {"0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias", "7.weight", "7.bias"};
    const std::vector<float>* t[8];
    for (int i = 0; i < 8; ++i) {
        t[i] = get(h, en + keys[i]);
        if (!t[i]) return 1;
    }
    const int Z = h->Z;
    // one blob: [c1w 16x9][c1b 16][c2w (144,32) transposed][c2b 32][c3w (2048,64) transposed][c3b 64]
    //           [fcw (64,Z) transposed][fcb Z]
    std::vector<float> blob;
    auto push = [&](const std::vector<float>& v) {
        size_t o = blob.size();
        blob.insert(blob.end(), v.begin(), v.end());
        while (blob.size() % 4) blob.push_back(0.f);
        return o;
    };
    msiren::EncoderParams ep{};
    size_t o_c1w = push(*t[0]);
    size_t o_c1b = push(*t[1]);
    std::vector<float> c2t((size_t)144 * 32);
    for (int o = 0; o < 32; ++o)
        for (int k = 0; k < 144; ++k) c2t[(size_t)k * 32 + o] = (*t[2])[(size_t)o * 144 + k];
    size_t o_c2w = push(c2t);
    size_t o_c2b = push(*t[3]);
    std::vector<float> c3t((size_t)2048 * 64);
    for (int o = 0; o < 64; ++o)
        for (int k = 0; k < 2048; ++k) c3t[(size_t)k * 64 + o] = (*t[4])[(size_t)o * 2048 + k];
    size_t o_c3w = push(c3t);
    size_t o_c3b = push(*t[5]);
    std::vector<float> fct((size_t)64 * Z);
    for (int o = 0; o < Z; ++o)
        for (int k = 0; k < 64; ++k) fct[(size_t)k * Z + o] = (*t[6])[(size_t)o * 64 + k];
    size_t o_fcw = push(fct);
    size_t o_fcb = push(*t[7]);
    size_t o_c3rm = push(*t[4]);  // (64, 2048) and (Z, 64) as stored: operands of the batched MFMA GEMMs
    size_t o_fcrm = push(*t[6]);
    int rc;
    if ((rc = upload(&h->d_encw, blob))) return rc;
    ep.c1w = h->d_encw + o_c1w;
    ep.c1b = h->d_encw + o_c1b;
    ep.c2w = h->d_encw + o_c2w;
    ep.c2b = h->d_encw + o_c2b;
    ep.c3w = h->d_encw + o_c3w;
    ep.c3b = h->d_encw + o_c3b;
    ep.fcw = h->d_encw + o_fcw;
    ep.fcb = h->d_encw + o_fcb;
    ep.Z = Z;
    h->enc = ep;
    h->d_c3w_rm = h->d_encw + o_c3rm;
    h->d_fcw_rm = h->d_encw + o_fcrm;
    return 0;
}

// ---- encoder tail + Modulator, split-fp16 (encoder_modulator_f16x3.hip.h) ------------------------------------------------
// Per wave one stream of k-steps in the order the kernel consumes them, each [tile 0 hi | tile 0 lo | tile 1 hi | tile 1 lo]
// x [64 lanes][8 x f16]; lane (m = lane & 15, q = lane >> 4), element j: output feature 16 T + m, input
// k(s, q, j) = 32 s + 16 (j >> 2) + 4 q + (j & 3) of k-step s.  Sections: conv3 (32 k-steps: the wave's K half of its tile
// pair), Linear(64, Z) (NPZ passes x 4 k-steps, the upper two zero), the latent part of every Modulator layer (L NPH passes
// x Z / 32), the hidden part of layers 1.. ((L - 1) NPH passes x H / 32).  Each layer is scaled by the power of two that
// brings max|W| into [2^13, 2^14) before the hi / lo split.
int pack_prologue_f16x3(msiren_ctx* h) {
    h->em_enc = h->em_mod = false;
    const int H = h->H, Z = h->Z, L = h->L;
    const bool enc = h->have_encoder && h->O == 32, mod = h->have_modulator;  // (a trunk + Modulator checkpoint has no encoder.* keys)
    if (!h->em_enabled || h->cfg.precision == MSIREN_PREC_F32 || (!enc && !mod)) return 0;
    if (!((H == 256 && Z == 256) || (H == 512 && Z == 128)) || L > 64) return 0;  // the instantiated (NPH, NPZ) pairs
    const int NPH = H / 128, NPZ = Z / 128, KH = H / 32, KZ = Z / 32;
    auto scale_of = [](const float* w, size_t n0, size_t stride, size_t rows, size_t cols) {  // exponent a: max|w| 2^a in [2^13, 2^14)
        double mx = 0.0;
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) mx = std::max(mx, std::fabs((double)w[n0 + r * stride + c]));
        if (!(mx > 0.0) || !std::isfinite(mx)) return 0;
        int e = 0;
        (void)std::frexp(mx, &e);  // mx = f 2^e, f in [0.5, 1)
        return std::max(-100, std::min(100, 14 - e));
    };
    const msiren::EmStreamLayout lay = msiren::em_stream_layout(NPH, NPZ, L, enc, mod, msiren::EM_C3_KSTEPS / 2, msiren::EM_FC_KSTEPS);  // (host_plan.h)
    const int zp_start = lay.zp_start, nk = lay.total;
    std::vector<uint16_t> ws(((size_t)4 * nk + msiren::EM_MAX_DEPTH) * 4 * 64 * 8, 0);  // (+ padding: the ring prefetches past the end)
    auto put = [&](int wave, int g, int t, int lane, int j, double v) {  // v already scaled
        const float f = (float)v;
        const uint16_t hi = f32_to_f16_rne(f), lo = f32_to_f16_rne(f - f16_to_f32(hi));
        const size_t base = (((size_t)wave * nk + g) * 4 + 2 * t) * 64 * 8 + (size_t)lane * 8 + j;
        ws[base] = hi;
        ws[base + 64 * 8] = lo;
    };
    auto kin = [](int s, int q, int j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); };
    std::vector<float> bias((size_t)64 + Z + (size_t)L * H, 0.f);
    if (enc) {
    const std::string en = "encoder.encoder.encoder.";
    const std::vector<float>&W3 = *get(h, en + "4.weight"), &B3 = *get(h, en + "4.bias"), &Wf = *get(h, en + "7.weight"), &Bf = *get(h, en + "7.bias");
    // conv2 as MFMA A fragments: lane (m, q), element j of k-step s: channel 16 mt + m, tap 2 s + (q >> 1), input channel 8 (q & 1) + j
    {
        const std::vector<float>& W2 = *get(h, en + "2.weight");  // (32, 16, 3, 3)
        const int a2 = scale_of(W2.data(), 0, 144, 32, 144);
        std::vector<uint16_t> c2((size_t)2 * 5 * 2 * 64 * 8, 0);
        for (int mt = 0; mt < 2; ++mt)
            for (int ks = 0; ks < 5; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int tap = 2 * ks + (lane >> 5), ci = 8 * ((lane >> 4) & 1) + j, o = 16 * mt + (lane & 15);
                        const float f = tap < 9 ? (float)std::ldexp((double)W2[(size_t)o * 144 + ci * 9 + tap], a2) : 0.f;
                        const uint16_t hi = f32_to_f16_rne(f), lo = f32_to_f16_rne(f - f16_to_f32(hi));
                        const size_t base = ((size_t)(mt * 5 + ks) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                        c2[base] = hi;
                        c2[base + 64 * 8] = lo;
                    }
        if (h->d_emc2) HIPCHK(hipFree(h->d_emc2));
        h->d_emc2 = nullptr;
        HIPCHK(hipMalloc(&h->d_emc2, c2.size() * 2));
        HIPCHK(hipMemcpy(h->d_emc2, c2.data(), c2.size() * 2, hipMemcpyHostToDevice));
        h->enc.c2f16 = h->d_emc2;
        h->enc.c2_winv = (float)std::ldexp(1.0, -a2);
    }
    // conv3: its k order is the order in which the conv kernel's threads hold the features (encoder_conv_f16x3_kernel<VARIANT>)
    const int a3 = scale_of(W3.data(), 0, 2048, 64, 2048);
    h->em_winv_c3 = (float)std::ldexp(1.0, -a3);
    for (int wave = 0; wave < 4; ++wave)
        for (int ks = 0; ks < msiren::EM_C3_KSTEPS / 2; ++ks)
            for (int t = 0; t < 2; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int s2 = (msiren::EM_C3_KSTEPS / 2) * (wave >> 1) + ks, f = 32 * (wave & 1) + 16 * t + (lane & 15);
                        int k;  // torch's flattened (channel, position) index of element (k-step s2, q = lane >> 4, j) of the conv kernel's images
                        {
                            const int cw = s2 >> 4, cl = 4 * (s2 & 15) + (lane >> 4);  // the conv kernel's (wave, lane) that stored this piece
                            k = (16 * (cw & 1) + 4 * (cl >> 4) + (j & 3)) * 64 + 16 * (2 * (cw >> 1) + (j >> 2)) + (cl & 15);
                        }
                        put(wave, ks, t, lane, j, std::ldexp((double)W3[(size_t)f * 2048 + k], a3));
                    }
    for (int f = 0; f < 64; ++f) bias[f] = B3[f];
    const int af = scale_of(Wf.data(), 0, 64, Z, 64);
    h->em_winv_fc = (float)std::ldexp(1.0, -af);
    for (int wave = 0; wave < 4; ++wave)
        for (int pz = 0; pz < NPZ; ++pz)
            for (int ks = 0; ks < 2; ++ks)  // (k-steps 2, 3 of a pass stay zero)
                for (int t = 0; t < 2; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int f = 128 * pz + 32 * wave + 16 * t + (lane & 15);
                            put(wave, msiren::EM_C3_KSTEPS / 2 + pz * msiren::EM_FC_KSTEPS + ks, t, lane, j,
                                std::ldexp((double)Wf[(size_t)f * 64 + kin(ks, lane >> 4, j)], af));
                        }
    for (int f = 0; f < Z; ++f) bias[64 + f] = Bf[f];
    }
    for (int l = 0; l < L && mod; ++l) {
        const std::vector<float>& W = *get(h, "modulator.layers." + std::to_string(l) + ".0.weight");
        const std::vector<float>& Bm = *get(h, "modulator.layers." + std::to_string(l) + ".0.bias");
        const int Kh = l == 0 ? 0 : H, K = Kh + Z;
        const int az = scale_of(W.data(), (size_t)Kh, (size_t)K, H, Z);
        h->em_winv_z[l] = (float)std::ldexp(1.0, -az);
        for (int wave = 0; wave < 4; ++wave)
            for (int ph = 0; ph < NPH; ++ph)
                for (int ks = 0; ks < KZ; ++ks)
                    for (int t = 0; t < 2; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int f = 128 * ph + 32 * wave + 16 * t + (lane & 15);
                                put(wave, zp_start + (l * NPH + ph) * KZ + ks, t, lane, j,
                                    std::ldexp((double)W[(size_t)f * K + Kh + kin(ks, lane >> 4, j)], az));
                            }
        if (l > 0) {
            const int ah = scale_of(W.data(), 0, (size_t)K, H, H);
            h->em_winv_h[l] = (float)std::ldexp(1.0, -ah);
            for (int wave = 0; wave < 4; ++wave)
                for (int ph = 0; ph < NPH; ++ph)
                    for (int ks = 0; ks < KH; ++ks)
                        for (int t = 0; t < 2; ++t)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int f = 128 * ph + 32 * wave + 16 * t + (lane & 15);
                                    put(wave, zp_start + L * NPH * KZ + ((l - 1) * NPH + ph) * KH + ks, t, lane, j,
                                        std::ldexp((double)W[(size_t)f * K + kin(ks, lane >> 4, j)], ah));
                                }
        }
        for (int f = 0; f < H; ++f) bias[(size_t)64 + Z + (size_t)l * H + f] = Bm[f];
    }
    if (h->d_emw) HIPCHK(hipFree(h->d_emw));
    h->d_emw = nullptr;
    HIPCHK(hipMalloc(&h->d_emw, ws.size() * 2));
    HIPCHK(hipMemcpy(h->d_emw, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
    int rc;
    if ((rc = upload(&h->d_embias, bias))) return rc;
    h->em_wave_stride = nk * 256;
    h->em_zp_start = zp_start;
    h->em_enc = enc;
    h->em_mod = mod;
    return 0;
}

int pack_fold_weights(msiren_ctx* h) {
    // w[i][j] = exp(-0.1 * dist((i,j), centre)) / max   (src/util/tiling.py:67-88; fp64 maths
    // rounded to fp32 element-wise, then divided by the fp32 maximum, as the reference does)
    const int S = h->S;
    std::vector<float> w((size_t)S * S);
    const double c = (S - 1) / 2.0;
    float mx = 0.f;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const double d = std::sqrt((i - c) * (i - c) + (j - c) * (j - c));
            w[(size_t)i * S + j] = (float)std::exp(-0.1 * d);
            mx = std::max(mx, w[(size_t)i * S + j]);
        }
    for (auto& v : w) v = v / mx;
    return upload(&h->d_foldw, w);
}

}  // namespace mh

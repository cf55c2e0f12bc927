"""The flat state_dict image ("blob", mri_inr_amd/csrc/weights_blob.h) that msiren_weights_export / _import hand over and
msiren_broadcast_weights sends: compiled with g++ and driven on the CPU -- pack -> unpack round trips with every key, with
absent keys (a trunk-only model: no encoder, no modulator), and the refusals (another model, truncated, corrupt)."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = textwrap.dedent(r"""
    #include <cassert>
    #include <cstdio>
    #include <string>
    #include <vector>
    #include "weights_blob.h"
    using namespace msiren;

    static BlobLayout layout_of(int H, int L, int Z) {   // the keys weights_pack.hip:declare_expected lists, shortened
        BlobLayout e;
        e["grid"] = 576 * 2;
        for (int l = 0; l < L; ++l) {
            e["net.layers." + std::to_string(l) + ".weight"] = (size_t)H * (l == 0 ? 2 : H);
            e["net.layers." + std::to_string(l) + ".bias"] = H;
            e["modulator.layers." + std::to_string(l) + ".0.weight"] = (size_t)H * (l == 0 ? Z : H + Z);
            e["modulator.layers." + std::to_string(l) + ".0.bias"] = H;
        }
        e["net.last_layer.weight"] = H;
        e["net.last_layer.bias"] = 1;
        e["encoder.encoder.encoder.7.weight"] = (size_t)Z * 64;
        e["encoder.encoder.encoder.7.bias"] = Z;
        return e;
    }

    static BlobTensors fill(const BlobLayout& e, bool with_encoder, bool with_modulator) {
        BlobTensors t;
        unsigned s = 12345;
        for (const auto& kv : e) {
            if (!with_encoder && kv.first.rfind("encoder.", 0) == 0) continue;
            if (!with_modulator && kv.first.rfind("modulator.", 0) == 0) continue;
            auto& v = t[kv.first];
            v.resize(kv.second);
            for (auto& x : v) { s = s * 1664525u + 1013904223u; x = (float)(int)(s >> 8) * 1e-7f - 0.8f; }
        }
        // values a float-typed transport must not disturb: a NaN payload, -0, a subnormal, infinity
        auto& w = t["net.layers.1.weight"];
        w[0] = blob_word(0x7fc12345u); w[1] = -0.0f; w[2] = blob_word(1u); w[3] = blob_word(0x7f800000u);
        return t;
    }

    static bool same(const BlobTensors& a, const BlobTensors& b) {
        if (a.size() != b.size()) return false;
        for (const auto& kv : a) {
            auto it = b.find(kv.first);
            if (it == b.end() || it->second.size() != kv.second.size()) return false;
            if (std::memcmp(it->second.data(), kv.second.data(), kv.second.size() * 4) != 0) return false;  // bitwise
        }
        return true;
    }

    int main() {
        const BlobLayout e = layout_of(64, 3, 32);
        assert(blob_elems(e) == BLOB_HEADER + e.size() + blob_payload_elems(e));
        for (int variant = 0; variant < 4; ++variant) {
            const BlobTensors src = fill(e, variant & 1, variant & 2);
            std::vector<float> flat(blob_elems(e), 7.f);
            blob_pack(e, src, flat.data());
            BlobTensors dst;
            dst["stale"] = {1.f};  // whatever the receiver held is replaced
            std::string err;
            assert(blob_unpack(e, flat.data(), flat.size(), dst, &err) == 0);
            assert(same(src, dst));
            assert(dst.count("stale") == 0);
            assert((dst.count("encoder.encoder.encoder.7.weight") == 1) == (bool)(variant & 1));
            assert((dst.count("modulator.layers.0.0.weight") == 1) == (bool)(variant & 2));
            // an absent key's slot travels as zeros
            if (!(variant & 1)) {
                size_t off = BLOB_HEADER + e.size(), i = 0;
                for (const auto& kv : e) { if (kv.first == "encoder.encoder.encoder.7.bias") break; off += kv.second; ++i; }
                assert(blob_bits(flat[BLOB_HEADER + i]) == 0u && flat[off] == 0.f);
            }
        }
        // refusals: the receiver keeps its tensors
        const BlobTensors src = fill(e, true, true);
        std::vector<float> flat(blob_elems(e));
        blob_pack(e, src, flat.data());
        BlobTensors keep = fill(e, false, false), before = keep;
        std::string err;
        assert(blob_unpack(e, flat.data(), flat.size() - 1, keep, &err) == -1 && same(keep, before));   // truncated
        assert(blob_unpack(e, flat.data(), 3, keep, &err) == -1);
        const BlobLayout other = layout_of(64, 4, 32);                                                    // another depth
        assert(blob_unpack(other, flat.data(), flat.size(), keep, &err) == -3 && !err.empty() && same(keep, before));
        BlobLayout renamed = e;                                                                            // same sizes, other key
        renamed.erase("grid"); renamed["grie"] = 576 * 2;
        assert(blob_elems(renamed) == blob_elems(e));
        assert(blob_unpack(renamed, flat.data(), flat.size(), keep, &err) == -3);
        std::vector<float> bad = flat;
        bad[0] = 1.0f;                                                                                     // not a blob
        assert(blob_unpack(e, bad.data(), bad.size(), keep, &err) == -2);
        bad = flat;
        bad[BLOB_HEADER + 2] = blob_word(2u);                                                              // flag neither 0 nor 1
        assert(blob_unpack(e, bad.data(), bad.size(), keep, &err) == -4 && same(keep, before));
        // a tensor of the wrong size on the sending side is treated as absent, never read out of bounds
        BlobTensors odd = src;
        odd["net.last_layer.bias"] = {1.f, 2.f, 3.f};
        blob_pack(e, odd, flat.data());
        BlobTensors got;
        assert(blob_unpack(e, flat.data(), flat.size(), got, &err) == 0 && got.count("net.last_layer.bias") == 0);
        std::puts("weights blob ok");
        return 0;
    }
""")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_weights_blob_pack_unpack(tmp_path):
    src = tmp_path / "wb.cpp"
    src.write_text(PROG)
    exe = tmp_path / "wb"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "weights blob ok" in r.stdout, r.stdout + r.stderr

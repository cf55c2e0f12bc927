// Flat image ("blob") of a state_dict: what msiren_weights_export / msiren_weights_import hand over and what
// msiren_broadcast_weights sends through ONE ncclBroadcast (plain C++, no HIP: unit-tested on the CPU by
// tests/test_weights_blob.py).  Stands in for the reference's load_state_dict(torch.load(...)) executed on one
// rank instead of all (test_mod_siren.py:116-118).
//
//   word 0          magic 'MSWB'                         (32-bit patterns stored in float slots: the blob travels
//   word 1          format version                         as ncclFloat32, a broadcast never does arithmetic on it)
//   word 2          number of keys K of the layout
//   word 3          FNV-1a hash of the layout (key names + element counts): both sides must describe the same model
//   word 4, 5       payload elements (low, high 32 bits)
//   words 6..6+K    one presence flag per key (1 = present), keys in the sorted order of the layout
//   then            every key's tensor, in the same order; absent tensors travel as zeros and stay absent
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace msiren {

using BlobLayout = std::map<std::string, size_t>;              // key -> element count (the configuration's)
using BlobTensors = std::map<std::string, std::vector<float>>;  // key -> host copy

constexpr uint32_t BLOB_MAGIC = 0x4257534du;  // "MSWB"
constexpr uint32_t BLOB_VERSION = 1;
constexpr size_t BLOB_HEADER = 6;

inline float blob_word(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t blob_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

inline uint32_t blob_layout_hash(const BlobLayout& layout) {
    uint32_t hsh = 2166136261u;
    auto mix = [&](unsigned char c) { hsh = (hsh ^ c) * 16777619u; };
    for (const auto& kv : layout) {
        for (unsigned char c : kv.first) mix(c);
        mix(0);
        for (int i = 0; i < 8; ++i) mix((unsigned char)(((uint64_t)kv.second >> (8 * i)) & 0xff));
    }
    return hsh;
}

inline size_t blob_payload_elems(const BlobLayout& layout) {
    size_t n = 0;
    for (const auto& kv : layout) n += kv.second;
    return n;
}

inline size_t blob_elems(const BlobLayout& layout) { return BLOB_HEADER + layout.size() + blob_payload_elems(layout); }

// `out` must hold blob_elems(layout) floats.
inline void blob_pack(const BlobLayout& layout, const BlobTensors& tensors, float* out) {
    const size_t payload = blob_payload_elems(layout);
    out[0] = blob_word(BLOB_MAGIC);
    out[1] = blob_word(BLOB_VERSION);
    out[2] = blob_word((uint32_t)layout.size());
    out[3] = blob_word(blob_layout_hash(layout));
    out[4] = blob_word((uint32_t)(payload & 0xffffffffu));
    out[5] = blob_word((uint32_t)((uint64_t)payload >> 32));
    size_t i = BLOB_HEADER, off = BLOB_HEADER + layout.size();
    for (const auto& kv : layout) {
        auto it = tensors.find(kv.first);
        const bool have = it != tensors.end() && it->second.size() == kv.second;
        out[i++] = blob_word(have ? 1u : 0u);
        if (have) std::memcpy(out + off, it->second.data(), kv.second * sizeof(float));
        else std::memset(out + off, 0, kv.second * sizeof(float));
        off += kv.second;
    }
}

// Replaces `tensors` by the blob's tensors.  Returns 0, or a negative code with *err set and `tensors` untouched:
//   -1 blob too short / wrong size, -2 bad magic or version, -3 layout mismatch (another model), -4 corrupt flag.
inline int blob_unpack(const BlobLayout& layout, const float* flat, size_t n, BlobTensors& tensors, std::string* err) {
    auto bad = [&](int code, const char* msg) {
        if (err) *err = msg;
        return code;
    };
    if (n < BLOB_HEADER) return bad(-1, "weights blob is shorter than its header");
    if (blob_bits(flat[0]) != BLOB_MAGIC || blob_bits(flat[1]) != BLOB_VERSION) return bad(-2, "not a weights blob (magic / version)");
    const uint64_t payload = (uint64_t)blob_bits(flat[4]) | ((uint64_t)blob_bits(flat[5]) << 32);
    if (blob_bits(flat[2]) != (uint32_t)layout.size() || blob_bits(flat[3]) != blob_layout_hash(layout) ||
        payload != (uint64_t)blob_payload_elems(layout))
        return bad(-3, "weights blob describes another model (key set / sizes differ from this handle's configuration)");
    if (n != blob_elems(layout)) return bad(-1, "weights blob has the wrong length for this handle's configuration");
    size_t i = BLOB_HEADER;
    for (size_t k = 0; k < layout.size(); ++k)
        if (blob_bits(flat[i + k]) > 1u) return bad(-4, "weights blob is corrupt (presence flag is neither 0 nor 1)");
    BlobTensors fresh;
    size_t off = BLOB_HEADER + layout.size();
    for (const auto& kv : layout) {
        if (blob_bits(flat[i++]) == 1u) fresh[kv.first].assign(flat + off, flat + off + kv.second);
        off += kv.second;
    }
    tensors.swap(fresh);
    return 0;
}

}  // namespace msiren

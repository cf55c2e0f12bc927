// The steps either side of the model call (reference: src/util/tiling.py), as gather kernels:
// every output element is computed by one thread from at most a handful of inputs, so there are
// no atomics and results do not depend on scheduling.  All of this is HBM-bound byte moving:
// the kernels only have to keep accesses coalesced along the fastest output axis.
#pragma once
#include <hip/hip_runtime.h>

namespace msiren {

// dst[r, 0:HP] = {src[r, 0:H], 0...}
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int H, int HP) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * HP) return;
    const int64_t r = i / HP;
    const int c = (int)(i - r * HP);
    dst[i] = c < H ? src[r * H + c] : 0.f;
}

// image_to_patches (tiling.py:10-64): reflect-pad by `pad` on every side plus bottom/right up to a
// multiple of I, then O x O windows at stride I, row-major over (nV, nH).
__global__ void image_to_patches_kernel(const float* __restrict__ img, float* __restrict__ patches, int64_t n, int Hh, int Ww,
                                        int nV, int nH, int O, int I, int pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_slice = (int64_t)nV * nH * O * O;
    if (i >= n * per_slice) return;
    const int64_t s = i / per_slice;
    int64_t r = i - s * per_slice;
    const int x = (int)(r % O);
    r /= O;
    const int y = (int)(r % O);
    r /= O;
    const int hh = (int)(r % nH);
    const int v = (int)(r / nH);
    int sy = v * I + y - pad, sx = hh * I + x - pad;
    sy = sy < 0 ? -sy : sy;
    sy = sy >= Hh ? 2 * (Hh - 1) - sy : sy;
    sx = sx < 0 ? -sx : sx;
    sx = sx >= Ww ? 2 * (Ww - 1) - sx : sx;
    patches[i] = img[(s * Hh + sy) * Ww + sx];
}

// classify_patches (tiling.py:184-198): mean < 1e-10  ->  black (flag 1).  One workgroup per patch.
__global__ __launch_bounds__(256) void black_flags_kernel(const float* __restrict__ patches, int* __restrict__ flags, int elems) {
    __shared__ float red[4];
    const float* p = patches + (size_t)blockIdx.x * elems;
    float s = 0.f;
    for (int i = threadIdx.x; i < elems; i += 256) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)elems;
        flags[blockIdx.x] = mean < 1e-10f ? 1 : 0;
    }
}

// filter_and_remember_black_patches (tiling.py:244-271) as a device-side plan: the non-black patches in
// order.  plan[0] = their number, plan[1] = plan[0] * units_per_patch (the trunk's unit count),
// plan[2 + j] = index of the j-th kept patch, plan[2 + n + b] = position of patch b among the kept ones
// (-1: black).  Encoder, modulator and trunk then work on the kept patches only -- like the reference,
// which never evaluates the model on a black tile -- without the host learning the count.
// One workgroup of 256 threads (one wave per SIMD, few registers: it has to fit beside a resident trunk
// workgroup of the other stream); n is a few hundred per slice.
__device__ __forceinline__ void compact_flags_block(const int* __restrict__ flags, int n, int units_per_patch, int* __restrict__ plan) {
    __shared__ int wsum[4];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int b = base + tid;
        const int keep = (b < n && flags[b] == 0) ? 1 : 0;
        int incl = keep;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int j = before + incl - keep;
        if (b < n) {
            plan[2 + n + b] = keep ? j : -1;
            if (keep) plan[2 + j] = b;
        }
        __syncthreads();
        if (tid == 255) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        plan[0] = carry;
        plan[1] = carry * units_per_patch;
    }
}

__global__ __launch_bounds__(256) void compact_flags_kernel(const int* __restrict__ flags, int n, int units_per_patch, int* __restrict__ plan) {
    compact_flags_block(flags, n, units_per_patch, plan);
}

// image_to_patches + black_flags + compact_flags in ONE launch (round 5: the slice pipeline is a chain of small dependent launches in
// front of the trunk, each worth ~2.5 us of kernel and ~2 us of gap).  One workgroup per patch: gather it from the image (img == null:
// the patches are given), sum it in black_flags_kernel's order (the same flag, bit for bit), take a ticket; the workgroup that draws the
// last ticket -- every flag has been written and fenced by then -- builds the plan (compact_flags_block) and puts the ticket counter back.
struct TilingPlanParams {
    const float* img;   // (n, Hh, Ww) or null
    float* patches;     // (NP, O, O): written when img is given, read otherwise
    int* flags;         // (NP)
    int* plan;          // (2 + 2 NP)
    unsigned* ticket;   // zero between launches
    int n, Hh, Ww, nV, nH, O, I, pad, NP, units_per_patch;
};
__global__ __launch_bounds__(256) void patches_flags_plan_kernel(TilingPlanParams p) {
    __shared__ float red[4];
    __shared__ int last;
    const int b = blockIdx.x, tid = threadIdx.x, elems = p.O * p.O;
    float* dst = p.patches + (size_t)b * elems;
    float s = 0.f;
    if (p.img) {
        const int per = p.nV * p.nH, sl = b / per, r = b - sl * per, v = r / p.nH, hh = r - v * p.nH;
        const float* im = p.img + (size_t)sl * p.Hh * p.Ww;
        for (int i = tid; i < elems; i += 256) {
            const int y = i / p.O, x = i - y * p.O;
            int sy = v * p.I + y - p.pad, sx = hh * p.I + x - p.pad;
            sy = sy < 0 ? -sy : sy;
            sy = sy >= p.Hh ? 2 * (p.Hh - 1) - sy : sy;
            sx = sx < 0 ? -sx : sx;
            sx = sx >= p.Ww ? 2 * (p.Ww - 1) - sx : sx;
            const float val = im[(size_t)sy * p.Ww + sx];
            dst[i] = val;
            s += val;
        }
    } else {
        for (int i = tid; i < elems; i += 256) s += dst[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)elems;
        p.flags[b] = mean < 1e-10f ? 1 : 0;
        __threadfence();  // the flag is visible device-wide before the ticket is
        last = atomicAdd(p.ticket, 1u) == (unsigned)(p.NP - 1);
    }
    __syncthreads();
    if (!last) return;  // workgroup-uniform
    __threadfence();      // (acquire: the other workgroups' flags)
    compact_flags_block(p.flags, p.NP, p.units_per_patch, p.plan);
    if (tid == 0) *p.ticket = 0u;
}

// filter_and_remember_black_patches' `patches[non_black_indices]` (tiling.py:268) and reintegrate_black_patches' copy-back
// (tiling.py:294-301) as row copies through an index list: gather: dst[j] = src[idx[j]]; scatter: dst[idx[j]] = src[j]
// (the caller zero-fills dst first: black rows stay zeros).  One workgroup per row.
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ idx,
                                                        int elems, int scatter) {
    const int j = blockIdx.x, r = idx[j];
    const float* s = src + (size_t)(scatter ? j : r) * elems;
    float* d = dst + (size_t)(scatter ? r : j) * elems;
    for (int i = threadIdx.x; i < elems; i += 256) d[i] = s[i];
}

// patches_to_image_weighted_average (tiling.py:91-140): fold(tiles * w) / fold(w) with kernel S,
// stride I, padding `pad`.  Black patches (flags[b] != 0) contribute zeros with their full weight,
// as reintegrate_black_patches + fold do in the reference (tiling.py:287-301, :117-118).
// `pos` (optional): tiles holds the kept patches only, patch b at row pos[b] (compact_flags_kernel).
// `reset_word` (optional): set to 0 by the first thread -- the pass counter of a launch whose number of passes only the device knew
// (one memset node less behind the trunk).
__global__ void weighted_fold_kernel(const float* __restrict__ tiles, const float* __restrict__ w, float* __restrict__ recon,
                                     const int* __restrict__ flags, const int* __restrict__ pos, int64_t n, int nV, int nH, int S, int I, int pad,
                                     int* __restrict__ reset_word = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (reset_word && i == 0) *reset_word = 0;
    const int OH = nV * I, OW = nH * I;
    if (i >= n * OH * (int64_t)OW) return;
    const int64_t s = i / ((int64_t)OH * OW);
    const int r = (int)(i - s * (int64_t)OH * OW);
    const int Y = r / OW, Xc = r - Y * OW;
    const int py = Y + pad, px = Xc + pad;
    int v0 = py - S + 1;
    v0 = v0 <= 0 ? 0 : (v0 + I - 1) / I;
    int h0 = px - S + 1;
    h0 = h0 <= 0 ? 0 : (h0 + I - 1) / I;
    const int v1 = min(nV - 1, py / I), h1 = min(nH - 1, px / I);
    float num = 0.f, den = 0.f;
    for (int v = v0; v <= v1; ++v)
        for (int hh = h0; hh <= h1; ++hh) {
            const int ty = py - v * I, tx = px - hh * I;
            const int64_t b = (s * nV + v) * nH + hh;
            const float ww = w ? w[ty * S + tx] : 1.f;  // w == nullptr: plain overlap average (patches_to_image, tiling.py:143-181)
            den += ww;
            if (!flags || flags[b] == 0) num += tiles[((pos ? (int64_t)pos[b] : b) * S + ty) * S + tx] * ww;
        }
    recon[i] = num / den;
}

}  // namespace msiren

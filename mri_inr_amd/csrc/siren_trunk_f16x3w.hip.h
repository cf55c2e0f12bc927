// Fused modulated-SIREN trunk for gfx950, split-fp16 arithmetic, WEIGHT-STATIONARY form ("f16x3w").
//
// Same arithmetic as siren_trunk_f16x3n.hip.h (three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation, the same
// packed weight stream, the same per-accumulator order of products -- hidden layers come out bit for bit the same),
// different data flow (reference maths unchanged: src/networks/modulated_siren.py:215-233):
//
//   * a wave owns 64 OUTPUT FEATURES (wave w: features 64w .. 64w+63) and keeps their hi/lo weights of the current
//     layer, all 256 input features of them, in the accumulator half of the register file: 4 feature tiles x 8 k-steps
//     x {hi, lo} fragments of 4 registers = 256 AGPRs, read by the MFMAs as their A operand;
//   * the ACTIVATIONS go through LDS: a unit (32 coordinates of one patch) is a 32 KB image of ready-made B
//     fragments  [8 k-steps][2 column groups][hi|lo][64 lanes][8 x f16]; all four waves read it (4 x ds_read_b128 per
//     24 MFMAs -- half the LDS traffic per MFMA of the register-resident kernel, where every wave re-reads every
//     weight fragment for its own 32 coordinates), and each wave writes the 64 features it has produced for the next
//     layer back IN PLACE (its two k-steps of the image) behind one workgroup barrier per unit and layer;
//   * a workgroup takes a PASS of nb = 2..4 units through the layers together ("slots" (layer, unit), executed
//     layer by layer); the next layer's weights are fetched from L2 straight into the registers of the fragments as the
//     layer's last unit retires them (64 x global_load_dwordx4 per wave and layer, issued 8 per k-step, consumed a
//     whole slot later behind counted vmcnt waits) -- no weight ring, no LDS DMA;
//   * the slot is the unit of software pipelining: slot n's 192 MFMAs run beside the epilogue of slot n-1 (sine,
//     modulation, fp16 split, fragment stores), beside the final layer's dot product with last_layer.weight, and beside
//     layer 0 of the NEXT pass (a per-weight-set table of act0(W0 x_p + b0), modulated and split) -- a pass has no
//     serial prologue, so its duration is proportional to its number of units;
//   * which is what the schedule exploits: passes come from the device-wide queue as before, but the host lays them out
//     as full passes of 4 units first and passes of 3 and 2 units at the end (ws_schedule), so the last round of a
//     launch is cut to the work that is left -- a 320x320 slice (7200 units on 256 CUs = 7.03 rounds of 4-unit passes,
//     which cost the register-resident kernel 8 rounds) ends after 7.25.
//
// One barrier per slot, at the start of its last k-step: by then the slot's own LDS stores (epilogue of the slot
// before) are complete and every wave has issued its last read of the unit image, so the image may be overwritten in
// the next slot and the images written in this slot may be read from the next slot's first k-step on.
#pragma once
#include <hip/hip_runtime.h>

#include "siren_trunk_f16x3n.hip.h"
#include "siren_trunk_f16x3w_gaps.hip.h"  // generated: tools/gen_ws_gaps.py

namespace msiren {

// depths the kernel is built and tested for: its unit images + (L + 1)-row modulation tables must fit the 160 KB of LDS
// (L = 6 would need 166 992 B), and a pass needs a hidden layer on either side of the one in flight
constexpr int WS_MIN_L = 3, WS_MAX_L = 5;

// passes [0, n4) take 4 units each, [n4, n4 + n3) take 3, [n4 + n3, npasses) take 2 (the very last one may reach past
// the end of the batch: its surplus unit is computed on clamped inputs and not stored)
struct WsSchedule {
    int n4, n3, n2;
    __host__ __device__ int npasses() const { return n4 + n3 + n2; }
};

// All rounds but the last full one as 4-unit passes on every workgroup; what is left (4..8 units per workgroup, or all
// of a small batch) as two waves of equal-or-smaller passes, so that the workgroups finish within one unit of each
// other.  Pass sizes never increase along the queue (the in-place pipeline relies on it).
__host__ __device__ inline WsSchedule ws_schedule(long long units, int grid) {
    WsSchedule s{0, 0, 0};
    if (units <= 0 || grid <= 0) return s;
    long long full = units / (4LL * grid);           // whole rounds of 4-unit passes
    long long lead = full > 0 ? full - 1 : 0;        // keep the last one back
    s.n4 = (int)(lead * grid);
    long long rem = units - 4LL * s.n4;              // < 8 * grid
    const long long per = (rem + grid - 1) / grid;   // units per workgroup still to hand out: 1..8
    // first wave of passes: `a` units each on every workgroup, then the rest in passes of `b`
    int a, b;
    switch ((int)per) {
        case 8: a = 4; b = 4; break;
        case 7: a = 4; b = 3; break;
        case 6: a = 3; b = 3; break;
        case 5: a = 3; b = 2; break;
        case 4: a = 4; b = 4; break;   // (b unused unless rem > 4 * grid, which per == 4 excludes)
        case 3: a = 3; b = 3; break;
        default: a = 2; b = 2; break;  // 1 or 2 units per workgroup
    }
    auto add = [&](int size, long long count) {
        if (size == 4) s.n4 += (int)count;
        else if (size == 3) s.n3 += (int)count;
        else s.n2 += (int)count;
    };
    long long first = rem / a < grid ? rem / a : grid;  // passes of `a` units (never more than one per workgroup)
    if (per <= 4) first = (rem + a - 1) / a;            // a single wave: everything in passes of `a`, last one padded
    add(a, first);
    rem -= first * a;
    if (rem > 0) add(b, (rem + b - 1) / b);
    return s;
}

struct TrunkWsParams {
    const float* s0t;         // (64, P, 4): layer-0 activations act0(W0 x_p + b0), feature-group major
    const _Float16* wp;       // [(L-1)*8 chunks][8 k-steps][2 sub-tiles][hi|lo][64 lanes][8] (the f16x3n stream)
    const float* bias;        // (L-1, 256) in revolutions
    const float* wout;        // (256) * w0/2pi
    const float* mods;        // (L, B, 256)
    float* out;               // (B, P)
    float* dump;              // >= 32 floats: where the surplus unit of a padded pass stores
    float mscale[16];         // factor of each layer's modulation row (the next layer's weight scale, inverted)
    float bout, cg0, cg;
    int B, P, L, units_per_patch, total_units, unit_base;
    unsigned div_m, div_k;    // unit / units_per_patch == (unit * div_m) >> div_k   (units < 2^30)
    const int* plan;          // optional (compact_flags_kernel): the unit count is plan[1] (<= total_units)
    int* pass_counter;        // work queue (never reset: the host passes the value it holds at launch)
    unsigned pass_base;
    unsigned long long* stamps;
    int* status;              // domain guard (f16_out_of_range): the stream's flag word (device memory), may be null
    int status_val;           // what a launch that meets an out-of-range modulation writes there: its own number
};

template <int NB>
struct WsLds {  // byte offsets into dynamic LDS
    static constexpr int act = 0;  // NB unit images of 32 KB
    static constexpr int bias = NB * 32768;  // (L-1) x 256 floats
    static __host__ __device__ constexpr int wout(int L) { return bias + (L - 1) * 1024; }   // 256 floats
    static __host__ __device__ constexpr int mods(int L) { return wout(L) + 1024; }          // NB units x (L+1) rows x 256 floats
    static __host__ __device__ constexpr int red(int L) { return mods(L) + NB * (L + 1) * 1024; }  // 2 x 32 coordinates x 4 waves floats
    static __host__ __device__ constexpr int queue(int L) { return red(L) + 1024; }          // 2 ints
    static __host__ __device__ constexpr int mscale(int L) { return queue(L) + 16; }         // 16 floats
    static __host__ __device__ constexpr int total(int L) { return mscale(L) + 64; }
};

__device__ __forceinline__ h8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const h8*>(p); }

template <int ACT, int NB, int DBG = 0>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(192))) void siren_trunk_f16x3w_kernel(TrunkWsParams p) {
    using LY = WsLds<NB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;    // which four features of a 16-feature tile this lane holds
    const int n16 = lane & 15;  // coordinate inside a 16-column group
    const int L = p.L;
    const int P = p.P;
    const int NR = L + 1;  // rows of a unit's modulation table: L in use + the one the next pass starts in

    const int total_units = __builtin_amdgcn_readfirstlane(p.plan ? p.plan[1] : p.total_units);
    const WsSchedule sch = ws_schedule(total_units, (int)gridDim.x);
    const unsigned npasses = (unsigned)sch.npasses();
    int cur_pass = (int)blockIdx.x;
    if ((unsigned)cur_pass >= npasses) return;
    // diagnostic instance: launch-level marks, s_memrealtime (100 MHz, comparable across CUs) in entry [7] of the workgroup's
    // first three slot records: [0][7] workgroup entry, [1][7] prologue done (first MFMA next), [2][7] workgroup done
    auto mark = [&](int which) {
        if constexpr (DBG) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) p.stamps[((size_t)blockIdx.x * 96 + which) * 8 + 7] = t;
        }
    };
    mark(0);

    // ---- per-lane bases (every LDS access below is base + compile-time constant, or + one per-slot scalar) -----------
    unsigned char* const actL = smem + LY::act + lane * 16;                       // + unit * 32768 + fragment offset
    const unsigned char* const biasL = smem + LY::bias + wave * 256 + q * 16;     // + (l-1) * 1024 + t * 64
    unsigned char* const modsW = smem + LY::mods(L) + wave * 256;                 // this wave's 64 features of a row
    const unsigned char* const modsL = modsW + q * 16;                            // + (unit * (L+1) + row) * 1024 + t * 64
    float* const redW = reinterpret_cast<float*>(smem + LY::red(L));              // [parity][coordinate 0..31][wave]
    int* const qslot = reinterpret_cast<int*>(smem + LY::queue(L));  // (plain LDS accesses: a volatile one became a flat op + vmcnt(0))
    float* const mscaleT = reinterpret_cast<float*>(smem + LY::mscale(L));

    // pass id -> (first unit, number of units)
    auto pass_units = [&](int id, int& u0, int& nb) {
        if (id < sch.n4) { u0 = 4 * id; nb = 4; }
        else if (id < sch.n4 + sch.n3) { u0 = 4 * sch.n4 + 3 * (id - sch.n4); nb = 3; }
        else { u0 = 4 * sch.n4 + 3 * sch.n3 + 2 * (id - sch.n4 - sch.n3); nb = 2; }
    };
    // unit (clamped into the batch) -> patch, first coordinate, whether it exists
    auto unit_info = [&](int u, int& patch, int& c0, bool& live) {
        live = u < total_units;
        const unsigned uu = (unsigned)((live ? u : total_units - 1) + p.unit_base);
        patch = (int)(((unsigned long long)uu * p.div_m) >> p.div_k);
        c0 = ((int)uu - patch * p.units_per_patch) * 32;
    };

    // ---- the weights of the layer in flight: A fragments (tile t, k-step s, hi|lo) in AGPRs a[0:255], BY NAME --------
    // fragment (t, s, hl) = a[32 s + 8 t + 4 hl .. + 3].  The accumulator half of the register file is managed by hand:
    // the loads, the waits and the MFMAs below are asm statements that name these registers, the compiler never sees a
    // weight value (left to the register allocator the 256 loop-carried fragments were split and spilled: 345-644 spills
    // in every formulation tried).  Arch VGPRs (accumulators, B fragments, epilogue) stay the compiler's.
    const unsigned woff = (unsigned)lane * 16u, woff2 = woff + 32768u;  // tiles 0, 1 / tiles 2, 3 (the next 32 KB chunk)
    // chunk (layer l, 32-feature tile T) of the stream is 32 KB: [k-step s][sub-tile u][hi|lo][lane][8]; this wave's
    // tiles t = 0..3 are (T = 2 wave + (t >> 1), u = t & 1)
    auto wlayer = [&](int l) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(p.wp) + ((size_t)(l - 1) * 8 + 2 * wave) * 32768;
    };
#define MSIREN_WS_A(S, T, HL) (32 * (S) + 8 * (T) + 4 * (HL))
// The 8 fragments of k-step S: two runs of 4 KB of the stream (T = 2 wave, 2 wave + 1), [u][hi|lo] each.  Issued by a
// layer's LAST unit, behind the MFMAs of k-step S that retire them.  (No branch in the MFMA stream: which slots load and
// which wait is a compile-time property of the slot body -- FL below.)
#define MSIREN_WS_LOADK(S, WB)                                                                                         \
    asm volatile("global_load_dwordx4 a[%3:%4], %0, %1 offset:0\n\t"                                                   \
                 "global_load_dwordx4 a[%5:%6], %0, %1 offset:1024\n\t"                                                \
                 "global_load_dwordx4 a[%7:%8], %0, %1 offset:2048\n\t"                                                \
                 "global_load_dwordx4 a[%9:%10], %0, %1 offset:3072\n\t"                                               \
                 "global_load_dwordx4 a[%11:%12], %0, %2 offset:0\n\t"                                                 \
                 "global_load_dwordx4 a[%13:%14], %0, %2 offset:1024\n\t"                                              \
                 "global_load_dwordx4 a[%15:%16], %0, %2 offset:2048\n\t"                                              \
                 "global_load_dwordx4 a[%17:%18], %0, %2 offset:3072"                                                  \
                 :                                                                                                     \
                 : "v"(woff), "s"((WB) + (S) * 4096), "s"((WB) + 32768 + (S) * 4096),                                  \
                   "n"(MSIREN_WS_A(S, 0, 0)), "n"(MSIREN_WS_A(S, 0, 0) + 3), "n"(MSIREN_WS_A(S, 0, 1)), "n"(MSIREN_WS_A(S, 0, 1) + 3), \
                   "n"(MSIREN_WS_A(S, 1, 0)), "n"(MSIREN_WS_A(S, 1, 0) + 3), "n"(MSIREN_WS_A(S, 1, 1)), "n"(MSIREN_WS_A(S, 1, 1) + 3), \
                   "n"(MSIREN_WS_A(S, 2, 0)), "n"(MSIREN_WS_A(S, 2, 0) + 3), "n"(MSIREN_WS_A(S, 2, 1)), "n"(MSIREN_WS_A(S, 2, 1) + 3), \
                   "n"(MSIREN_WS_A(S, 3, 0)), "n"(MSIREN_WS_A(S, 3, 0) + 3), "n"(MSIREN_WS_A(S, 3, 1)), "n"(MSIREN_WS_A(S, 3, 1) + 3)  \
                 : "memory")
// One fragment (T, S, HL) of the next layer, issued behind the last MFMA that reads the register (see MSIREN_WS_MFMAS):
// eight back-to-back loads at the end of a region stalled the wave for ~130 cycles (the address path takes a 1 KB
// instruction every ~16 cycles); spread over the region they cost nothing.
#define MSIREN_WS_LOAD1(S, T, HL, WB)                                                                                  \
    asm volatile("global_load_dwordx4 a[%2:%3], %0, %1 offset:%4"                                                      \
                 :                                                                                                     \
                 : "v"((T) < 2 ? woff : woff2), "s"((WB) + (S) * 4096), "n"(MSIREN_WS_A(S, T, HL)),                    \
                   "n"(MSIREN_WS_A(S, T, HL) + 3), "n"(((T) & 1) * 2048 + (HL) * 1024)                                \
                 : "memory")
// k-step S's fragments have landed once at most 8 (7 - S) younger loads are outstanding (loads return in order; any
// other vector-memory operation issued since only makes the wait stricter).  Issued by a layer's FIRST unit (the slot
// behind the loading one; passes have >= 2 units, so a slot never does both).  asm volatile statements keep their
// order: the MFMAs that read the fragments are the statements behind it.
#define MSIREN_WS_WAITK(S) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(8 * (7 - (S))) : "memory")
// D = A * B + C with A = fragment (T, S, HL).  The first product of an accumulator takes the bias as C.
// The 16 accumulators (2 slot parities x 4 tiles x 2 column groups, 4 registers each) are v[192:255], BY NAME as well: the
// kernel is compiled with amdgpu_num_vgpr(192), so the compiler's own allocation ends at v191 / a191 and nothing it does
// can touch them.  As C++ values they were a loop-carried 64-register phi web over the twelve slot bodies: 16 v_mov_b64 of
// reconciliation at the end of every body; pinned with register constraints: 231 spills.
#define MSIREN_WS_V(PAR, T, G) (192 + 32 * (PAR) + 8 * (T) + 4 * (G))
#if defined(MSIREN_WS_ABL) && (MSIREN_WS_ABL & 8)  /* ablation (timing only): no MFMAs at all -- what is left of a slot is its control flow */
#define MSIREN_WS_MFMA(PAR, T, G, S, HL, B) asm volatile("" : : "v"(B))
#define MSIREN_WS_MFMA0(PAR, T, G, S, HL, B, C) asm volatile("" : : "v"(B), "v"(C))
#else
#define MSIREN_WS_MFMA(PAR, T, G, S, HL, B)                                                                            \
    asm volatile("v_mfma_f32_16x16x32_f16 v[%1:%2], a[%3:%4], %0, v[%1:%2]"                                            \
                 : : "v"(B), "n"(MSIREN_WS_V(PAR, T, G)), "n"(MSIREN_WS_V(PAR, T, G) + 3), "n"(MSIREN_WS_A(S, T, HL)), \
                   "n"(MSIREN_WS_A(S, T, HL) + 3))
#define MSIREN_WS_MFMA0(PAR, T, G, S, HL, B, C)                                                                        \
    asm volatile("v_mfma_f32_16x16x32_f16 v[%2:%3], a[%4:%5], %0, %1"                                                  \
                 : : "v"(B), "v"(C), "n"(MSIREN_WS_V(PAR, T, G)), "n"(MSIREN_WS_V(PAR, T, G) + 3),                     \
                   "n"(MSIREN_WS_A(S, T, HL)), "n"(MSIREN_WS_A(S, T, HL) + 3))
#endif

    // Keeping the register allocator OUT of the accumulator file: under pressure it splits arch-VGPR values into AGPRs
    // (v_accvgpr_write / _read), i.e. over the fragments (seen: a pointer parked in a0..a3, then a fault).  a[192:255] are
    // beyond its limit (amdgpu_num_vgpr(192)); for a[0:191], 48 placeholder values of AGPR class, "defined" before the loop
    // and "used" (by empty asm statements) in every k-step region, keep all of them allocated as far as the compiler can
    // tell; which placeholder sits in which register is irrelevant -- the statements above name the registers themselves.
    // The build checks the result (no v_accvgpr, no scratch: tests/test_register_budget.py).
    asm volatile("; v[192:255] accumulators, a[0:255] weight fragments" ::: "v255", "a255");  // the kernel's register counts
    h8 wres[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) asm volatile("; placeholder" : "=a"(wres[i]));
#define MSIREN_WS_HOLD()                                                                                               \
    do {                                                                                                               \
        asm volatile("" ::"a"(wres[0]), "a"(wres[1]), "a"(wres[2]), "a"(wres[3]), "a"(wres[4]), "a"(wres[5]), "a"(wres[6]), "a"(wres[7]),     \
                     "a"(wres[8]), "a"(wres[9]), "a"(wres[10]), "a"(wres[11]), "a"(wres[12]), "a"(wres[13]), "a"(wres[14]), "a"(wres[15]),    \
                     "a"(wres[16]), "a"(wres[17]), "a"(wres[18]), "a"(wres[19]), "a"(wres[20]), "a"(wres[21]), "a"(wres[22]), "a"(wres[23])); \
        asm volatile("" ::"a"(wres[24]), "a"(wres[25]), "a"(wres[26]), "a"(wres[27]), "a"(wres[28]), "a"(wres[29]), "a"(wres[30]), "a"(wres[31]), \
                     "a"(wres[32]), "a"(wres[33]), "a"(wres[34]), "a"(wres[35]), "a"(wres[36]), "a"(wres[37]), "a"(wres[38]), "a"(wres[39]), \
                     "a"(wres[40]), "a"(wres[41]), "a"(wres[42]), "a"(wres[43]), "a"(wres[44]), "a"(wres[45]), "a"(wres[46]), "a"(wres[47])); \
    } while (0)

    // ---- per-slot state -------------------------------------------------------------------------------------------------
    h8 Bf[2][4];         // B fragments of the k-step in flight / the next one: [k-step parity][hi g0, lo g0, hi g1, lo g1]
    f32x4 bia[4];        // bias (C operand) of the slot's layer, per tile

    // current pass / next pass
    int u0_cur, nb_cur, u0_nxt = 0, nb_nxt = 0;
    pass_units(cur_pass, u0_cur, nb_cur);
    int row0 = 0;  // row of the modulation table that holds layer 0 of the current pass (rows rotate modulo L + 1)

    // modulation rows of one unit: this wave's 64 features of the L rows (L <= 8), scaled, into table rows (r0 + l) mod NR.
    // Lane (row = lane >> 4, i = lane & 15) moves features 4i..4i+3 of rows `row` and 4 + `row`: two loads, 8 registers.
    float msc0 = 1.f, msc1 = 1.f;  // mscaleT[q], mscaleT[4 + q]: set once the tables are in LDS
    // last_layer.weight of the four features a lane stages (kept in registers: an LDS read at a slot boundary costs its
    // whole latency, and the wait for it also waits for the B fragments prefetched for the next slot)
    const f32x4 wrow = *reinterpret_cast<const f32x4*>(p.wout + wave * 64 + n16 * 4);
    // (global addresses below are a wave-uniform base + a 32-bit per-lane byte offset: one VGPR instead of a 64-bit pair)
    // Rows beyond L are clamped (read, not stored).  The loads are asm with a counted wait of their own: the compiler's wait
    // for a load of its own is vmcnt(0) here (it cannot see the weight loads), i.e. a drain of everything in flight.
    const unsigned mods_off0 = ((unsigned)(q < L ? q : L - 1) * (unsigned)p.B * 256u + (unsigned)n16 * 4u) * 4u;
    const unsigned mods_off1 = ((unsigned)(4 + q < L ? 4 + q : L - 1) * (unsigned)p.B * 256u + (unsigned)n16 * 4u) * 4u;
    auto mods_fetch = [&](int patch, f32x4 (&m)[2]) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.mods + (size_t)patch * 256 + wave * 64);  // uniform
        asm volatile("global_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4"
                     : "=&v"(m[0]), "=&v"(m[1]) : "v"(mods_off0), "v"(mods_off1), "s"(src) : "memory");
    };
    auto mods_store = [&](int slot_unit, int r0, const f32x4 (&m)[2]) {
        int ra = r0 + q, rb = r0 + 4 + q;
        ra = ra >= NR ? ra - NR : ra;
        rb = rb >= NR ? rb - NR : rb;
        rb = rb >= NR ? rb - NR : rb;
        f32x4 ma = m[0] * msc0, mb = m[1] * msc1;
        if ((f16_out_of_range(ma) || f16_out_of_range(mb)) && p.status) *p.status = p.status_val;  // (clamped rows repeat a checked one)
        // the final layer's row only ever meets last_layer.weight (its scale is 1): the table holds the product
        if (q == L - 1) ma *= wrow;
        if (4 + q == L - 1) mb *= wrow;
        if (q < L) *reinterpret_cast<f32x4*>(modsW + (slot_unit * NR + ra) * 1024 + n16 * 16) = ma;
        if (4 + q < L) *reinterpret_cast<f32x4*>(modsW + (slot_unit * NR + rb) * 1024 + n16 * 16) = mb;
    };

    // layer 0 of one unit, this wave's 64 features: table -> x modulation -> fp16 split -> the unit image (k-steps 2 wave, 2 wave + 1)
    const unsigned char* const s0w = reinterpret_cast<const unsigned char*>(p.s0t) + (size_t)(16 * wave) * P * 16;  // uniform
    const unsigned s0q = (unsigned)q * (unsigned)P * 16u;
    auto l0_load = [&](int c0, f32x4 (&raw)[4][2]) {
        int pc0 = c0 + n16, pc1 = c0 + 16 + n16;
        pc0 = pc0 < P ? pc0 : P - 1;
        pc1 = pc1 < P ? pc1 : P - 1;
        const unsigned o0 = s0q + (unsigned)pc0 * 16u, o1 = s0q + (unsigned)pc1 * 16u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            raw[t][0] = *reinterpret_cast<const f32x4*>(s0w + (size_t)(4 * t) * P * 16 + o0);
            raw[t][1] = *reinterpret_cast<const f32x4*>(s0w + (size_t)(4 * t) * P * 16 + o1);
        }
    };
    // the same, one k-step's worth (tiles 2u, 2u+1) at a time, as asm loads with counted waits of their own (MSIREN_WS_RAW0_WAIT):
    // what the final slots issue beside their MFMAs.  (A load of the compiler's own is waited for with a count that ignores
    // the weight loads in flight -- in a loading slot that is a drain of all of them.)
    auto l0_load_half = [&](int c0, int u, f32x4 (&raw)[4][2]) {
        int pc0 = c0 + n16, pc1 = c0 + 16 + n16;
        pc0 = pc0 < P ? pc0 : P - 1;
        pc1 = pc1 < P ? pc1 : P - 1;
        const unsigned o0 = s0q + (unsigned)pc0 * 16u, o1 = s0q + (unsigned)pc1 * 16u;
        const unsigned char* ba = s0w + (size_t)(8 * u) * P * 16;       // tile 2u
        const unsigned char* bb = s0w + (size_t)(8 * u + 4) * P * 16;   // tile 2u + 1
        asm volatile("global_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %5, %6\n\t"
                     "global_load_dwordx4 %2, %4, %7\n\tglobal_load_dwordx4 %3, %5, %7"
                     : "=&v"(raw[2 * u][0]), "=&v"(raw[2 * u][1]), "=&v"(raw[2 * u + 1][0]), "=&v"(raw[2 * u + 1][1])
                     : "v"(o0), "v"(o1), "s"(ba), "s"(bb)
                     : "memory");
    };
    auto l0_store = [&](int slot_unit, int row, const f32x4 (&raw)[4][2]) {
        const unsigned char* mr = modsL + (slot_unit * NR + row) * 1024;
        unsigned char* img = actL + slot_unit * 32768 + (2 * wave) * 4096;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(mr + (2 * u) * 64);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(mr + (2 * u + 1) * 64);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                fp16x2 h0, h1, h2, h3, l0, l1, l2, l3;
                const f32x4 a0 = raw[2 * u][g], a1 = raw[2 * u + 1][g];
                split_products_pk(a0[0], m0[0], a0[1], m0[1], h0, l0);
                split_products_pk(a0[2], m0[2], a0[3], m0[3], h1, l1);
                split_products_pk(a1[0], m1[0], a1[1], m1[1], h2, l2);
                split_products_pk(a1[2], m1[2], a1[3], m1[3], h3, l3);
                h8 ph = pack_h8(h0, h1, h2, h3), pl = pack_h8(l0, l1, l2, l3);
                asm volatile("s_nop 0" : "+v"(ph), "+v"(pl));  // (a half-register write is not stored by the very next instruction)
                *reinterpret_cast<h8*>(img + u * 4096 + (2 * g) * 1024) = ph;
                *reinterpret_cast<h8*>(img + u * 4096 + (2 * g + 1) * 1024) = pl;
            }
        }
    };

    // ---- prologue (once per workgroup, not overlapped): constant tables, layer-1 weights, layer 0 of the first pass ---
    // 7.2 us of a 273 us single-slice launch (tools/timeline_ws_launch.py, profiles/r4/): 2.6 %.  Round 4 rebuilt it so that
    // everything -- modulation rows of all units, weights, queue atomic, layer-0 table values, constant tables -- was in
    // flight before anything was waited for (one dependent memory round trip instead of ten): same bits, 7.3 us, launch
    // 0.2731 against 0.2736 ms same-box (profiles/r4/): the round trips were not what the prologue is made of, and this
    // simpler form stayed.
    {
        float* bw = reinterpret_cast<float*>(smem + LY::bias);
        for (int i = tid; i < (L - 1) * 256; i += 256) bw[i] = p.bias[i];
        reinterpret_cast<float*>(smem + LY::wout(L))[tid] = p.wout[tid];
        if (tid < 16) mscaleT[tid] = p.mscale[tid];
    }
    {
        const unsigned char* wb = wlayer(1);
        MSIREN_WS_LOADK(0, wb); MSIREN_WS_LOADK(1, wb); MSIREN_WS_LOADK(2, wb); MSIREN_WS_LOADK(3, wb);
        MSIREN_WS_LOADK(4, wb); MSIREN_WS_LOADK(5, wb); MSIREN_WS_LOADK(6, wb); MSIREN_WS_LOADK(7, wb);
    }
    int nxt = 0;
    if (tid == 0) nxt = (int)((unsigned)atomicAdd(p.pass_counter, 1) - p.pass_base) + (int)gridDim.x;
    __syncthreads();  // constant tables visible
    msc0 = mscaleT[q];
    msc1 = mscaleT[4 + q];
    for (int b = 0; b < nb_cur; ++b) {
        int patch, c0;
        bool live;
        unit_info(u0_cur + b, patch, c0, live);
        f32x4 m[2];
        mods_fetch(patch, m);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(m[0]), "+v"(m[1]));
        mods_store(b, row0, m);
    }
    for (int b = 0; b < nb_cur - 1; ++b) {  // the last unit's layer 0 rides in the first slot (as in every later pass)
        int patch, c0;
        bool live;
        unit_info(u0_cur + b, patch, c0, live);
        f32x4 raw[4][2];
        l0_load(c0, raw);
        l0_store(b, row0, raw);
    }
    if (tid == 0) qslot[0] = nxt;
    __syncthreads();
    {
        const int id = __builtin_amdgcn_readfirstlane(qslot[0]);
        if ((unsigned)id < npasses) pass_units(id, u0_nxt, nb_nxt);
        else nb_nxt = 0;
        cur_pass = id;  // from here on: the id of the NEXT pass
    }

    // Slot bookkeeping.  (l, b): the slot about to run.  The slot before it ("prev") is described by what its epilogue has
    // to do: pv_final -- it was a final-hidden-layer slot (dot product + output, and layer 0 of the next pass for the same
    // image); pv_unit -- its unit image; pv_row -- the modulation row of its layer.
    int l = 1, b = 0;
    // the virtual slot before the first one: "final layer of unit nb-1 of a pass before", with nothing to output
    bool pv_final = true, pv_out = false;
    int pv_unit = nb_cur - 1;
    // layer 0 to produce in a pv_final slot: unit pv_unit of the pass whose layer 1 comes next (first slot: the current pass)
    int l0_u0 = u0_cur, l0_nb = nb_cur, l0_row = row0;
    int pv_patch = 0, pv_c0 = 0;  // where the prev slot's outputs go
    bool pv_live = false;
    int pv_mrow = 0;  // modulation row (table row index, already rotated) of the prev slot's layer
    int fin_par = 0;  // which half of the reduction buffer the final slot in flight uses

    const unsigned char* wnext = wlayer(1);  // weight base of the layer to fetch during this slot (a layer's last unit)
    mark(1);

    // first fragments + bias of the first slot
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const unsigned char* img = actL;  // unit 0, k-step 0
        Bf[0][0] = lds_frag(img + 0);
        Bf[0][1] = lds_frag(img + 1024);
        Bf[0][2] = lds_frag(img + 2048);
        Bf[0][3] = lds_frag(img + 3072);
#pragma unroll
        for (int t = 0; t < 4; ++t) bia[t] = *reinterpret_cast<const f32x4*>(biasL + t * 64);
    }

    // ---- one k-step: 24 MFMAs (per accumulator the order of the register-resident kernel: W_lo x_hi, W_hi x_lo, W_hi x_hi) --
// E0..E23: what is issued behind each MFMA (the epilogue slice of the region, one statement per gap)
// E(i): what is issued behind MFMA i (the epilogue slice of the region).  FL == 2 (a layer's last unit): the fragment a
// pair of MFMAs has just retired is refilled from the next layer -- W_lo of tile t behind MFMA 2t+1, W_hi behind 17+2t.
#define MSIREN_WS_MFMAS(PAR, S, FL, E)                                                                           \
    do {                                                                                                         \
        if ((S) == 0) {                                                                                          \
            MSIREN_WS_MFMA0(PAR, 0, 0, S, 1, Bf[(S) & 1][0], bia[0]); E(S, 0, FL); \
            MSIREN_WS_MFMA0(PAR, 0, 1, S, 1, Bf[(S) & 1][2], bia[0]); E(S, 1, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 0, 1, wnext); \
            MSIREN_WS_MFMA0(PAR, 1, 0, S, 1, Bf[(S) & 1][0], bia[1]); E(S, 2, FL); \
            MSIREN_WS_MFMA0(PAR, 1, 1, S, 1, Bf[(S) & 1][2], bia[1]); E(S, 3, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 1, 1, wnext); \
            MSIREN_WS_MFMA0(PAR, 2, 0, S, 1, Bf[(S) & 1][0], bia[2]); E(S, 4, FL); \
            MSIREN_WS_MFMA0(PAR, 2, 1, S, 1, Bf[(S) & 1][2], bia[2]); E(S, 5, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 2, 1, wnext); \
            MSIREN_WS_MFMA0(PAR, 3, 0, S, 1, Bf[(S) & 1][0], bia[3]); E(S, 6, FL); \
            MSIREN_WS_MFMA0(PAR, 3, 1, S, 1, Bf[(S) & 1][2], bia[3]); E(S, 7, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 3, 1, wnext); \
        } else {                                                                                                 \
            MSIREN_WS_MFMA(PAR, 0, 0, S, 1, Bf[(S) & 1][0]); E(S, 0, FL); \
            MSIREN_WS_MFMA(PAR, 0, 1, S, 1, Bf[(S) & 1][2]); E(S, 1, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 0, 1, wnext); \
            MSIREN_WS_MFMA(PAR, 1, 0, S, 1, Bf[(S) & 1][0]); E(S, 2, FL); \
            MSIREN_WS_MFMA(PAR, 1, 1, S, 1, Bf[(S) & 1][2]); E(S, 3, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 1, 1, wnext); \
            MSIREN_WS_MFMA(PAR, 2, 0, S, 1, Bf[(S) & 1][0]); E(S, 4, FL); \
            MSIREN_WS_MFMA(PAR, 2, 1, S, 1, Bf[(S) & 1][2]); E(S, 5, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 2, 1, wnext); \
            MSIREN_WS_MFMA(PAR, 3, 0, S, 1, Bf[(S) & 1][0]); E(S, 6, FL); \
            MSIREN_WS_MFMA(PAR, 3, 1, S, 1, Bf[(S) & 1][2]); E(S, 7, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 3, 1, wnext); \
        }                                                                                                        \
        MSIREN_WS_MFMA(PAR, 0, 0, S, 0, Bf[(S) & 1][1]); E(S, 8, FL); \
        MSIREN_WS_MFMA(PAR, 0, 1, S, 0, Bf[(S) & 1][3]); E(S, 9, FL); \
        MSIREN_WS_MFMA(PAR, 1, 0, S, 0, Bf[(S) & 1][1]); E(S, 10, FL); \
        MSIREN_WS_MFMA(PAR, 1, 1, S, 0, Bf[(S) & 1][3]); E(S, 11, FL); \
        MSIREN_WS_MFMA(PAR, 2, 0, S, 0, Bf[(S) & 1][1]); E(S, 12, FL); \
        MSIREN_WS_MFMA(PAR, 2, 1, S, 0, Bf[(S) & 1][3]); E(S, 13, FL); \
        MSIREN_WS_MFMA(PAR, 3, 0, S, 0, Bf[(S) & 1][1]); E(S, 14, FL); \
        MSIREN_WS_MFMA(PAR, 3, 1, S, 0, Bf[(S) & 1][3]); E(S, 15, FL); \
        MSIREN_WS_MFMA(PAR, 0, 0, S, 0, Bf[(S) & 1][0]); E(S, 16, FL); \
        MSIREN_WS_MFMA(PAR, 0, 1, S, 0, Bf[(S) & 1][2]); E(S, 17, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 0, 0, wnext); \
        MSIREN_WS_MFMA(PAR, 1, 0, S, 0, Bf[(S) & 1][0]); E(S, 18, FL); \
        MSIREN_WS_MFMA(PAR, 1, 1, S, 0, Bf[(S) & 1][2]); E(S, 19, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 1, 0, wnext); \
        MSIREN_WS_MFMA(PAR, 2, 0, S, 0, Bf[(S) & 1][0]); E(S, 20, FL); \
        MSIREN_WS_MFMA(PAR, 2, 1, S, 0, Bf[(S) & 1][2]); E(S, 21, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 2, 0, wnext); \
        MSIREN_WS_MFMA(PAR, 3, 0, S, 0, Bf[(S) & 1][0]); E(S, 22, FL); \
        MSIREN_WS_MFMA(PAR, 3, 1, S, 0, Bf[(S) & 1][2]); E(S, 23, FL); if ((FL) == 2) MSIREN_WS_LOAD1(S, 3, 0, wnext); \
    } while (0)

    // ---- epilogue of the slot before, issued in the gaps between this slot's MFMAs ------------------------------------------
    // An MFMA holds the vector issue port for 8 of its 16 cycles, so a gap has 8 cycles for everything else: ONE
    // transcendental or ONE plain VALU instruction with room for the LDS reads / writes and waits the compiler places around
    // them (two plain VALU fill it to the brim, and every overfull gap delays the MFMA stream for good: measured +7 cycles
    // per filled gap with two).  The unit of work is half an accumulator (elements 2 HH, 2 HH + 1 of tile T, column group G):
    //   S0, S1          sine of the two elements (the accumulator IS the sine argument, in revolutions)
    //   normal layers   H0, H1: hi = f16(a m) of both; (one gap off: the partial-register write of H1 must not be read by
    //                   the next VALU) L0, L1: lo = f16(a m - hi)       -- one v_fma_mix{lo,hi}_f16 each, 7 gaps
    //   final layer     F0, F1: the dot product with last_layer.weight, one FMA each into the lane's partial sum, 4 gaps
    // All asm volatile: the order written is the order issued.  Distances (the compiler cannot pad hazards behind asm): an
    // accumulator is read >= 7 MFMAs (112 cycles) after the MFMA that completed it; a sine result is used >= 2
    // instructions later.
    unsigned ehu[2][2][4], elu[2][2][4];  // 16-byte pieces being assembled: [column group g][k-step 2 wave + u][4 x (2 x f16)]
    f32x4 em[4];          // modulation of the prev slot's layer, per tile
    f32x4 mw[4];          // final slots: modulation x last_layer.weight, per tile (the final row of the table holds the product)
    f32x4 em0_[4];        // layer-0 modulation row of the unit being produced (final slots)
    float part[2] = {0.f, 0.f};
    float sv0_ = 0.f, sv1_ = 0.f;
    [[maybe_unused]] float mt0_ = 0.f, mt1_ = 0.f;  // Morlet: cg r^2 -> exp2(cg r^2) of the two elements in flight
    f32x4 raw0[4][2];     // layer-0 table values of the unit being produced (final slots)
    unsigned l0h_[4], l0l_[4];
#define MSIREN_WS_LD_EM(T) em[T] = *reinterpret_cast<const f32x4*>(emr_ + (T) * 64)
#define MSIREN_WS_LD_MWT(T) mw[T] = *reinterpret_cast<const f32x4*>(emr_ + (T) * 64) /* final row: modulation x last_layer.weight */
#define MSIREN_WS_LD_EM0(T) em0_[T] = *reinterpret_cast<const f32x4*>(em0r_ + (T) * 64)
// sine: S0, S1 = one v_sin_f32 each.  Morlet, sin(2 pi r) * exp2(cg r^2) (the products in activate<1>'s order): both
// elements in S0 -- ten instructions straight from the accumulator registers, the two chains interleaved so that no
// transcendental's result is read by the instruction behind it -- and S1 empty.
#define MSIREN_WS_S0(T, G, HH)                                                                                       \
    do {                                                                                                             \
        if constexpr (ACT == 0) asm volatile("v_sin_f32 %0, v[%1]" : "=v"(sv0_) : "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH))); \
        else {                                                                                                       \
            float t0_, t1_;                                                                                          \
            asm volatile("v_sin_f32 %0, v[%5]\n\tv_sin_f32 %1, v[%6]\n\t"                                            \
                         "v_mul_f32 %2, %4, v[%5]\n\tv_mul_f32 %3, %4, v[%6]\n\t"                                    \
                         "v_mul_f32 %2, %2, v[%5]\n\tv_mul_f32 %3, %3, v[%6]\n\t"                                    \
                         "v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"                                                  \
                         "v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3"                                              \
                         : "=&v"(sv0_), "=&v"(sv1_), "=&v"(t0_), "=&v"(t1_)                                          \
                         : "v"(p.cg), "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH)), "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH) + 1)); \
        }                                                                                                            \
    } while (0)
#define MSIREN_WS_S1(T, G, HH)                                                                                       \
    do {                                                                                                             \
        if constexpr (ACT == 0) asm volatile("v_sin_f32 %0, v[%1]" : "=v"(sv1_) : "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH) + 1)); \
    } while (0)
// Morlet in a NORMAL slot (round 4): the same ten instructions, in activate<1>'s order per element, over seven gaps -- a pair of
// plain multiplies or ONE transcendental per gap (either is free beside an MFMA, ten in one gap are not: tools/mfma_gap_probe.hip);
// the transcendentals' results are read a gap (an MFMA) later at the earliest.  Final-layer slots keep the compact form above
// (their gaps are taken by the layer-0 work of the next pass).
#define MSIREN_WS_MO_M0(T, G, HH)                                                                                    \
    asm volatile("v_mul_f32 %0, %2, v[%3]\n\tv_mul_f32 %1, %2, v[%4]" : "=&v"(mt0_), "=&v"(mt1_)                     \
                 : "v"(p.cg), "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH)), "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH) + 1))
#define MSIREN_WS_MO_M1(T, G, HH)                                                                                    \
    asm volatile("v_mul_f32 %0, %0, v[%2]\n\tv_mul_f32 %1, %1, v[%3]" : "+v"(mt0_), "+v"(mt1_)                       \
                 : "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH)), "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH) + 1))
#define MSIREN_WS_MO_E0() asm volatile("v_exp_f32 %0, %0" : "+v"(mt0_))
#define MSIREN_WS_MO_E1() asm volatile("v_exp_f32 %0, %0" : "+v"(mt1_))
#define MSIREN_WS_MO_S0(T, G, HH) asm volatile("v_sin_f32 %0, v[%1]" : "=v"(sv0_) : "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH)))
#define MSIREN_WS_MO_S1(T, G, HH) asm volatile("v_sin_f32 %0, v[%1]" : "=v"(sv1_) : "n"(MSIREN_WS_V(PP_, T, G) + 2 * (HH) + 1))
#define MSIREN_WS_MO_P() asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3" : "+v"(sv0_), "+v"(sv1_) : "v"(mt0_), "v"(mt1_))
#define MSIREN_WS_MIXH0(DST, A, M) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(DST) : "v"(A), "v"(M))
#define MSIREN_WS_MIXH1(DST, A, M) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(DST) : "v"(A), "v"(M))
#define MSIREN_WS_MIXL0(DST, A, M, HI) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(DST) : "v"(A), "v"(M), "v"(HI))
#define MSIREN_WS_MIXL1(DST, A, M, HI) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(DST) : "v"(A), "v"(M), "v"(HI))
// both halves of a register in one statement (two plain VALU in one gap; fewer statement boundaries for the compiler to pad)
#define MSIREN_WS_MIXH01(DST, A0, M0, A1, M1)                                                                        \
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]"  \
                 : "=&v"(DST) : "v"(A0), "v"(M0), "v"(A1), "v"(M1))
#define MSIREN_WS_MIXL01(DST, A0, M0, A1, M1, HI)                                                                    \
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                              \
                 "v_fma_mixhi_f16 %0, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"                                  \
                 : "=&v"(DST) : "v"(A0), "v"(M0), "v"(A1), "v"(M1), "v"(HI))
#define MSIREN_WS_MIXL01_LAST(DST, A0, M0, A1, M1, HI)                                                               \
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                              \
                 "v_fma_mixhi_f16 %0, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 0"                        \
                 : "=&v"(DST) : "v"(A0), "v"(M0), "v"(A1), "v"(M1), "v"(HI))
// the last half-register write of a 16-byte piece: the ds_write that takes the piece may be scheduled right behind it
#define MSIREN_WS_MIXL1_LAST(DST, A, M, HI) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 0" : "+v"(DST) : "v"(A), "v"(M), "v"(HI))
    // a finished 16-byte piece pair (hi, lo) -> the unit image, k-step 2 wave + U, column group G
#define MSIREN_WS_STORE_PIECE(HI, LO, U, G)                                                                          \
    do {                                                                                                             \
        u32x4 hh_, ll_;                                                                                              \
        hh_[0] = (HI)[0]; hh_[1] = (HI)[1]; hh_[2] = (HI)[2]; hh_[3] = (HI)[3];                                      \
        ll_[0] = (LO)[0]; ll_[1] = (LO)[1]; ll_[2] = (LO)[2]; ll_[3] = (LO)[3];                                      \
        *reinterpret_cast<u32x4*>(pimg_ + (U) * 4096 + (2 * (G)) * 1024) = hh_;                                      \
        *reinterpret_cast<u32x4*>(pimg_ + (U) * 4096 + (2 * (G) + 1) * 1024) = ll_;                                  \
    } while (0)
    // final slots: sum of the lane's partial over the four feature sub-groups q (two lane swaps: rows 16 apart, halves 32
    // apart -- the order of the register-resident kernel's __shfl_xor 16, 32), then one float per coordinate and wave
    // layer-0 table values of half U (4 loads, issued at the start of region 0 / 2): landed once at most KL (a loading
    // slot: the weight loads issued since) or KP (otherwise) younger operations are outstanding
#define MSIREN_WS_RAW0_WAIT(FL, U, KL, KP)                                                                           \
    do {                                                                                                             \
        if ((FL) == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(raw0[2 * (U)][0]), "+v"(raw0[2 * (U)][1]), "+v"(raw0[2 * (U) + 1][0]), "+v"(raw0[2 * (U) + 1][1]) : "n"(KL)); \
        else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(raw0[2 * (U)][0]), "+v"(raw0[2 * (U)][1]), "+v"(raw0[2 * (U) + 1][0]), "+v"(raw0[2 * (U) + 1][1]) : "n"(KP)); \
    } while (0)
    // region 7 of a final slot, behind the barrier: the prev (final-layer) slot's partial sums are all in LDS -> sum over
    // the waves, last_layer's sine, store.  Every lane computes, lanes that have nothing to store write to the dump buffer.
#define MSIREN_WS_FIN0() fin_r_ = *reinterpret_cast<const f32x4*>(redW + (fin_par * 32 + (lane & 31)) * 4)
#define MSIREN_WS_FIN1() fin_s_ = ((fin_r_[0] + fin_r_[1]) + (fin_r_[2] + fin_r_[3])) + p.bout
#define MSIREN_WS_FIN2() fin_s_ = sin_rev(fin_s_)
#define MSIREN_WS_FIN3() *outp_ = fin_s_
#define MSIREN_WS_RED0() do { pa_[0] = part[0]; pb_[0] = part[0]; pa_[1] = part[1]; pb_[1] = part[1]; } while (0)
#define MSIREN_WS_RED1() asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(pa_[0]), "+v"(pb_[0]), "+v"(pa_[1]), "+v"(pb_[1]))
#define MSIREN_WS_RED2() do { pa_[0] += pb_[0]; pa_[1] += pb_[1]; pb_[0] = pa_[0]; pb_[1] = pa_[1]; } while (0)
#define MSIREN_WS_RED3() asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(pa_[0]), "+v"(pb_[0]), "+v"(pa_[1]), "+v"(pb_[1]))
#define MSIREN_WS_RED4()                                                                                             \
    do {                                                                                                             \
        if (q < 2) redW[(fin_par * 32 + q * 16 + n16) * 4 + wave] = q == 0 ? pa_[0] + pb_[0] : pa_[1] + pb_[1];      \
        part[0] = 0.f; part[1] = 0.f;                                                                                \
    } while (0)
// Ablation builds (timing only, results wrong; never shipped): -DMSIREN_WS_ABL=bitmask (8: no MFMAs, see MSIREN_WS_MFMA)
//   1 = no epilogue in the gaps, 2 = no B-fragment LDS reads, 4 = no barrier
// Every VGPR-destination load this kernel issues through asm (layer-0 table, modulation rows, the queue atomic) is followed
// by a counted s_waitcnt that carries the destination as an operand BEFORE anything else may touch those registers --
// tests/test_asm_hazards.py checks that in the ISA.  Round 3's -DMSIREN_WS_ABL=15 build broke exactly this and faulted
// (gpurun_out/r3/abl/abl15.err): bit 1 removed the gaps, and with them the MSIREN_WS_RAW0_WAITs and the final body's
// output store that the boundary waits count on, but NOT the table loads in MSIREN_WS_PRE_B.  Their destinations were dead
// on arrival, so the allocator handed the registers on (ISA of that build: `global_load_dwordx4 v[20:23]` ... no wait ...
// `v_add_u32 v20, 0x10400, v0`; `ds_read_b128 v[20:23], v20`; in the other body variant v32/v33 -- destinations here --
// are the OFFSET registers of the next table loads).  With MFMAs (ABL = 7) a slot lasts >= 3000 cycles and the data landed
// before the registers were reused; without them (bit 8) a slot is shorter than a memory round trip, the late data
// overwrote live offsets, and the next `global_load_dwordx4 ..., v32, s[..]` went wherever that pointed.  The shipped
// build never had the hazard (0 of 89 such loads; the ablations: 45).  Since round 4 bit 1 also drops the loads whose
// waits it drops, and the boundary waits that counted on the removed store drain instead.
#ifndef MSIREN_WS_ABL
#define MSIREN_WS_ABL 0
#endif
#if MSIREN_WS_ABL & 1
#define MSIREN_WS_GAP_A(S, I, FL) do {} while (0)
#define MSIREN_WS_GAP_B(S, I, FL) do {} while (0)
#else
#define MSIREN_WS_GAP_A(S, I, FL) MSIREN_WS_GA_##S##_##I(FL)
#define MSIREN_WS_GAP_B(S, I, FL) MSIREN_WS_GB_##S##_##I(FL)
#endif
    float pa_[2] = {0.f, 0.f}, pb_[2] = {0.f, 0.f};
    f32x4 fin_r_ = {0.f, 0.f, 0.f, 0.f};
    float fin_s_ = 0.f;

    // Table reads, a region ahead of their first use (two tiles of a row at a time).  Region 7 of EVERY slot reads what the
    // next slot's first region needs whichever variant it is: tiles 0, 1 of the modulation row of this slot's layer and of
    // last_layer.weight (nemr_: the row the next slot's epilogue works on, i.e. this slot's).
#define MSIREN_WS_PRE_A(S)                                                                                           \
    do {                                                                                                             \
        if ((S) == 0) { MSIREN_WS_LD_EM(0); MSIREN_WS_LD_EM(1); } /* used from gap 10 on */                          \
        if ((S) == 2) { MSIREN_WS_LD_EM(2); MSIREN_WS_LD_EM(3); } /* used from gap 68 on */                          \
    } while (0)
#define MSIREN_WS_PRE_B(S)                                                                                           \
    do {                                                                                                             \
        if ((S) == 0) { MSIREN_WS_LD_MWT(0); MSIREN_WS_LD_MWT(1); MSIREN_WS_LD_MWT(2); MSIREN_WS_LD_MWT(3);          \
                        if (!(MSIREN_WS_ABL & 1)) l0_load_half(l0c0_, 0, raw0); } /* (no load without its wait) */   \
        if ((S) == 2) { MSIREN_WS_LD_EM0(0); MSIREN_WS_LD_EM0(1); if (!(MSIREN_WS_ABL & 1)) l0_load_half(l0c0_, 1, raw0); } \
        if ((S) == 3) { MSIREN_WS_LD_EM0(2); MSIREN_WS_LD_EM0(3); }                                                  \
    } while (0)

    // One k-step region of a slot.  VAR: 0 = normal epilogue, 1 = final-layer epilogue + layer 0.  FL: 0 = a layer's first
    // unit (waits for the fragments fetched during the slot before), 1 = middle, 2 = last (fetches the next layer's).
// B fragment J (hi g0, lo g0, hi g1, lo g1) of the NEXT k-step (S == 7: of the next slot's first), one per gap of the
// generated schedule, fenced so that the compiler leaves it in its gap (it would bunch the four at the region's start, or
// sink them to their first use).  Same bits, same-box A/B (profiles/r3/13_ab_spread_reads.txt): 0.2723-0.2753 ms against
// 0.2717-0.2722 ms with the four in a bunch at the region's start -- no gain, so the bunch stays the default
// (-DMSIREN_WS_SPREAD_READS=1 builds the spread form).
#ifndef MSIREN_WS_SPREAD_READS
#define MSIREN_WS_SPREAD_READS 0
#endif
#define MSIREN_WS_BREAD(S, J)                                                                                        \
    do {                                                                                                             \
        if (MSIREN_WS_SPREAD_READS && (!(MSIREN_WS_ABL & 2) || (S) == 7)) {                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            Bf[((S) + 1) & 1][J] = lds_frag(((S) < 7 ? cimg_ + ((S) + 1) * 4096 : nimg_) + (J) * 1024);              \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    } while (0)

#define MSIREN_WS_REGION(PAR, VAR, FL, S)                                                                            \
    do {                                                                                                             \
        [[maybe_unused]] constexpr int PP_ = (PAR) ^ 1; /* (the accumulator set the epilogue in the gaps reads) */     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        MSIREN_WS_HOLD();                                                                                            \
        if ((FL) == 0) MSIREN_WS_WAITK(S);                                                                           \
        if ((S) == 7 && !(MSIREN_WS_ABL & 4)) { /* the slot's barrier: own LDS stores done, every wave past its last read of the unit image */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
            __builtin_amdgcn_s_barrier();                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        if (!MSIREN_WS_SPREAD_READS && (!(MSIREN_WS_ABL & 2) || (S) == 7)) {   /* B fragments of the next k-step, in a bunch (A/B build) */ \
            const unsigned char* src_ = (S) < 7 ? cimg_ + ((S) + 1) * 4096 : nimg_;                                  \
            Bf[((S) + 1) & 1][0] = lds_frag(src_ + 0);                                                               \
            Bf[((S) + 1) & 1][1] = lds_frag(src_ + 1024);                                                            \
            Bf[((S) + 1) & 1][2] = lds_frag(src_ + 2048);                                                            \
            Bf[((S) + 1) & 1][3] = lds_frag(src_ + 3072);                                                            \
        }                                                                                                            \
        if ((VAR) == 0) MSIREN_WS_PRE_A(S); else MSIREN_WS_PRE_B(S);                                                 \
        if ((S) == 7) {                                                                                              \
            bia[0] = *reinterpret_cast<const f32x4*>(nbias_ + 0);                                                    \
            bia[1] = *reinterpret_cast<const f32x4*>(nbias_ + 64);                                                   \
            bia[2] = *reinterpret_cast<const f32x4*>(nbias_ + 128);                                                  \
            bia[3] = *reinterpret_cast<const f32x4*>(nbias_ + 192);                                                  \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0); /* the reads are issued here, ahead of the MFMAs, not wherever they fit */ \
        if ((VAR) == 0) MSIREN_WS_MFMAS(PAR, S, FL, MSIREN_WS_GAP_A); else MSIREN_WS_MFMAS(PAR, S, FL, MSIREN_WS_GAP_B); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)

#define MSIREN_WS_SLOT(PAR, VAR, FL)                                                                                 \
    do {                                                                                                             \
        MSIREN_WS_REGION(PAR, VAR, FL, 0);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 1);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 2);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 3);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 4);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 5);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 6);                                                                           \
        MSIREN_WS_REGION(PAR, VAR, FL, 7);                                                                           \
    } while (0)

    int slot_par = 0;
    int k_in_pass = 2;          // slots of the current pass executed so far (drives the pass-id pipeline below; the first
                                // pass got its successor's id in the prologue)
    int fetched_id = 0;         // tid 0: the id the queue returned for the pass after next (in flight for one slot)
    f32x4 mnext[2];             // modulation rows of the next pass's unit being staged (fetched before a slot, stored behind it)
    mnext[0] = mnext[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    int dbg_slot = 0;
    (void)dbg_slot;
    auto stamp = [&](int which) {  // diagnostic instance only: [workgroup][slot < 96][8] s_memtime (3: s_memrealtime; 4..6: inside the slot boundary)
        if constexpr (DBG) {
            const unsigned long long t = which == 3 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
            if (tid == 0 && dbg_slot < 96) p.stamps[((size_t)blockIdx.x * 96 + dbg_slot) * 8 + which] = t;
        }
    };

    // One slot of flavour FL (0 = a layer's first unit, 1 = middle, 2 = last: a compile-time property of where the slot sits
    // in the loops below) at (l, b); (NL, NBQ) = the slot after it.
#define MSIREN_WS_RUN(FL, NL, NBQ)                                                                                   \
    do {                                                                                                             \
        stamp(0);                                                                                                    \
        const unsigned char* const cimg_ = actL + b * 32768;                                                         \
        const unsigned char* const nimg_ = actL + (NBQ) * 32768;                                                     \
        const unsigned char* const nbias_ = biasL + ((NL) - 1) * 1024;                                               \
        unsigned char* const pimg_ = actL + pv_unit * 32768 + (2 * wave) * 4096;                                     \
        /* tables of the prev slot's epilogue */                                                                     \
        const unsigned char* const emr_ = modsL + (pv_unit * NR + pv_mrow) * 1024;                                   \
        const unsigned char* const em0r_ = modsL + (pv_unit * NR + l0_row) * 1024;                                   \
        int l0c0_ = 0;                                                                                               \
        float* outp_ = p.dump + tid; /* where this lane's output of the prev final slot goes (region 7 of a final body) */ \
        if (pv_final) {                                                                                              \
            int patch0;                                                                                              \
            bool live0;                                                                                              \
            unit_info(l0_u0 + (pv_unit < l0_nb ? pv_unit : 0), patch0, l0c0_, live0);                                \
            if (wave == 0 && lane < 32 && pv_out && pv_live && pv_c0 + lane < P) outp_ = p.out + (size_t)pv_patch * P + pv_c0 + lane; \
        }                                                                                                            \
        /* modulation rows of the next pass's unit b: fetched before the final layer's slot b, stored behind it */   \
        const bool stage_mods = l == L - 1 && b < nb_nxt;                                                            \
        if (stage_mods) {                                                                                            \
            int patch, c0;                                                                                           \
            bool live;                                                                                               \
            unit_info(u0_nxt + b, patch, c0, live);                                                                  \
            mods_fetch(patch, mnext);                                                                                \
        }                                                                                                            \
        stamp(1);                                                                                                    \
        switch (slot_par * 2 + (pv_final ? 1 : 0)) {                                                                 \
            case 0: MSIREN_WS_SLOT(0, 0, FL); break;                                                                 \
            case 1: MSIREN_WS_SLOT(0, 1, FL); break;                                                                 \
            case 2: MSIREN_WS_SLOT(1, 0, FL); break;                                                                 \
            default: MSIREN_WS_SLOT(1, 1, FL); break;                                                                \
        }                                                                                                            \
        stamp(2);                                                                                                    \
        stamp(3);                                                                                                    \
        /* ---- slot boundary ---- */                                                                                \
        /* (the prev final slot's output was finished in this slot's region 7) */                                   \
        if (pv_final && pv_out) fin_par ^= 1;                                                                        \
        if (stage_mods) { /* (fetched a slot ago; behind it at most this slot's 64 weight loads, 8 table loads and, youngest, */ \
            /* a final body's output store -- which must NOT be waited for: a store takes ~1000 cycles to be acknowledged) */ \
            if ((FL) == 2) asm volatile("s_waitcnt vmcnt(63)" : "+v"(mnext[0]), "+v"(mnext[1]));                     \
            else if (pv_final && !(MSIREN_WS_ABL & 1)) asm volatile("s_waitcnt vmcnt(1)" : "+v"(mnext[0]), "+v"(mnext[1])); \
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(mnext[0]), "+v"(mnext[1]));                                \
            int r0n = row0 + L;                                                                                      \
            r0n = r0n >= NR ? r0n - NR : r0n;                                                                        \
            mods_store(b, r0n, mnext);                                                                               \
        }                                                                                                            \
        stamp(4);                                                                                                    \
        /* pass-id pipeline: the atomic is issued at the pass boundary, its result written to LDS one slot later (no */ \
        /* wait on the way), read by everybody another slot later (a barrier in between); needed from the final layer on */ \
        if (k_in_pass == 0) { /* the atomic was issued a whole slot ago; the body since was a final one (a pass's first */ \
            /* slot follows a final-layer slot): its output store is the one younger operation */                   \
            if (MSIREN_WS_ABL & 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fetched_id)); /* (no younger store / table loads there) */ \
            else asm volatile("s_waitcnt vmcnt(1)" : "+v"(fetched_id));                                              \
            if (tid == 0) qslot[0] = (int)((unsigned)fetched_id - p.pass_base) + (int)gridDim.x;                     \
        }                                                                                                            \
        if (k_in_pass == 1) {                                                                                        \
            const int id = __builtin_amdgcn_readfirstlane(qslot[0]);                                                 \
            if ((unsigned)id < npasses) pass_units(id, u0_nxt, nb_nxt);                                              \
            else nb_nxt = 0;                                                                                         \
        }                                                                                                            \
        stamp(5);                                                                                                    \
        ++k_in_pass;                                                                                                 \
        /* this slot becomes the prev one */                                                                         \
        pv_final = l == L - 1;                                                                                       \
        pv_out = pv_final;                                                                                           \
        pv_unit = b;                                                                                                 \
        {                                                                                                            \
            const int r = row0 + l;                                                                                  \
            pv_mrow = r >= NR ? r - NR : r;                                                                          \
        }                                                                                                            \
        if (pv_final) {                                                                                              \
            unit_info(u0_cur + b, pv_patch, pv_c0, pv_live);                                                         \
            l0_u0 = u0_nxt; /* the layer 0 it produces beside its dot product: unit b of the next pass */            \
            l0_nb = nb_nxt;                                                                                          \
            const int r0n = row0 + L;                                                                                \
            l0_row = r0n >= NR ? r0n - NR : r0n;                                                                     \
        }                                                                                                            \
        slot_par ^= 1;                                                                                               \
        stamp(6);                                                                                                    \
        if constexpr (DBG) ++dbg_slot;                                                                               \
    } while (0)

    for (;;) {  // passes
        for (l = 1; l < L; ++l) {
            const bool lastl = l == L - 1;
            // what a layer's last unit fetches: the next layer's weights, or layer 1's for the next pass (also when no pass
            // follows: harmless)
            wnext = wlayer(lastl ? 1 : l + 1);
            b = 0;
            MSIREN_WS_RUN(0, l, 1);  // (passes have >= 2 units)
            for (b = 1; b < nb_cur - 1; ++b) MSIREN_WS_RUN(1, l, b + 1);
            MSIREN_WS_RUN(2, lastl ? 1 : l + 1, 0);
        }
        if (nb_nxt <= 0) break;
        // the next pass becomes the current one; the queue is asked for the one after it
        u0_cur = u0_nxt;
        nb_cur = nb_nxt;
        const int r0n = row0 + L;
        row0 = r0n >= NR ? r0n - NR : r0n;
        nb_nxt = -1;  // unknown until the pipeline above delivers it (two slots from now)
        k_in_pass = 0;
        // (asm: the compiler's atomicAdd waits for the returned value on the spot -- vmcnt(0), ~2000 cycles at every pass end)
        if (tid == 0)
            asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=&v"(fetched_id) : "v"(0u), "v"(1), "s"(p.pass_counter) : "memory");
    }
#undef MSIREN_WS_RUN

    // ---- drain: the last slot's epilogue (final layer, nothing beside it) and its output -----------------------------------
    __builtin_amdgcn_sched_barrier(0);
    {
        const int pp = slot_par ^ 1;  // parity of the slot that has just run
        const unsigned char* const emr_ = modsL + (pv_unit * NR + pv_mrow) * 1024;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs (asm) are still writing the accumulators read below
#pragma unroll
        for (int t = 0; t < 4; ++t) MSIREN_WS_LD_MWT(t);
        float pr[2] = {0.f, 0.f};
        // the last slot's accumulators, by name (parity pp is a run-time value here: both candidates are read, one is kept)
#define MSIREN_WS_DRAIN1(T, G, E)                                                                                      \
        do {                                                                                                           \
            float a0_, a1_;                                                                                            \
            asm volatile("v_mov_b32 %0, v[%1]" : "=v"(a0_) : "n"(MSIREN_WS_V(0, T, G) + (E)));                           \
            asm volatile("v_mov_b32 %0, v[%1]" : "=v"(a1_) : "n"(MSIREN_WS_V(1, T, G) + (E)));                           \
            pr[G] = __builtin_fmaf(activate<ACT>(pp ? a1_ : a0_, p.cg), mw[T][E], pr[G]);                              \
        } while (0)
#define MSIREN_WS_DRAIN4(T, G) MSIREN_WS_DRAIN1(T, G, 0); MSIREN_WS_DRAIN1(T, G, 1); MSIREN_WS_DRAIN1(T, G, 2); MSIREN_WS_DRAIN1(T, G, 3)
        MSIREN_WS_DRAIN4(0, 0); MSIREN_WS_DRAIN4(0, 1); MSIREN_WS_DRAIN4(1, 0); MSIREN_WS_DRAIN4(1, 1);
        MSIREN_WS_DRAIN4(2, 0); MSIREN_WS_DRAIN4(2, 1); MSIREN_WS_DRAIN4(3, 0); MSIREN_WS_DRAIN4(3, 1);
#undef MSIREN_WS_DRAIN4
#undef MSIREN_WS_DRAIN1
        float s0 = pr[0], s1 = pr[1];
        s0 += __shfl_xor(s0, 16);
        s1 += __shfl_xor(s1, 16);
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (q < 2) redW[(fin_par * 32 + q * 16 + n16) * 4 + wave] = q == 0 ? s0 : s1;
        __syncthreads();
        if (wave == 0 && lane < 32) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(redW + (fin_par * 32 + lane) * 4);
            const float sum = (r[0] + r[1]) + (r[2] + r[3]);
            const int pc = pv_c0 + lane;
            if (pv_live && pc < P) p.out[(size_t)pv_patch * P + pc] = sin_rev(sum + p.bout);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mark(2);
    MSIREN_WS_HOLD();
#undef MSIREN_WS_HOLD
#undef MSIREN_WS_SLOT
#undef MSIREN_WS_REGION
#undef MSIREN_WS_PRE_A
#undef MSIREN_WS_PRE_B
#undef MSIREN_WS_MFMAS
#undef MSIREN_WS_MFMA
#undef MSIREN_WS_MFMA0
#undef MSIREN_WS_WAITK
#undef MSIREN_WS_LOADK
#undef MSIREN_WS_LOAD1
}

}  // namespace msiren

// Weight RING of the 16-point field kernels (forward: field_fwd_ring.hip, dgrad: field_bwd_ring.hip).
//
// What it replaces: the double-buffered 2 x 64 KiB weight stream of field_device.h (WeightStreamT).  With two buffers
// the workgroup barrier has to sit ON the chunk boundary, so every 64 KiB all eight waves (i) wait, (ii) issue their 8 KiB
// of L2 -> LDS DMA back to back, (iii) request the first A fragments of the new chunk and sit out the LDS latency --
// both waves of every SIMD at the same moment, with the matrix pipe empty (round-2 profile: MFMA-busy 0.51-0.60, one
// third of the wave cycles in s_waitcnt / s_barrier).  The ring removes the three bubbles without touching the arithmetic:
//
//   * The stream is a sequence of 8 KiB UNITS (4 output blocks x (hi, lo) x 64 lanes x 16 B = the A fragments of one group
//     of 12 MFMAs per wave); the LDS ring holds 17 of them (136 KiB) and unit u lives in slot u mod 17.
//   * A chunk is 8 units (64 KiB, one unit per wave to fetch).  The ONE barrier per chunk is executed at unit 7 of the
//     chunk (not at its end): before it every wave has waited for its own DMA of the NEXT chunk, after it that chunk is
//     complete for everybody and the slots of units [8j-1, 8j+6] are free, which is exactly where chunk j+2 lands.  So the
//     transition from one chunk to the next has no barrier, no wait and no DMA burst in it.
//   * Fragment requests run 8 MFMAs ahead of their use (unit_pipelined), also across chunk and layer boundaries: three
//     fragment register sets of 16 (two hi sets alternating between units, one lo set).
//   * The DMA of a chunk (8 KiB per wave) is issued in four parts of 2 KiB BEHIND the MFMAs of the four units that
//     follow the barrier instead of as one burst in front of them: a DMA piece costs the issuing wave 60-180 clk, which
//     the other wave of the SIMD covers with its own MFMAs as long as the parts are short.
#pragma once
#include <type_traits>
#include "split_types.h"

namespace nerf {

constexpr int RING_UNITS = 17;
constexpr int UNIT_WORDS = 4 * 2 * 64 * 4;                  // 2048 words = 8 KiB
constexpr int RING_FLOATS = RING_UNITS * UNIT_WORDS;        // 136 KiB
constexpr int RING_LDS_FLOATS = RING_FLOATS + SMALL_FLOATS;
constexpr int CHUNK_UNITS = 8;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct Frag { u32x4 w[4]; };         // the hi (or lo) A fragments of a unit's four output blocks

__device__ inline void stage_small_ring(const float* small_src, float* lds, int nthreads) {
    const f32x4* src = reinterpret_cast<const f32x4*>(small_src);
    f32x4* dst = reinterpret_cast<f32x4*>(lds + RING_FLOATS);
    for (int i = threadIdx.x; i < SMALL_FLOATS / 4; i += nthreads) dst[i] = src[i];
}
__device__ inline const float* ring_small_ptr(const float* lds, int sm_offset) { return lds + RING_FLOATS + (sm_offset - SM_BIAS); }

// Stream description: unit u of the consumed sequence lives at word offset (u < skip_at ? u : u + skip_units) * UNIT_WORDS
// of `src` (the folded feature layer's units stay in the packed stream and are jumped over); n_units in total.
// NWAVES waves share the ring; each fetches 8 / NWAVES consecutive units of every chunk (8 waves: the 16-point kernels, one
// unit each; 4 waves: the 32-point kernels, two units each).
template <int NWAVES>
struct WeightRingT {
    static constexpr int UPW = CHUNK_UNITS / NWAVES;    // units per wave and chunk
    const float* src;
    const u32x4* lane_ptr;      // LDS: ring base + lane (u32x4 units)
    unsigned ring_base;         // LDS byte address of the ring
    int wave, lane;
    int skip_at, skip_units, n_units;
    int slot;                   // ring slot of the next unit to REQUEST (wave-uniform)
    int dma_chunk;              // next chunk to fetch
    int dma_slot;               // ring slot of the first unit of this wave's span of chunk dma_chunk
    int dma_live0, dma_live1;   // this wave's first / second unit of chunk dma_chunk exists (scalar: only the last chunk is ragged)
    const float* dma_g;         // per-lane global address of the next part to fetch
    unsigned dma_l, dma_l1;     // LDS byte address of the next part / of the span's second unit

    // part K (0..3) of this wave's span of the chunk being fetched: 2 KiB of its unit (8 waves) or 4 KiB = half a unit (4 waves).
    // The global and LDS addresses of the next part are carried (dma_g, dma_l: one pointer add and one scalar add per part)
    // instead of being rebuilt from the chunk and unit numbers each time.
    template <int K>
    __device__ __forceinline__ void fetch_part(int /*c*/) {
        static_assert(UPW == 1 || UPW == 2, "8 or 4 waves");
        constexpr int part_words = UPW == 2 ? 1024 : 512;
        if (UPW == 2 && K == 2) dma_l = dma_l1;                            // second unit of the span: its own ring slot (may wrap)
        if ((UPW == 2 && K / 2 == 1) ? dma_live1 : dma_live0) {
            if (UPW == 2) dma_4k(dma_g, dma_l);
            else dma_2k(dma_g, dma_l);
        }
        dma_g += part_words;
        dma_l += part_words * 4u;
    }
    __device__ __forceinline__ void fetch_unit(int c) { fetch_part<0>(c); fetch_part<1>(c); fetch_part<2>(c); fetch_part<3>(c); }
    // state of the chunk dma_chunk: which of this wave's units exist, where they come from and where they land
    __device__ __forceinline__ void set_chunk() {
        const int u0 = CHUNK_UNITS * dma_chunk + wave * UPW;
        dma_live0 = __builtin_amdgcn_readfirstlane(u0 < n_units ? 1 : 0);
        dma_live1 = __builtin_amdgcn_readfirstlane(u0 + 1 < n_units ? 1 : 0);
        dma_g = src + (size_t)(u0 < skip_at ? u0 : u0 + skip_units) * UNIT_WORDS + lane * 4;      // (a span never straddles the skip)
        dma_l = ring_base + (unsigned)dma_slot * (UNIT_WORDS * 4u);
        const int s1 = dma_slot + 1 == RING_UNITS ? 0 : dma_slot + 1;
        dma_l1 = ring_base + (unsigned)s1 * (UNIT_WORDS * 4u);
    }
    __device__ __forceinline__ void advance_dma() {
        ++dma_chunk;
        dma_slot += CHUNK_UNITS;
        if (dma_slot >= RING_UNITS) dma_slot -= RING_UNITS;
        set_chunk();
    }
    // chunks 0 and 1 (units 0..15 -> slots 0..15); finish with ready()
    __device__ inline void start(const float* src_, float* lds, int wave_, int lane_, int skip_at_, int skip_units_, int n_units_) {
        src = src_; wave = wave_; lane = lane_; skip_at = skip_at_; skip_units = skip_units_; n_units = n_units_;
        ring_base = __builtin_amdgcn_readfirstlane(lds_addr(lds));
        lane_ptr = reinterpret_cast<const u32x4*>(lds) + lane;
        slot = 0;
        dma_chunk = 0; dma_slot = wave * UPW;
        set_chunk();
        fetch_unit(0); advance_dma();
        fetch_unit(1); advance_dma();           // dma_slot = (16 + wave * UPW) mod 17
    }
    __device__ __forceinline__ void ready() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // LDS addresses (u32x4 units, lane included) of the unit in progress and of the next one; advances the ring position
    __device__ __forceinline__ void unit_ptrs(const u32x4*& p, const u32x4*& pn) {
        p = lane_ptr + slot * (UNIT_WORDS / 4);
        slot = slot + 1 == RING_UNITS ? 0 : slot + 1;
        pn = lane_ptr + slot * (UNIT_WORDS / 4);
    }
    // the hi fragments of the very first unit (slot stays on that unit)
    __device__ __forceinline__ void request_first(Frag& hi) {
        const u32x4* p = lane_ptr + slot * (UNIT_WORDS / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) hi.w[i] = p[(2 * i) * 64];
    }
    // the barrier of a chunk (see the header comment): call it at unit 7 of the chunk, BEFORE requesting unit 8.
    // NPEND: vector-memory instructions (row stores) this wave is guaranteed to have issued after the last part of its
    // fetch; they keep draining (vmcnt retires in order).  0 = drain everything.
    template <int NPEND>
    __device__ __forceinline__ void barrier() {
        if (NPEND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPEND) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // behind the MFMAs of the unit at position POS of its chunk: part (POS + 1) mod 8 of the chunk after next, if < 4
    template <int POS>
    __device__ __forceinline__ void fetch_after_unit() {
        constexpr int K = (POS + 1) % CHUNK_UNITS;
        if constexpr (K < 4) fetch_part<K>(dma_chunk);
        if constexpr (K == 3) advance_dma();
    }
};
using WeightRing = WeightRingT<8>;

// 8 fp32 values -> (hi, lo) B fragments of the split SP (split_types.h), pair by pair
template <typename SP>
__device__ __forceinline__ void split8_pk(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        SP::split_pair(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h;
        lo[i] = l;
    }
}
constexpr int lgkmcnt_only(int n) { return 0xC07F | (n << 8); }     // s_waitcnt immediate: lgkmcnt(n), vmcnt / expcnt not waited for

// ONE UNIT, instruction by instruction (sched_barrier(0) pins the order; hipcc's own schedule of the same work is
// [8 requests][operand split][12 MFMAs], and because the two waves of a SIMD run the same stream in lockstep, a phase
// without MFMAs is a phase in which that SIMD's matrix pipe idles -- measured on the phased kernel: removing the
// requests / the split / the DMA gained their full issue time, the second wave covered none of it).  Every MFMA is
// followed by at most one LDS request or a few VALU operations, which issue in its 16-cycle shadow:
//     hi[i] * bhi   + request lo[i] of THIS unit        (used 8 MFMAs later)   [16x16x32 or 32x32x16 MFMAs of the split SP]
//     hi[i] * blo   + request hi[i] of the NEXT unit    (used 8 MFMAs later, in the next unit)
//     lo[i] * bhi   + `tail(i)`: a quarter of the next k-step's operand split, the unit's row store
// The order per accumulator (hi*hi, hi*lo, lo*hi) is the order of mma16_group: results stay bit-identical.
template <typename SP, int NB, typename Acc, typename Tail>
__device__ __forceinline__ void unit_pipelined(Acc (&acc)[NB], int g4, const Frag& cur, Frag& lo, Frag& nxt, const u32x4* p, const u32x4* pn,
                                               const u32x4 bhi, const u32x4 blo, Tail tail) {
    // ONE wait for the four hi fragments (requested 8 MFMAs ago; hipcc otherwise waits in front of each MFMA: 8 s_waitcnt per
    // unit, each an issue slot) ...
    __builtin_amdgcn_s_waitcnt(lgkmcnt_only(0));
    static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        acc[g4 + i] = SP::mfma(cur.w[i], bhi, acc[g4 + i]);
        lo.w[i] = p[(2 * i + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        acc[g4 + i] = SP::mfma(cur.w[i], blo, acc[g4 + i]);
        nxt.w[i] = pn[(2 * i) * 64];
        __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_waitcnt(lgkmcnt_only(4));        // ... and one for the four lo fragments (the next unit's hi requests stay in flight)
    static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        acc[g4 + i] = SP::mfma(lo.w[i], bhi, acc[g4 + i]);
        tail(ic);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// NU units of one contraction: G groups (units) per k-step, B operand of k-step s = v[VOFF + 8 s ..+7].  The first
// unit is at position 0 of its chunk (every contraction starts on a chunk boundary); FIRST = the very first units of
// the kernel (no fetch in progress at positions 0..2).  `after(k-step, group, bhi, blo)` runs behind the 10th MFMA of each
// unit (row stores; bhi = the hi words of the k-step in progress, which ARE the 16-bit values of its rows; blo = their remainders,
// stored as well by the two-word save forms).  NPEND: see WeightRing::barrier (the row stores of positions 3..6 follow the last part of a fetch: 4 when
// one per unit).  fa holds the hi fragments of the first unit on entry and of the unit after the last one on exit (NU is
// even); fb is the second hi set, fl the lo set of the unit in progress.
template <typename SP, int NU, int G, int VOFF, bool FIRST, int NPEND, int NW, typename Acc, int NB, int NV, typename After>
__device__ __forceinline__ void ring_units(WeightRingT<NW>& ring, Frag& fa, Frag& fb, Frag& fl, Acc (&acc)[NB], const float (&v)[NV], After after) {
    static_assert(NU % 2 == 0, "fragment sets alternate");
    static_assert(G == 4 || G == 2, "units per k-step");
    // B operand (hi, lo) of the k-step in progress and of the next one: the split of k-step s+1 is spread over the units
    // of k-step s (4 / G word pairs each), in the shadow of their last MFMAs
    // (two operand sets, alternating by k-step parity: no register copies when a k-step begins)
    u32x4 Bh[2], Bl[2];
    split8_pk<SP>(&v[VOFF], Bh[0], Bl[0]);
    static_for<0, NU>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / G, g = i % G, pos = i % CHUNK_UNITS;
        constexpr bool more = (s + 1) * G < NU;             // another k-step follows
        Frag& cur = (i & 1) ? fb : fa;
        Frag& nxt = (i & 1) ? fa : fb;
        if constexpr (pos == CHUNK_UNITS - 1) ring.template barrier<NPEND>();
        u32x4& bhi = Bh[s & 1];
        u32x4& blo = Bl[s & 1];
        u32x4& nhi = Bh[(s + 1) & 1];
        u32x4& nlo = Bl[(s + 1) & 1];
        const u32x4 *p, *pn;
        ring.unit_ptrs(p, pn);
        __builtin_amdgcn_sched_barrier(0);
        unit_pipelined<SP>(acc, 4 * g, cur, fl, nxt, p, pn, bhi, blo, [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            if constexpr (more && ((G == 4 && t == 0) || (G == 2 && t < 2))) {
                constexpr int j = G == 4 ? g : 2 * g + t;
                unsigned wh, wl;
                SP::split_pair(v[VOFF + 8 * (s + 1) + 2 * j], v[VOFF + 8 * (s + 1) + 2 * j + 1], wh, wl);
                asm volatile("" : "+v"(wh), "+v"(wl));      // pins the six VALU operations HERE (pure code would sink to its use)
                nhi[j] = wh;
                nlo[j] = wl;
            }
            if constexpr (t == 2) after(std::integral_constant<int, s>{}, std::integral_constant<int, g>{}, bhi, blo);
        });
        if constexpr (!(FIRST && i < 3)) ring.template fetch_after_unit<pos>();
    });
}

}  // namespace nerf

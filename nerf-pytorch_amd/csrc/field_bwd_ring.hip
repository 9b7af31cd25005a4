// field_dgrad3r_kernel: the delta chain of the split datapaths (d_raw -> dL/d(pre-activation) of every layer; the autograd of
// run_nerf_helpers.py:96-119) on the weight RING of field_ring.h.  32 points per wave on v_mfma_f32_32x32x16_{bf16,f16}, 4 waves,
// one per SIMD.  A wave alone on its SIMD has nobody to cover what it does between MFMAs (the double-buffered kernel of round 2,
// deleted in round 5, spent that time after every 64 KiB chunk: barrier, 16 DMA pieces per wave back to back, LDS latency
// of the first fragments, 32 row stores in a burst: MFMA-busy 0.53).  Here every MFMA (32 cycles of pipe) carries its own share of
// that work in its shadow: one fragment request 8 MFMAs ahead of its use, a piece of the next k-step's operand split, one row
// store, and behind each of four units per chunk a 4 KiB DMA part (field_ring.h: unit_pipelined, WeightRingT<4>).
// Same transposed fragment stream (P3B), same MFMA order per accumulator, same masks as that kernel: with SP = SplitBF16 every
// delta written was BIT-IDENTICAL to it while both existed (the digests of tests/golden/kernel_digests.json were recorded then).  Deltas leave as 16-bit elements
// of the split's type SP (split_types.h; operands of wgrad1_kernel).  SP = SplitF16: the chain runs on s * d_raw, s a power of
// two per launch (delta_amax_kernel).
#include <type_traits>
#include "field_ring.h"
#include "launchers.h"

namespace nerf {

struct FieldBwdRingArgs {
    const float* packed3;
    const float* act;       // saved by the forward (ReLU bitmasks in the bf16x3 lane order)
    const float* d_raw;     // [P][4]
    float* delta;           // delta_layout3(P): 32-point feature-major tiles
    int n_rays, S;
};

// units of the transposed stream (P3B) in consumption order: W'^T 8 k-steps x 2 | (feature_linear^T 32: skipped) | L7^T .. L1^T 7 x 32
constexpr int BWD3_UNITS_VIEWS = 16;
constexpr int BWD3_UNITS_SKIP = 32;
constexpr int BWD3_UNITS = BWD3_UNITS_VIEWS + 7 * 32;
static_assert((BWD3_UNITS_VIEWS + BWD3_UNITS_SKIP) * UNIT_WORDS == P3B_L7 - P3B_VIEWS, "unit arithmetic vs nerf_common.h");
static_assert(BWD3_UNITS_VIEWS % CHUNK_UNITS == 0, "the trunk starts on a chunk boundary");

template <int NV>
__device__ inline void apply_mask3r(float (&d)[NV], const f32x16* acc, u32x4 m) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        // sign-extended 1-bit field = all ones / zero (one v_bfe_i32), AND-ed onto the accumulator: bit ? acc : +0
        // (asm: LLVM canonicalises `x & sext(bit)` back into and + compare + select, three operations)
        unsigned keep;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m[i >> 5]), "n"(i & 31));
        d[i] = __uint_as_float(__float_as_uint(acc[i >> 4][i & 15]) & keep);
    }
}

// max|d_raw| over the launch (its bit pattern: non-negative floats order like unsigned integers, so the atomic maximum does not
// depend on the order of the atomics) -> DeltaLayout3::scale word 0, zeroed by the launcher; the consumers derive the power-of-two
// scale from it (nerf_common.h, delta_scale_bits).
__global__ __launch_bounds__(1024) void delta_amax_kernel(const f32x4* __restrict__ d_raw, long n4, unsigned* __restrict__ slot) {
    float m = 0.0f;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long)gridDim.x * 1024) {
        const f32x4 v = d_raw[i];
        m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), m);     // (NaN: fmaxf keeps the other operand)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[16];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 16) {
        m = wm[threadIdx.x];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (threadIdx.x == 0) atomicMax(slot, __float_as_uint(m));
    }
}

// TWO (round 6, "fp16x3w"): every delta leaves as TWO 16-bit words, hi = T(v) and lo = T(v - hi) -- the lo words go to the mirror of
// the layout at DeltaLayout3::lo (delta_layout3(P, true)); the save buffer it reads was written by the two-word forward (SAVE = 3).
template <typename SP, bool TWO = false>
__global__ __launch_bounds__(FIELD3_WAVES * 64) void field_dgrad3r_kernel(FieldBwdRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const size_t P = (size_t)a.n_rays * a.S;
    const size_t p_raw = ((size_t)blockIdx.x * FIELD3_WAVES + wave) * PTS_PER_WAVE3 + (lane & 31);
    const bool valid = p_raw < P;
    const size_t p = valid ? p_raw : P - 1;

    WeightRingT<FIELD3_WAVES> ring;
    ring.start(a.packed3 + P3B_VIEWS, lds, wave, lane, BWD3_UNITS_VIEWS, BWD3_UNITS_SKIP, BWD3_UNITS);
    stage_small_ring(a.packed3 + P3_SMALL, lds, FIELD3_WAVES * 64);

    const ActLayout3 al = act_layout3(P, (size_t)a.n_rays, TWO);
    const DeltaLayout3 dl = delta_layout3(P, TWO);
    const size_t tile = (size_t)blockIdx.x * FIELD3_WAVES + wave;          // this wave's tile of every delta region
    f32x4 g = *reinterpret_cast<const f32x4*>(a.d_raw + p * 4);             // (d_rgb3, d_sigma)
    if (SP::F16) {      // the whole chain is linear in d_raw: run it on s * d_raw (s = 2^k, exact), see DeltaLayout3::scale
        const float sc = __uint_as_float(delta_scale_bits(reinterpret_cast<const unsigned*>(a.delta + dl.scale)[0], false));
        g = f32x4{g[0] * sc, g[1] * sc, g[2] * sc, g[3] * sc};
    }
    if (valid) {        // tile-major copy of d_raw: the A operand of the rgb_linear / alpha_linear weight gradients
        const size_t goff = tile * (4 * 32) + half * 64 + (lane & 31);
        unsigned short* gt = reinterpret_cast<unsigned short*>(a.delta + dl.graw) + goff;
        nt_store(gt, SP::cvt1(half ? g[2] : g[0]));
        nt_store(gt + 32, SP::cvt1(half ? g[3] : g[1]));
        if (TWO) {
            nt_store(gt + 2 * dl.lo, split_lo<SP>(half ? g[2] : g[0]));
            nt_store(gt + 2 * dl.lo + 32, split_lo<SP>(half ? g[3] : g[1]));
        }
    }
    u32x4 msk[D + 1];
    {
        const u32x4* mp = reinterpret_cast<const u32x4*>(a.act + al.mask) + p * 2 + half;
#pragma unroll
        for (int l = 0; l <= D; ++l) msk[l] = mp[(size_t)l * P * 2];
    }
    ring.ready();           // small parameters staged, chunks 0 and 1 landed

    // ---- rgb_linear^T (VALU) + ReLU mask of the view branch: lane value i = feature 32*(i>>4) + d32row(i&15, half)
    float dhv[64];
    {
        const float* wr = ring_small_ptr(lds, SM_WRGB);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int col = 32 * ob + 8 * q4 + 4 * half;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + col);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + WV + col);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 2 * WV + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ob + 4 * q4 + r;
                    const float v = g[0] * w0[r] + g[1] * w1[r] + g[2] * w2[r];
                    dhv[i] = ((msk[D][i >> 5] >> (i & 31)) & 1u) ? v : 0.0f;
                }
            }
    }

    f32x16 acc[8];
    float d[128];
    // 16-bit deltas: unconditional paired stores.  A wave whose tile lies beyond the
    // padded point range (last workgroup only) writes to a dump tile: the unused `feat` region.
    const bool tile_ok = __builtin_amdgcn_readfirstlane((int)(tile * 32 < pad32(P))) != 0;
    const unsigned odd = (unsigned)lane & 1u;
    const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;
    const int pair_off = ((lane >> 5) * 4 + (int)odd) * 16 + ((lane & 31) >> 1);      // dwords inside the tile
    // one paired 16-bit store: rows (r, r + 1) of 32-feature block ob of a F-wide region (32-point feature-major tiles,
    // nerf_common.h), `own` = the two values of this lane's point
    auto store_pair16 = [&](size_t region_off, int F, int ob, int r, unsigned own, bool lo_part = false) __attribute__((always_inline)) {
        unsigned* base = reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(a.delta + (tile_ok ? region_off : dl.feat) + (lo_part ? dl.lo : (size_t)0))
                                                     + (tile_ok ? tile * (size_t)(F * 32) : (size_t)0)) + pair_off;
        paired_store(own, pair_sel, [&](unsigned word) __attribute__((always_inline)) { nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, word); });
    };
    // the deltas in `v` (value 16 ob + r) leave while the next contraction consumes them: unit i of NU writes its share
    // (NV / NU values as 16-bit pairs)
    size_t store_region = 0;
    // row stores guaranteed behind the last fetch part (two per unit, positions 3..6); none under the store-less timing ablations
    constexpr int NP = (NERF_ABL_SAVE == 1 || NERF_ABL_SAVE == 3) ? 0 : (TWO ? 16 : 8);

    // ---- view branch folded with feature_linear (nerf_common.h): delta of the trunk output =
    //      (alpha_linear^T d_sigma + W'^T d_hv) * relu'(h7); the feature_linear^T units of the stream are skipped
    {
        const float* wa = ring_small_ptr(lds, SM_WALPHA);
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 32 * ob + 8 * q4 + 4 * half);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ob][4 * q4 + r] = g[3] * w[r];
            }
    }
    Frag fa, fb, fl;
    ring.request_first(fa);
    store_region = dl.hv;
    // unit (k-step kk, group gg) writes values 8 kk + 4 gg .. + 3: as bf16 pairs they ARE words 2 gg, 2 gg + 1 of the k-step's
    // hi fragment (value 2 q, 2 q + 1 = rows r = 2 (q % 8), r + 1 of block q / 8)
    ring_units<SP, 16, 2, 0, true, 0>(ring, fa, fb, fl, acc, dhv, [&](auto kk, auto gg, const u32x4& bhi, const u32x4& blo) __attribute__((always_inline)) {
        constexpr int i = 2 * decltype(kk)::value + decltype(gg)::value;            // unit 0..15: 64 values -> 4 per unit
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = 2 * i + t;
            store_pair16(store_region, WV, q / 8, 2 * (q % 8), bhi[2 * decltype(gg)::value + t]);
            if constexpr (TWO) store_pair16(store_region, WV, q / 8, 2 * (q % 8), blo[2 * decltype(gg)::value + t], true);
        }
    });
    apply_mask3r<128>(d, acc, msk[D - 1]);

    // ---- trunk: delta_{l-1} = (W_l^T delta_l) * relu'(h_{l-1}),  l = 7 .. 1
#pragma unroll 1
    for (int l = D - 1; l >= 1; --l) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
        store_region = (size_t)l * region_words3(pad32(P), W);              // dl.h[l]: delta of layer l = input of this step
        ring_units<SP, 32, 2, 0, false, NP>(ring, fa, fb, fl, acc, d, [&](auto kk, auto gg, const u32x4& bhi, const u32x4& blo) __attribute__((always_inline)) {
            constexpr int i = 2 * decltype(kk)::value + decltype(gg)::value;        // unit 0..31: 128 values -> 4 per unit
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q = 2 * i + t;
                store_pair16(store_region, W, q / 8, 2 * (q % 8), bhi[2 * decltype(gg)::value + t]);
                if constexpr (TWO) store_pair16(store_region, W, q / 8, 2 * (q % 8), blo[2 * decltype(gg)::value + t], true);
            }
        });
        u32x4 m = msk[0];                   // ReLU bitmask of h_{l-1} (static indices only: msk stays in registers)
#pragma unroll
        for (int t = 1; t < D; ++t) if (t == l - 1) m = msk[t];
        apply_mask3r<128>(d, acc, m);
    }
    // dl.h[0]
    store_tile16_pair<SP, 0, 8>(reinterpret_cast<unsigned short*>(a.delta + (tile_ok ? (size_t)0 : dl.feat)) + (tile_ok ? tile * (size_t)(W * 32) : (size_t)0), lane, d);
    if constexpr (TWO)
        store_tile16_pair<SP, 0, 8, 128, true>(reinterpret_cast<unsigned short*>(a.delta + dl.lo + (tile_ok ? (size_t)0 : dl.feat)) + (tile_ok ? tile * (size_t)(W * 32) : (size_t)0), lane, d);
}

template <typename SP, bool TWO = false>
static hipError_t launch_dgrad_one(const FieldBwdRingArgs& ba, unsigned blocks, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)field_dgrad3r_kernel<SP, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((field_dgrad3r_kernel<SP, TWO>), dim3(blocks), dim3(FIELD3_WAVES * 64), RING_LDS_FLOATS * 4, stream, ba);
    return hipGetLastError();
}

// split: 0 bf16, 1 fp16 parts (of the products and of the stored deltas); 5 = fp16 with two-word deltas (and a two-word save buffer)
hipError_t launch_field_dgrad3r(const float* packed3, const float* act, const float* d_raw, int n_rays, int S,
                                float* delta, int split, hipStream_t stream) {
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    FieldBwdRingArgs ba{packed3, act, d_raw, delta, n_rays, S};
    const unsigned blocks = (unsigned)((P + PTS_PER_WG3 - 1) / PTS_PER_WG3);
    if (split) {
        // (the scale word sits in the hi part of the layout: at the same offset in the one- and the two-word layout)
        unsigned* slot = reinterpret_cast<unsigned*>(delta + delta_layout3((size_t)P).scale);
        hipError_t e = hipMemsetAsync(slot, 0, 16, stream);
        if (e != hipSuccess) return e;
        const unsigned sb = (unsigned)min((long)1024, (P + 1023) / 1024);      // one 16-byte load per thread: the launch is latency, not bytes
        hipLaunchKernelGGL(delta_amax_kernel, dim3(sb), dim3(1024), 0, stream, reinterpret_cast<const f32x4*>(d_raw), P, slot);
        return split == 5 ? launch_dgrad_one<SplitF16, true>(ba, blocks, stream) : launch_dgrad_one<SplitF16>(ba, blocks, stream);
    }
    return launch_dgrad_one<SplitBF16>(ba, blocks, stream);
}

}  // namespace nerf

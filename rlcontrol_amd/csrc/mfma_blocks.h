// mfma_blocks.h -- gfx950 fp32 matrix-core (v_mfma_f32_16x16x4_f32) building blocks shared by the fused DDPG,
// SoftActorCritic and NAF update kernels (ddpg_mfma_kernel.h, sac_mfma_kernel.h, naf_mfma_kernel.h).
//
// One workgroup of 512 threads (8 waves, two per SIMD) owns one agent.  Everything [B, H]-sized stays on the CU:
//
//   LDS   hbuf   fp32 [MT*16][LDH]   the activation that feeds a 16-deep-chunked contraction: A operand of every
//                                    forward GEMM and B operand of the weight-gradient GEMMs
//         mask   u8   [MT*16][MSTRIDE]  relu masks of the hidden layers, one BYTE per (row, unit); a byte can hold
//                                    several masks as bit planes (SAC: pi / Q, NAF: mu / V branch), so that
//                                    d(hidden) = mask * (seed . wvec) -- a rank-NS outer product -- is regenerated on
//                                    the fly as an MFMA operand instead of being stored as [B,H] fp32
//         part   fp32 [8][MT*16][NJ]  per-wave partials of reductions over features, summed in a fixed order
//         wvec   fp32 [NS][256]       the staged output-layer weights that turn seeds into d(hidden)
//   VGPR  accumulators of the GEMM in flight (MT x NTW tiles of 16x16), weight fragments streamed
//         global -> VGPR (each weight element is read once per GEMM per agent; no LDS staging)
//   HBM   weights, target weights, Adam m/v: the big matrices are updated (Adam + Polyak) in the epilogue of their
//         weight-gradient GEMM straight from the accumulators
//
// Tiling: batch rows on the MFMA M axis (MT = ceil(B/16) tiles), features on N; wave w of 8 owns the ADJACENT
// N tiles NTW*w .. NTW*w+NTW-1 for ALL M tiles, so reductions over the batch (bias / output-layer / first-layer
// gradients) are wave-local and only reductions over features cross waves through LDS partials, summed in a fixed
// order (deterministic: K updates in one launch == K launches, bit for bit).
// fp32 in / fp32 accumulate MFMA is a k-ordered fmaf chain (exact fp32), so the 1e-5 parity bar holds.
//
// Big matrices use the TILE-BLOCKED layout of rlc_common.h (rlc_blk_index): every instruction of the weight
// streams touches 1 KB contiguous.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "rlc_common.h"

namespace mfb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every LDS pointer is typed with its address space: hipcc's address-space inference loses pointers that travel
// through structs and lambdas and then emits flat_load / flat_store (measured: 9 % of the DDPG update).
#define RLC_LDS __attribute__((address_space(3)))
typedef RLC_LDS float lds_f32;
typedef RLC_LDS f32x4 lds_f32x4;
typedef RLC_LDS unsigned char lds_u8;
typedef RLC_LDS unsigned int lds_u32;
typedef RLC_LDS int lds_i32;
typedef RLC_LDS double lds_f64;
typedef RLC_LDS long long lds_i64;

// Timing-only ablations for diagnostic builds (scripts/ab_ablate.sh; results are wrong by construction, the product
// build has RLC_ABLATE == 0): bit 0 no weight-gradient GEMM at all, 1 its k-loops only (no prefetch / Adam epilogue),
// 2 no action-row loop, 3 no first-layer pass, 4 sample + gather only in the first update of a launch, 5 no backward
// k-loops, 6 no forward k-loops, 7 no first-layer gradient, 8 no row_dot, 9 no mask stores, 10 no bias_relu,
// 11 weight-gradient epilogue without its stores, 12 without its W / m / v / W' loads, 13 without the Adam arithmetic
#ifndef RLC_ABLATE
#define RLC_ABLATE 0
#endif
constexpr bool ablate(int bit) { return ((RLC_ABLATE) >> bit) & 1; }

// Experiment switch (diagnostic builds): workgroup i of a launch starts (i mod RLC_STAGGER_WAYS) * RLC_STAGGER_US /
// RLC_STAGGER_WAYS microseconds late, so that the memory-bound phases of the agents do not coincide chip-wide
#ifndef RLC_STAGGER_US
#define RLC_STAGGER_US 0
#endif
#ifndef RLC_STAGGER_WAYS
#define RLC_STAGGER_WAYS 4
#endif
__device__ __forceinline__ void stagger_start() {
    if constexpr (RLC_STAGGER_US > 0) {
        const long long wait = (long long)(blockIdx.x % RLC_STAGGER_WAYS) * RLC_STAGGER_US * 100 / RLC_STAGGER_WAYS;   // 100 MHz ticks
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        __syncthreads();
    }
}

// Workgroup barrier for phase boundaries that hand over LDS data only.  __syncthreads() also waits for every outstanding
// global-memory operation of the wave (s_waitcnt vmcnt(0)): after a phase that stored taps or Adam state, that is a
// store round trip to HBM on the critical path.  lds_barrier() waits for the wave's LDS / scalar traffic only; global
// stores keep draining behind it.  Use it only where no wave reads, after the barrier, global data that another wave
// wrote since the last __syncthreads() (each kernel keeps full barriers at those points).
// Measured (profiles/r03_variant_timings_s5.txt): no difference on any of the three kernels -- the store round trips are
// not on the critical path -- so the default stays __syncthreads(); -DRLC_LDS_BARRIERS enables the relaxed form.
__device__ __forceinline__ void lds_barrier() {
#ifdef RLC_LDS_BARRIERS
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

constexpr int kThreads = 512;
constexpr int kWaves = 8;     // two waves per SIMD: one can issue MFMA while the other does VALU / waits on loads
constexpr int SMAX = 8;       // state rows are padded to 8 floats in LDS (two ds_read_b128)

// Row stride (bytes) of the byte masks for NT16 N tiles: 4*odd dwords, so that the dword a lane reads in the
// backward GEMM (row 16mt+c, bytes nc+4g..+3) sits in bank (4*odd*c + g + const) mod 64: conflict-free for all lanes.
constexpr int mask_stride(int nt16) { return ((nt16 + 1) & 1) ? 16 * (nt16 + 1) : 16 * (nt16 + 2); }

// Adam's m / v slots are touched once per update, the target matrix twice (its forward GEMM and this epilogue): they are
// loaded / stored non-temporally (the `nt` bit), so that they do not displace the weight matrices the GEMMs re-read from
// L2 / the Infinity Cache several times per update.  Measured on one box, 256 agents (profiles/r03_variant_timings_*):
// m / v: DDPG +3.5 %, SoftActorCritic +8 %, NAF +3.7 %; the target stream on top: DDPG +1.3 %, NAF +3 %, SAC +-0.
// -DRLC_NO_NT_STATE / -DRLC_NO_NT_TARGET switch them off for A/B runs.
#ifndef RLC_NO_NT_STATE
#define RLC_NT_STATE 1
#endif
#ifndef RLC_NO_NT_TARGET
#define RLC_NT_TARGET 1
#endif
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
#ifdef RLC_NT_STATE
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void st_stream(float* p, f32x4 v) {
#ifdef RLC_NT_STATE
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}

// a replay field of the minibatch gather: a random 64-byte sector per field and sample, never re-read
template <class T>
__device__ __forceinline__ T ld_gather(const T* p) {
#ifdef RLC_NT_GATHER
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
// a weight dword of a forward k-loop; STREAM: the matrix is read once per update by this GEMM (a target network's): nt
template <bool STREAM>
__device__ __forceinline__ float ld_w(const float* p) {
#ifdef RLC_NT_TARGET_GEMM
    if constexpr (STREAM) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
__device__ __forceinline__ f32x4 ld_target(const float* p) {
#ifdef RLC_NT_TARGET
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void st_target(float* p, f32x4 v) {
#ifdef RLC_NT_TARGET
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer products, block l / 4 of the wave; lane l gives row l % 4 of
// its block's A column and column l % 4 of its B row, and receives column l % 4 of the block's 4 x 4 result (one row per
// register).  15 cycles of the matrix pipe against the 16x16x4's 32 (scripts/micro/mfma4x4_tail.hip): see Blk's T4.
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// sum over the 16 lanes that share lane>>4 (rotate-reduce with DPP row_ror: every lane gets the sum)
template <int ROR>
__device__ __forceinline__ float dpp_ror_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + ROR, 0xf, 0xf, false);
    return x + __int_as_float(y);
}
__device__ __forceinline__ float row16_sum(float x) {
    x = dpp_ror_add<8>(x);
    x = dpp_ror_add<4>(x);
    x = dpp_ror_add<2>(x);
    x = dpp_ror_add<1>(x);
    return x;
}
// sum over the 4 lane groups (lanes l, l+16, l+32, l+48)
__device__ __forceinline__ float col4_sum(float x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// Blk's T4 applies when the minibatch ends within the first four rows of its last 16-row tile
__host__ __device__ inline bool rlc_tail4(int B, int MT) { return B > 16 * (MT - 1) && B <= 16 * (MT - 1) + 4; }
// ... and the dispatchers take the tail-of-four instantiations then, unless RLC_NO_TAIL4=1 (A/B switch, read once)
inline bool rlc_tail4_enabled(int B, int MT) {
    static const bool off = [] { const char* e = getenv("RLC_NO_TAIL4"); return e && e[0] == '1'; }();
    return !off && rlc_tail4(B, MT);
}

__host__ __device__ inline int ldh_for(int H1) {
    // leading dimension with (LDH/4) % 16 == 2: conflict-free ds_read_b128 rows AND b32 columns
    int q = (H1 + 3) / 4;
    while ((q & 15) != 2) q++;
    return q * 4;
}

struct BlkLds {
    lds_f32* hbuf;
    lds_u8* mask;
    lds_f32x4* xbuf;    // [MT - ceil(MT/4)][64] hand-off of a split tile's accumulators, or null: no tile is split
};

struct NoExtra {};
// A further contribution to dL/dh1[b][k] from heads that hang off the first layer (NAF's L heads): head gradients
// dhd [MB][4] (LDS, zero beyond the heads in use) times head weights wh [4][256] (LDS).  The column's four weights are
// read once per column (HeadExtra::column), the row's gradients once per element (HeadExtra::at): one LDS read per
// element instead of four, and no branch around them.
struct HeadExtra {
    const RLC_LDS float* dhd;
    const RLC_LDS float* wh;
    typedef float f4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ f4 column(int k) const { return f4{wh[k], wh[256 + k], wh[512 + k], wh[768 + k]}; }
    __device__ __forceinline__ float at(const f4& w, int b) const {
        const f4 dh = *reinterpret_cast<const RLC_LDS f4*>(&dhd[b * 4]);
        return dh[0] * w[0] + dh[1] * w[1] + dh[2] * w[2] + dh[3] * w[3];
    }
};

// Adam on the small tensors (first layer, a concat layer's extra rows): the hardware-sqrt/rcp form like the big
// matrices' epilogue unless the build asks for the IEEE-exact expansions
#ifdef RLC_ADAM_SMALL_EXACT
#define RLC_ADAM_SMALL adam_step
#else
#define RLC_ADAM_SMALL adam_step_fast
#endif

// MT: M tiles (batch rows / 16); NTW: N tiles per wave (1: widths <= 128, 2: widths <= 256); MSTRIDE: mask row bytes;
// LERP: target update written (1-tau)*t + tau*w (sac_network.py:72-73) instead of t + tau*(w - t)
// (hydra_ddpg_network.py:29, naf_network.py:62-63)
// TADAM: torch.optim.Adam's step (the KL agents): the caller folds sqrt(1 - b2^t) into alpha and sets adam_eps =
// 1e-8 * sqrt(1 - b2^t) (see kl_generic.hip); otherwise TF's ApplyAdam with its constant epsilon
// T4: the minibatch ends within the first FOUR rows of its last 16-row tile (batch 100 = six tiles + 4 rows).  The
// k-loops then run that tile through v_mfma_f32_4x4x1_16b_f32 instead of a 16x16x4 whose other twelve rows are padding:
// with the very same B register (lane 16g + c: the weight of k = 4g + s, column c) and the A value of row c & 3
// instead of row c, block (g, c / 4) of the instruction accumulates rows 0..3 x columns 4(c/4)..+3 over the k's of lane
// group g; a butterfly over g after the loop (tail_finish) completes the sum, and lanes g == 0 hold rows 0..3 of column
// c -- exactly what they hold of a 16x16x4 accumulator, so every epilogue is unchanged (lanes g > 0 = rows 4..15 = 0).
template <int MT, int NTW, int MSTRIDE, bool LERP = false, bool TADAM = false, bool T4 = false>
struct Blk {
    static constexpr int MB = MT * 16;
    static constexpr int TROW = 16 * (MT - 1);          // first row of the last batch tile

    __device__ __forceinline__ static float polyak(float t, float w, float tau) {
        return LERP ? (1.0f - tau) * t + tau * w : t + tau * (w - t);
    }

    // per-thread geometry
    int tid, lane, wave, c, g;
    int S, H1, B, LDH;      // H1 = width of the activation held in hbuf (k-dim of the forward GEMMs)
    BlkLds L;
    float adam_eps;         // TADAM only
    __device__ __forceinline__ float astep_big(float w, float gr, float& m, float& v, float alpha) const {
        if constexpr (TADAM) return adam_step_fast_eps(w, gr, m, v, alpha, adam_eps);
        else return adam_step_fast(w, gr, m, v, alpha);
    }
    __device__ __forceinline__ float astep_small(float w, float gr, float& m, float& v, float alpha) const {
        if constexpr (TADAM) return adam_step_fast_eps(w, gr, m, v, alpha, adam_eps);
        else return RLC_ADAM_SMALL(w, gr, m, v, alpha);
    }
#ifdef RLC_STAMPS
    float* stamp_buf = nullptr;
    long long t_sub = 0;
    __device__ __forceinline__ void sub_begin() { if (tid == 0) t_sub = clock64(); }
    __device__ __forceinline__ void sub_stamp(int i) {
        if (tid == 0 && stamp_buf) { const long long t = clock64(); stamp_buf[i] += (float)(t - t_sub); t_sub = t; }
    }
#else
    __device__ __forceinline__ void sub_begin() {}
    __device__ __forceinline__ void sub_stamp(int) {}
#endif

    __device__ __forceinline__ void init_geometry() {
        tid = threadIdx.x; lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        c = lane & 15; g = lane >> 4;
        L.xbuf = nullptr;
    }

    // ---------------------------------------------------------------------------------------
    // N-tile ownership.  NTW = 1: wave w owns tile w.  NTW = 2: waves 0-3 own the adjacent tiles {2w, 2w+1}, waves 4-7
    // (the SIMD partners of waves 0-3) own {8 + (w-4), 12 + (w-4)}: with 13 tiles (widths 196..208) the SIMDs carry
    // 4/3/3/3 tiles instead of 4/4/3/2, and the ragged 13th tile is then SPLIT over the batch between waves 4-7 in
    // the k-loops (fwd_gemm / bwd_gemm), so that every SIMD issues 3 tiles + 2 of 7 batch tiles; wave 4 collects the
    // shares through L.xbuf and owns the tile in every epilogue.
    // ---------------------------------------------------------------------------------------
    __device__ __forceinline__ int tile0() const { return NTW == 1 ? wave : (wave < 4 ? 2 * wave : 4 + wave); }
    __device__ __forceinline__ int tstep() const { return (NTW == 2 && wave >= 4) ? 4 : 1; }
    __device__ __forceinline__ int tile_of(int i) const { return tile0() + i * tstep(); }
    __device__ __forceinline__ int nown_of(int NT) const {
        int n = 0;
#pragma unroll
        for (int i = 0; i < NTW; i++) n += tile_of(i) < NT ? 1 : 0;
        return n;
    }
    static constexpr int MXS = (MT + 3) / 4;            // batch tiles of a split tile per wave (at most)
    __device__ __forceinline__ bool split_mode(int NT) const { return NTW == 2 && NT == 13 && L.xbuf != nullptr; }
    // batch-tile range [lo, hi) of wave 4+s's share of the split tile.  T4: waves 4-6 take MXS full tiles each, wave 7
    // the four-row tail alone (XMODE 2 of the loops)
    static_assert(!T4 || NTW != 2 || 3 * MXS == MT - 1, "T4 shares of the split tile: three waves of MXS full tiles + the tail");
    __device__ __forceinline__ static int share_lo(int s) { return T4 ? (s < 4 ? s * MXS : MT) : (s * MT + 3) / 4; }
    // XMODE of the k-loops: 0 = the wave's own tiles only, 1 = + batch tiles [xm0, xm0 + MXS) of the split tile,
    // 2 = + the split tile's four-row tail (T4)
    __device__ __forceinline__ int xmode_of_wave() const { return (T4 && wave == 7) ? 2 : 1; }
    // complete a T4 tail accumulator: sum the four lane groups' partial products; rows 4..15 of the tile are zero
    __device__ __forceinline__ f32x4 tail_finish(f32x4 v) const {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] += __shfl_xor(v[r], 16, 64);
            v[r] += __shfl_xor(v[r], 32, 64);
        }
        return g == 0 ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---------------------------------------------------------------------------------------
    // hbuf[b][k] = relu(b1[k] + sum_i xs[b][i] W1[i][k])   (rows >= B and columns >= H1 zeroed)
    // ---------------------------------------------------------------------------------------
    // WP: const float* (the blob in global memory) or const lds_f32* (a copy staged in LDS: stage_load / stage_store)
    template <class WP>
    __device__ __forceinline__ void trunk(WP W1, WP b1, const lds_f32* xs) {
        if constexpr (ablate(3)) return;
        if (S <= 4) trunk_t<4>(W1, b1, xs);      // wave-uniform: Pendulum-sized states need one 16-byte read per row
        else trunk_t<SMAX>(W1, b1, xs);
    }
    // First-layer weights staged in LDS (opt-in, -DRLC_W1_STAGE: measured -0.6 % on DDPG, -1.4 % on SoftActorCritic,
    // profiles/r03_variant_timings_s6.txt -- the dependent load was not what the passes wait for).  The first layer ([S][H1] + bias, a few KB) is re-read from global memory by
    // every first-layer pass -- a dependent global load (L2 or HBM latency) in front of a microsecond of arithmetic,
    // several times per update.  Staged once per update instead: the loads are issued before the minibatch is sampled
    // and gathered (stage_load: kStage registers per set), stored to LDS behind it (stage_store), and the passes read
    // [S][H1] weights then [H1] biases from there; a first-layer Adam step that a later pass must see writes its new
    // values into the LDS copy as well (trunk_grad_adam's `stage` argument).
    static constexpr int kStage = ((SMAX + 1) * 256 + kThreads - 1) / kThreads;
    __device__ __forceinline__ void stage_load(float (&r)[kStage], const float* W1, const float* b1) const {
        const int nw = S * H1, n = nw + H1;
#pragma unroll
        for (int j = 0; j < kStage; j++) {
            const int i = tid + kThreads * j;
            r[j] = i < nw ? W1[i] : (i < n ? b1[i - nw] : 0.0f);
        }
    }
    __device__ __forceinline__ void stage_store(const float (&r)[kStage], lds_f32* ws) const {
        const int n = (S + 1) * H1;
#pragma unroll
        for (int j = 0; j < kStage; j++) {
            const int i = tid + kThreads * j;
            if (i < n) ws[i] = r[j];
        }
    }
    template <int SP, class WP>
    __device__ __forceinline__ void trunk_t(WP W1, WP b1, const lds_f32* xs) {
        // lane = a quad of 4 adjacent columns (its S x 4 weights and 4 biases stay in registers), wave w = rows
        // w, w+8, ...: per row one broadcast read of the state and ONE 16-byte store of four activations
        // (a quarter of the LDS store instructions of the one-column-per-thread form; same i-order per element)
        for (int q = lane; 4 * q < LDH; q += 64) {
            f32x4 w[SP], bias;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int k = 4 * q + e;
                const bool live = k < H1;
#pragma unroll
                for (int i = 0; i < SP; i++) w[i][e] = (live && i < S) ? W1[i * H1 + k] : 0.0f;
                bias[e] = live ? b1[k] : 0.0f;
            }
            const bool live4[4] = {4 * q < H1, 4 * q + 1 < H1, 4 * q + 2 < H1, 4 * q + 3 < H1};
#ifndef RLC_TRUNK_UNROLL
#define RLC_TRUNK_UNROLL 2
#endif
#pragma unroll RLC_TRUNK_UNROLL
            for (int b = wave; b < MB; b += kWaves) {
                const f32x4 x0 = *reinterpret_cast<const lds_f32x4*>(&xs[b * SMAX]);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; i++) acc += x0[i] * w[i];
                if (SP > 4) {
                    const f32x4 x1 = *reinterpret_cast<const lds_f32x4*>(&xs[b * SMAX + 4]);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc += x1[i] * w[(SP > 4 ? 4 : 0) + i];
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (live4[e] && b < B) ? fmaxf(acc[e] + bias[e], 0.0f) : 0.0f;
                *reinterpret_cast<lds_f32x4*>(&L.hbuf[b * LDH + 4 * q]) = o;
            }
        }
    }

    // ---------------------------------------------------------------------------------------
    // forward GEMM: acc[mt][i] (tile rows 16mt.., cols 16*(NTW*wave+i)..) = hbuf[:, 0:K] . W[0:K, :]
    // A: one ds_read_b128 per M tile per 16-deep chunk, lane (c,g) holds k = kc+4g+s for step s;
    // B: tile-blocked W (rlc_blk_index): block (kc/16, t) holds rows kc..kc+15 of tile t; this lane needs rows
    //    4g+s of column c -> four dwords 16 B apart inside the block's 1 KB, streamed global -> VGPR.
    // No masks anywhere in the loop: blocks are zero-padded to 16x16 in memory (rows K..16*ceil(K/16)-1 are
    // zeros -- a critic's action rows live in their own block row), hbuf columns >= H1 are zeros or
    // finite neighbours (times a zero weight), and the number of tiles a wave owns is a template
    // parameter.  Two register sets (A and B fragments of the chunk in flight / the next chunk) alternate in a
    // loop unrolled by two, so there are no register-rotation moves either: per chunk a wave issues
    // MT ds_read_b128 + 4*NOWN global_load_dword + 4*MT*NOWN MFMAs and little else.
    // ---------------------------------------------------------------------------------------
    // XMODE (see share_lo): besides its NOWN tiles the wave computes its share of the split tile xt into accx
    template <int NOWN, int XMODE, bool STREAM = false>
    __device__ __forceinline__ void fwd_loop(f32x4 (&acc)[MT][NTW], const float* W, int NT, int KB, bool tail8,
                                             f32x4 (&accx)[MXS], int xt, int xm0) {
        constexpr bool XTRA = XMODE != 0;
        constexpr int NX = XMODE == 2 ? 1 : MXS;                    // slots of accx in use
        const int lofs = ((((c >> 2) << 4) + 4 * g) << 2) + (c & 3);
        const float* wp = W + ((size_t)tile0() << 8) + lofs;
        const int tst = tstep() << 8;
        const size_t wstep = (size_t)NT << 8;                       // floats between block rows
        const lds_f32* ap = L.hbuf + c * LDH + 4 * g;
        const lds_f32* apt = L.hbuf + (TROW + (c & 3)) * LDH + 4 * g;      // T4: the tail tile's A rows
        auto arow = [&](int mt) { return (T4 && mt == MT - 1) ? apt : ap + 16 * mt * LDH; };
        f32x4 a0[MT], a1[MT];
        float b0[NOWN][4], b1[NOWN][4];
        f32x4 tq[NOWN];                                             // T4: the tail tile's 4x4x1 accumulators
#pragma unroll
        for (int i = 0; i < NOWN; i++) tq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the split tile's share: its own A fragments (the batch tiles are a run-time range) and B fragment
        const float* wpx = W + ((size_t)xt << 8) + lofs;
        const lds_f32* apx = ap + 16 * xm0 * LDH;
        auto xrow = [&](int m) { return XMODE == 2 ? apt : apx + 16 * (xm0 + m < MT ? m : 0) * LDH; };
        f32x4 ax0[NX], ax1[NX];
        float bx0[4], bx1[4];
        auto loadX = [&](f32x4 (&da)[NX], float (&db)[4], int ch) {
            if (XTRA) {
#pragma unroll
                for (int m = 0; m < NX; m++)       // a share shorter than MXS repeats its first tile (result unused)
                    da[m] = *reinterpret_cast<const lds_f32x4*>(xrow(m) + 16 * ch);
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) db[s2] = ld_w<STREAM>(&wpx[(size_t)ch * wstep + 4 * s2]);
            }
        };
        auto macX = [&](const f32x4 (&da)[NX], const float (&db)[4]) {
            if (XTRA) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
                    for (int m = 0; m < NX; m++)
                        accx[m] = XMODE == 2 ? mfma4(da[m][s2], db[s2], accx[m]) : mfma16(da[m][s2], db[s2], accx[m]);
            }
        };
        auto loadA = [&](f32x4 (&dst)[MT], int ch) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) dst[mt] = *reinterpret_cast<const lds_f32x4*>(arow(mt) + 16 * ch);
        };
        auto loadB = [&](float (&dst)[NOWN][4], int ch) {
#pragma unroll
            for (int i = 0; i < NOWN; i++)
#pragma unroll
                for (int s = 0; s < 4; s++) dst[i][s] = ld_w<STREAM>(&wp[(size_t)ch * wstep + i * tst + 4 * s]);
        };
        auto mac = [&](const f32x4 (&a)[MT], const float (&b)[NOWN][4]) {
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (T4 && mt == MT - 1) tq[i] = mfma4(a[mt][s], b[i][s], tq[i]);
                        else acc[mt][i] = mfma16(a[mt][s], b[i][s], acc[mt][i]);
                    }
        };
        loadB(b0, 0);
        loadA(a0, 0);
        loadX(ax0, bx0, 0);
        int ch = 0;
        // Branch-free body (the prefetch index is clamped: an even chunk count re-reads its last chunk once) so that
        // both halves form ONE scheduling region, pinned by sched_group_barrier to: the 4*NOWN weight loads of the
        // next chunk first, then one A-fragment read per 4*NOWN MFMAs -- every load is issued a whole chunk
        // (4*MT*NOWN MFMAs) before its first use instead of wherever the scheduler sinks it.
        auto pin = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x020, 4 * NOWN + (XTRA ? 4 : 0), 0);      // VMEM read
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NOWN, 0);      // MFMA
            }
            if (XTRA) {
                __builtin_amdgcn_sched_group_barrier(0x100, NX, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NX, 0);
            }
        };
        for (; ch + 2 <= KB; ch += 2) {
            loadB(b1, ch + 1);
            loadA(a1, ch + 1);
            loadX(ax1, bx1, ch + 1);
            mac(a0, b0);
            macX(ax0, bx0);
            pin();
            const int nx = ch + 2 < KB ? ch + 2 : KB - 1;
            loadB(b0, nx);
            loadA(a0, nx);
            loadX(ax0, bx0, nx);
            mac(a1, b1);
            macX(ax1, bx1);
            pin();
        }
        if (ch < KB) { mac(a0, b0); macX(ax0, bx0); }      // odd chunk count: the last chunk is already loaded
        if (tail8) {
            // K = 16 KB + 8: the last 8 k's in TWO steps instead of a zero-padded chunk of four -- lane group g takes
            // k = 16 KB + 2g + s (an 8-byte read of hbuf, two weight dwords per tile)
            const int kofs = 16 * KB + 2 * g - 4 * g;              // relative to the chunk pointers (which carry + 4g)
            const int tofs = ((((c >> 2) << 4) + 2 * g) << 2) + (c & 3);
            const float* wt = W + ((size_t)tile0() << 8) + (size_t)KB * wstep + tofs;
            float bt[NOWN][2];
#pragma unroll
            for (int i = 0; i < NOWN; i++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) bt[i][s2] = wt[i * tst + 4 * s2];
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) av[mt] = *reinterpret_cast<const RLC_LDS f32x2*>(arow(mt) + kofs);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (T4 && mt == MT - 1) tq[i] = mfma4(av[mt][s2], bt[i][s2], tq[i]);
                        else acc[mt][i] = mfma16(av[mt][s2], bt[i][s2], acc[mt][i]);
                    }
            if (XTRA) {
                const float* wtx = W + ((size_t)xt << 8) + (size_t)KB * wstep + tofs;
#pragma unroll
                for (int m = 0; m < NX; m++) {
                    const f32x2 avx = *reinterpret_cast<const RLC_LDS f32x2*>(xrow(m) + kofs);
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++)
                        accx[m] = XMODE == 2 ? mfma4(avx[s2], wtx[4 * s2], accx[m]) : mfma16(avx[s2], wtx[4 * s2], accx[m]);
                }
            }
        }
        if constexpr (T4) {
#pragma unroll
            for (int i = 0; i < NOWN; i++) acc[MT - 1][i] += tail_finish(tq[i]);
            if constexpr (XMODE == 2) accx[0] = tail_finish(accx[0]);
        }
    }

    // wave 4 + s holds batch tiles [share_lo(s), share_lo(s+1)) of the split tile in accx: waves 5-7 pass theirs through
    // L.xbuf, wave 4 assembles the whole tile in acc[.][1] (ACCUM: adds it)
    template <bool ACCUM>
    __device__ __forceinline__ void collect_split(f32x4 (&acc)[MT][NTW], const f32x4 (&accx)[MXS]) {
        if constexpr (NTW >= 2) {
            if (wave >= 5) {
                const int lo = share_lo(wave - 4), hi = share_lo(wave - 3);
#pragma unroll
                for (int m = 0; m < MXS; m++)
                    if (lo + m < hi) L.xbuf[(lo + m - MXS) * 64 + lane] = accx[m];
            }
            __syncthreads();
            if (wave == 4) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    f32x4 v;
                    if (mt < MXS) v = accx[mt < MXS ? mt : 0];
                    else v = L.xbuf[(mt - MXS) * 64 + lane];
                    acc[mt][1] = ACCUM ? acc[mt][1] + v : v;
                }
            }
        }
    }

    template <bool STREAM = false>
    __device__ __forceinline__ void fwd_gemm(f32x4 (&acc)[MT][NTW], const float* W, int N, int K) {
        const int NT = (N + 15) >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) acc[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef RLC_STAMPS
        const long long t_w0 = clock64();
#endif
        if constexpr (ablate(6)) return;
        const int nown = nown_of(NT);              // tiles this wave owns: wave-uniform
        const bool tail8 = (K & 15) == 8 && K > 16;
        const int KB = tail8 ? K >> 4 : (K + 15) >> 4;
        f32x4 accx[MXS];
        if constexpr (NTW >= 2) {
            if (split_mode(NT)) {                  // workgroup-uniform
#pragma unroll
                for (int m = 0; m < MXS; m++) accx[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (T4 && wave == 7) fwd_loop<1, T4 ? 2 : 1, STREAM>(acc, W, NT, KB, tail8, accx, NT - 1, share_lo(3));
                else if (wave >= 4) fwd_loop<1, 1, STREAM>(acc, W, NT, KB, tail8, accx, NT - 1, share_lo(wave - 4));
                else fwd_loop<2, 0, STREAM>(acc, W, NT, KB, tail8, accx, 0, 0);
                collect_split<false>(acc, accx);
            } else if (nown >= 2) fwd_loop<2, 0, STREAM>(acc, W, NT, KB, tail8, accx, 0, 0);
            else if (nown == 1) fwd_loop<1, 0, STREAM>(acc, W, NT, KB, tail8, accx, 0, 0);
        } else {
            if (nown >= 1) fwd_loop<1, 0, STREAM>(acc, W, NT, KB, tail8, accx, 0, 0);
        }
#ifdef RLC_STAMPS
        if (lane == 0 && stamp_buf) stamp_buf[48 + wave] += (float)(clock64() - t_w0);   // per-wave k-loop cycles
#endif
    }

    // ---------------------------------------------------------------------------------------
    // TWO forward GEMMs over the same activation image in one k-loop: accA = hbuf . WA, accB = hbuf . WB (same N and K;
    // e.g. NAF's mu and V branches, DDPG's actor and critic second layers).  Every A fragment read from LDS feeds twice
    // the MFMAs, one loop prologue / drain and one barrier fall away, and neither accumulator set has to be parked in
    // registers across the other GEMM's loop (where the compiler spilled it: NAF 224 VGPRs = 0.6 MB of scratch traffic
    // per update).  Same structure and pinning as fwd_loop; summation order per output element unchanged.
    // ---------------------------------------------------------------------------------------
    template <int NOWN, int XMODE, bool STREAM = false>
    __device__ __forceinline__ void fwd_loop2(f32x4 (&accA)[MT][NTW], f32x4 (&accB)[MT][NTW], const float* WA, const float* WB,
                                              int NT, int KB, bool tail8, f32x4 (&accxA)[MXS], f32x4 (&accxB)[MXS], int xt,
                                              int xm0) {
        constexpr bool XTRA = XMODE != 0;
        constexpr int NX = XMODE == 2 ? 1 : MXS;
        const int lofs = ((((c >> 2) << 4) + 4 * g) << 2) + (c & 3);
        const size_t t0 = ((size_t)tile0() << 8) + lofs;
        const float* wpA = WA + t0;
        const float* wpB = WB + t0;
        const int tst = tstep() << 8;
        const size_t wstep = (size_t)NT << 8;
        const lds_f32* ap = L.hbuf + c * LDH + 4 * g;
        const lds_f32* apt = L.hbuf + (TROW + (c & 3)) * LDH + 4 * g;      // T4: the tail tile's A rows
        auto arow = [&](int mt) { return (T4 && mt == MT - 1) ? apt : ap + 16 * mt * LDH; };
        f32x4 a0[MT], a1[MT];
        float b0[2][NOWN][4], b1[2][NOWN][4];
        f32x4 tqA[NOWN], tqB[NOWN];                                        // T4: the tail tile's 4x4x1 accumulators
#pragma unroll
        for (int i = 0; i < NOWN; i++) { tqA[i] = f32x4{0.f, 0.f, 0.f, 0.f}; tqB[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const size_t tx = ((size_t)xt << 8) + lofs;
        const float* wpxA = WA + tx;
        const float* wpxB = WB + tx;
        const lds_f32* apx = ap + 16 * xm0 * LDH;
        auto xrow = [&](int m) { return XMODE == 2 ? apt : apx + 16 * (xm0 + m < MT ? m : 0) * LDH; };
        f32x4 ax0[NX], ax1[NX];
        float bx0[2][4], bx1[2][4];
        auto loadX = [&](f32x4 (&da)[NX], float (&db)[2][4], int ch) {
            if (XTRA) {
#pragma unroll
                for (int m = 0; m < NX; m++) da[m] = *reinterpret_cast<const lds_f32x4*>(xrow(m) + 16 * ch);
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) {
                    db[0][s2] = ld_w<STREAM>(&wpxA[(size_t)ch * wstep + 4 * s2]);
                    db[1][s2] = ld_w<STREAM>(&wpxB[(size_t)ch * wstep + 4 * s2]);
                }
            }
        };
        auto macX = [&](const f32x4 (&da)[NX], const float (&db)[2][4]) {
            if (XTRA) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
                    for (int m = 0; m < NX; m++) {
                        if (XMODE == 2) {
                            accxA[m] = mfma4(da[m][s2], db[0][s2], accxA[m]);
                            accxB[m] = mfma4(da[m][s2], db[1][s2], accxB[m]);
                        } else {
                            accxA[m] = mfma16(da[m][s2], db[0][s2], accxA[m]);
                            accxB[m] = mfma16(da[m][s2], db[1][s2], accxB[m]);
                        }
                    }
            }
        };
        auto loadA = [&](f32x4 (&dst)[MT], int ch) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) dst[mt] = *reinterpret_cast<const lds_f32x4*>(arow(mt) + 16 * ch);
        };
        auto loadB = [&](float (&dst)[2][NOWN][4], int ch) {
#pragma unroll
            for (int i = 0; i < NOWN; i++)
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    dst[0][i][s] = ld_w<STREAM>(&wpA[(size_t)ch * wstep + i * tst + 4 * s]);
                    dst[1][i][s] = ld_w<STREAM>(&wpB[(size_t)ch * wstep + i * tst + 4 * s]);
                }
        };
        auto mac = [&](const f32x4 (&a)[MT], const float (&b)[2][NOWN][4]) {
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (T4 && mt == MT - 1) {
                            tqA[i] = mfma4(a[mt][s], b[0][i][s], tqA[i]);
                            tqB[i] = mfma4(a[mt][s], b[1][i][s], tqB[i]);
                        } else {
                            accA[mt][i] = mfma16(a[mt][s], b[0][i][s], accA[mt][i]);
                            accB[mt][i] = mfma16(a[mt][s], b[1][i][s], accB[mt][i]);
                        }
                    }
        };
        loadB(b0, 0);
        loadA(a0, 0);
        loadX(ax0, bx0, 0);
        int ch = 0;
        auto pin = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x020, 8 * NOWN + (XTRA ? 8 : 0), 0);      // VMEM read
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * NOWN, 0);      // MFMA
            }
            if (XTRA) {
                __builtin_amdgcn_sched_group_barrier(0x100, NX, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * NX, 0);
            }
        };
        for (; ch + 2 <= KB; ch += 2) {
            loadB(b1, ch + 1);
            loadA(a1, ch + 1);
            loadX(ax1, bx1, ch + 1);
            mac(a0, b0);
            macX(ax0, bx0);
            pin();
            const int nx = ch + 2 < KB ? ch + 2 : KB - 1;
            loadB(b0, nx);
            loadA(a0, nx);
            loadX(ax0, bx0, nx);
            mac(a1, b1);
            macX(ax1, bx1);
            pin();
        }
        if (ch < KB) { mac(a0, b0); macX(ax0, bx0); }
        if (tail8) {
            const int kofs = 16 * KB + 2 * g - 4 * g;              // relative to the chunk pointers (which carry + 4g)
            const int tofs = ((((c >> 2) << 4) + 2 * g) << 2) + (c & 3);
            const size_t tb = ((size_t)tile0() << 8) + (size_t)KB * wstep + tofs;
            float bt[2][NOWN][2];
#pragma unroll
            for (int i = 0; i < NOWN; i++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    bt[0][i][s2] = WA[tb + i * tst + 4 * s2];
                    bt[1][i][s2] = WB[tb + i * tst + 4 * s2];
                }
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) av[mt] = *reinterpret_cast<const RLC_LDS f32x2*>(arow(mt) + kofs);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (T4 && mt == MT - 1) {
                            tqA[i] = mfma4(av[mt][s2], bt[0][i][s2], tqA[i]);
                            tqB[i] = mfma4(av[mt][s2], bt[1][i][s2], tqB[i]);
                        } else {
                            accA[mt][i] = mfma16(av[mt][s2], bt[0][i][s2], accA[mt][i]);
                            accB[mt][i] = mfma16(av[mt][s2], bt[1][i][s2], accB[mt][i]);
                        }
                    }
            if (XTRA) {
                const size_t txb = ((size_t)xt << 8) + (size_t)KB * wstep + tofs;
#pragma unroll
                for (int m = 0; m < NX; m++) {
                    const f32x2 avx = *reinterpret_cast<const RLC_LDS f32x2*>(xrow(m) + kofs);
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        if (XMODE == 2) {
                            accxA[m] = mfma4(avx[s2], WA[txb + 4 * s2], accxA[m]);
                            accxB[m] = mfma4(avx[s2], WB[txb + 4 * s2], accxB[m]);
                        } else {
                            accxA[m] = mfma16(avx[s2], WA[txb + 4 * s2], accxA[m]);
                            accxB[m] = mfma16(avx[s2], WB[txb + 4 * s2], accxB[m]);
                        }
                    }
                }
            }
        }
        if constexpr (T4) {
#pragma unroll
            for (int i = 0; i < NOWN; i++) {
                accA[MT - 1][i] += tail_finish(tqA[i]);
                accB[MT - 1][i] += tail_finish(tqB[i]);
            }
            if constexpr (XMODE == 2) { accxA[0] = tail_finish(accxA[0]); accxB[0] = tail_finish(accxB[0]); }
        }
    }

    template <bool STREAM = false>
    __device__ __forceinline__ void fwd_gemm2(f32x4 (&accA)[MT][NTW], f32x4 (&accB)[MT][NTW], const float* WA, const float* WB,
                                              int N, int K) {
        const int NT = (N + 15) >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) { accA[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if constexpr (ablate(6)) return;
        const int nown = nown_of(NT);
        const bool tail8 = (K & 15) == 8 && K > 16;
        const int KB = tail8 ? K >> 4 : (K + 15) >> 4;
        f32x4 accxA[MXS], accxB[MXS];
        if constexpr (NTW >= 2) {
            if (split_mode(NT)) {
#pragma unroll
                for (int m = 0; m < MXS; m++) { accxA[m] = f32x4{0.f, 0.f, 0.f, 0.f}; accxB[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                if (T4 && wave == 7) fwd_loop2<1, T4 ? 2 : 1, STREAM>(accA, accB, WA, WB, NT, KB, tail8, accxA, accxB, NT - 1, share_lo(3));
                else if (wave >= 4) fwd_loop2<1, 1, STREAM>(accA, accB, WA, WB, NT, KB, tail8, accxA, accxB, NT - 1, share_lo(wave - 4));
                else fwd_loop2<2, 0, STREAM>(accA, accB, WA, WB, NT, KB, tail8, accxA, accxB, 0, 0);
                collect_split<false>(accA, accxA);
                __syncthreads();                 // wave 4 has taken the first set out of the hand-off buffer
                collect_split<false>(accB, accxB);
            } else if (nown >= 2) fwd_loop2<2, 0, STREAM>(accA, accB, WA, WB, NT, KB, tail8, accxA, accxB, 0, 0);
            else if (nown == 1) fwd_loop2<1, 0, STREAM>(accA, accB, WA, WB, NT, KB, tail8, accxA, accxB, 0, 0);
        } else {
            if (nown >= 1) fwd_loop2<1, 0, STREAM>(accA, accB, WA, WB, NT, KB, tail8, accxA, accxB, 0, 0);
        }
    }

    // acc = relu(acc + bias[n] + sum_j E[b][j] * Wx[xrow0+j][n])        (E = extra input columns of a concat layer)
    template <int NE>
    __device__ __forceinline__ void bias_relu(f32x4 (&acc)[MT][NTW], const float* bias, int N, const lds_f32* E = nullptr,
                                              const float* Wx = nullptr /* tile-blocked matrix whose rows xrow0+j multiply E */,
                                              int xrow0 = 0) {
        if constexpr (ablate(10)) return;
        const int NT = (N + 15) >> 4;
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = tile_of(i);
            const int n = 16 * t + c;
            const bool ok = t < NT && n < N;
            const float bs = ok ? bias[n] : 0.0f;
            float wx[NE > 0 ? NE : 1];
#pragma unroll
            for (int j = 0; j < NE; j++) wx[j] = ok ? Wx[rlc_blk_index(xrow0 + j, n, N)] : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[mt][i][r] + bs;
                    if (NE > 0) {
                        const int b = 16 * mt + 4 * g + r;
#pragma unroll
                        for (int j = 0; j < NE; j++) v += E[b * NE + j] * wx[j];
                    }
                    acc[mt][i][r] = ok ? fmaxf(v, 0.0f) : 0.0f;
                }
        }
    }

    // part[wave][b][j] = sum over this wave's columns n of f(acc[b][n]) * cf(n, j); f = identity or step.
    // cf is only evaluated for valid columns.
    template <bool STEP, int NJ, class CF>
    __device__ __forceinline__ void row_dot(const f32x4 (&acc)[MT][NTW], int N, CF cf, lds_f32* part) {
        if constexpr (ablate(8)) return;
        const int NT = (N + 15) >> 4;
        float co[NTW][NJ];
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = tile_of(i);
            const int n = 16 * t + c;
            const bool ok = t < NT && n < N;
#pragma unroll
            for (int j = 0; j < NJ; j++) co[i][j] = ok ? cf(n, j) : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    float p = 0.0f;
#pragma unroll
                    for (int i = 0; i < NTW; i++) {
                        const float v = acc[mt][i][r];
                        p += STEP ? (v > 0.0f ? co[i][j] : 0.0f) : v * co[i][j];
                    }
                    p = row16_sum(p);
                    if (c == 0) part[((size_t)wave * MB + 16 * mt + 4 * g + r) * NJ + j] = p;
                }
    }

    // fixed-order sum of the waves' partials
    template <int NJ>
    __device__ __forceinline__ float part_sum(const lds_f32* part, int b, int j) const {
        float s = part[((size_t)0 * MB + b) * NJ + j];
#pragma unroll
        for (int w = 1; w < kWaves; w++) s += part[((size_t)w * MB + b) * NJ + j];
        return s;
    }

    // Heads of a concat layer evaluated at a second set of extra inputs WITHOUT touching the accumulators:
    //   v = relu(acc + bias[n] + sum_j E[b][j] Wx[xrow0+j][n]);  part[wave][b][0]   = sum_n v * w3[n]   (the layer's scalar head)
    //                                                            part[wave][b][1+j] = sum_n step(v) * w3[n] * Wx[xrow0+j][n]
    // (d head / d E[b][j]): Q(s, pi) and dQ/da of a critic whose hidden contraction is shared with Q(s, a).
    template <int NE>
    __device__ __forceinline__ void concat_head_dots(const f32x4 (&acc)[MT][NTW], const float* bias, int N, const lds_f32* E,
                                                     const float* Wx, int xrow0, const float* w3, lds_f32* part) {
        const int NT = (N + 15) >> 4;
        float bs[NTW], w3n[NTW], wx[NTW][NE];
        bool okk[NTW];
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = tile_of(i);
            const int n = 16 * t + c;
            const bool ok = t < NT && n < N;
            okk[i] = ok;
            bs[i] = ok ? bias[n] : 0.0f;
            w3n[i] = ok ? w3[n] : 0.0f;
#pragma unroll
            for (int j = 0; j < NE; j++) wx[i][j] = ok ? Wx[rlc_blk_index(xrow0 + j, n, N)] : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int b = 16 * mt + 4 * g + r;
                float e[NE];
#pragma unroll
                for (int j = 0; j < NE; j++) e[j] = E[b * NE + j];
                float p0 = 0.0f, pj[NE];
#pragma unroll
                for (int j = 0; j < NE; j++) pj[j] = 0.0f;
#pragma unroll
                for (int i = 0; i < NTW; i++) {
                    float v = acc[mt][i][r] + bs[i];
#pragma unroll
                    for (int j = 0; j < NE; j++) v += e[j] * wx[i][j];
                    v = okk[i] ? fmaxf(v, 0.0f) : 0.0f;
                    p0 += v * w3n[i];
#pragma unroll
                    for (int j = 0; j < NE; j++) pj[j] += v > 0.0f ? w3n[i] * wx[i][j] : 0.0f;
                }
                p0 = row16_sum(p0);
#pragma unroll
                for (int j = 0; j < NE; j++) pj[j] = row16_sum(pj[j]);
                if (c == 0) {
                    lds_f32* dst = part + ((size_t)wave * MB + b) * (1 + NE);
                    dst[0] = p0;
#pragma unroll
                    for (int j = 0; j < NE; j++) dst[1 + j] = pj[j];
                }
            }
    }

    // relu masks of the accumulators -> bit plane BIT of one byte per (row, unit).  OVERWRITE: the byte becomes
    // the mask of this plane alone (other planes cleared); otherwise the plane is OR-ed in (the same thread owns
    // the same byte for every plane: no race).  BIT == -2 (single plane, OVERWRITE): the byte is 0x38 = 1.0 in OCP
    // fp8 e4m3, which the backward GEMM turns into floats two at a time (v_cvt_pk_f32_fp8).
    template <int BIT, bool OVERWRITE>
    __device__ __forceinline__ void store_masks(const f32x4 (&acc)[MT][NTW], int N) {
        if constexpr (ablate(9)) return;
        const int NT = (N + 15) >> 4;
#ifdef RLC_PACKED_MASKS
        // (opt-in: measured +-0 on DDPG, -1.4 % on SoftActorCritic, profiles/r03_variant_timings_s6.txt)
        // One dword store per lane and tile instead of four byte stores (which land four to a bank).  A lane holds the
        // flags of rows 4g..4g+3 at unit c; a mask dword is four units of one row: the four lanes of a quad (units
        // 4q..4q+3, same rows) transpose their 4 x 4 bytes with quad-broadcast DPP moves and byte permutes, and lane j of
        // the quad stores row 4g + j.  (Row stride 4 * odd dwords: the 64 stores of an instruction hit 64 banks.)
        constexpr unsigned code = BIT == -2 ? 0x38u : 1u << (BIT < 0 ? 0 : BIT);
        const int r0 = c & 3;
        const unsigned sel = (unsigned)r0 | ((unsigned)(4 + r0) << 8) | 0x0c0c0000u;     // bytes: lo[r0], hi[r0], 0, 0
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = tile_of(i);
            if (t < NT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    unsigned w = 0;
#pragma unroll
                    for (int r = 0; r < 4; r++) w |= acc[mt][i][r] > 0.0f ? code << (8 * r) : 0u;
                    const unsigned w0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x00, 0xf, 0xf, false);
                    const unsigned w1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x55, 0xf, 0xf, false);
                    const unsigned w2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xaa, 0xf, 0xf, false);
                    const unsigned w3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xff, 0xf, 0xf, false);
                    const unsigned lo = __builtin_amdgcn_perm(w1, w0, sel), hi = __builtin_amdgcn_perm(w3, w2, sel);
                    const unsigned out = lo | (hi << 16);
                    lds_u32* p = reinterpret_cast<lds_u32*>(&L.mask[(16 * mt + 4 * g + r0) * MSTRIDE + 16 * t + (c & 12)]);
                    if (OVERWRITE) *p = out;
                    else *p = (*p & ~(0x01010101u * code)) | out;
                }
            }
        }
        return;
#endif
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = tile_of(i);
            if (t < NT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        lds_u8* p = &L.mask[(16 * mt + 4 * g + r) * MSTRIDE + 16 * t + c];
                        const unsigned char bit = acc[mt][i][r] > 0.0f ? (unsigned char)(BIT == -2 ? 0x38u : 1u << (BIT < 0 ? 0 : BIT)) : (unsigned char)0;
                        if (OVERWRITE) *p = bit;
                        else *p = (unsigned char)((*p & ~(1u << (BIT < 0 ? 0 : BIT))) | bit);
                    }
            }
        }
    }

    // four mask bytes -> four floats (0.0 / 1.0).  BIT == -2: fp8-coded bytes, two per v_cvt_pk_f32_fp8; otherwise 0/1
    // bytes through v_cvt_f32_ubyte<s>
    template <int BITV>
    __device__ __forceinline__ static void mask4(unsigned mw, float (&f)[4], std::integral_constant<int, BITV>) {
        if constexpr (BITV == -2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)mw, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)mw, true);
            f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) f[s] = (float)((mw >> (8 * s)) & 0xffu);
        }
    }

    // ---------------------------------------------------------------------------------------
    // backward-to-input GEMM: acc[b][k'] (+)= sum_n D[b][n] * W[k'][n],  D[b][n] = mask(b,n) * sum_j seed[b][j]*wv[j][n]
    // (D is never materialised).  k-dim = n in chunks of 16 with lane (c,g) taking n = nc+4g+s:
    //   B = the lane's 16 bytes of block (t, nc/16) of the tile-blocked W: 1 KB contiguous per instruction;
    //   A = the relu mask bytes of row 16mt+c (one ds_read_b32 -> four v_cvt_f32_ubyte).
    // NS == 1 without accumulation: D is rank one, so the seed leaves the loop --
    //   acc[b][k'] = seed[b] * sum_n maskf(b,n) * (wv[n] W[k'][n]):  A = the 0/1 mask floats as they are, B is
    //   scaled by wv (4 multiplies per tile per chunk) and the rows are scaled by seed[b] once at the end.
    // BIT < 0: the mask bytes are 0/1 as stored (single plane written with BIT 0 + OVERWRITE); BIT >= 0 selects a plane.
    // Same structure as fwd_loop: tiles owned is a template parameter, two register sets, no masks.
    // ---------------------------------------------------------------------------------------
    // XMODE (see share_lo): besides its NOWN output tiles the wave computes its share of the split output tile xt
    template <int NS, int NOWN, int BIT, bool TRICK, int XMODE>
    __device__ __forceinline__ void bwd_loop(f32x4 (&acc)[MT][NTW], const float* W, int NTk, const lds_f32* seed,
                                             const lds_f32* wvec, bool tail8, int NTblk, f32x4 (&accx)[MXS], int xt, int xm0) {
        constexpr bool XTRA = XMODE != 0;
        constexpr int NX = XMODE == 2 ? 1 : MXS;
        const float* wp = W + (((size_t)tile0() * NTblk) << 8) + (lane << 2);
        const size_t tst = ((size_t)tstep() * NTblk) << 8;
        // batch row whose mask / seed this lane feeds as the A operand of batch tile mt (T4: the tail's rows c & 3)
        auto brow = [&](int mt) { return (T4 && mt == MT - 1) ? TROW + (c & 3) : 16 * mt + c; };
        const lds_u8* mp = L.mask + 4 * g;
        const lds_f32* wvp = wvec + 4 * g;
        float sd[MT][NS];
        if (!TRICK) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int j = 0; j < NS; j++) sd[mt][j] = seed[brow(mt) * NS + j];
        }
        f32x4 tq[NOWN];                                             // T4: the tail tile's 4x4x1 accumulators
#pragma unroll
        for (int i = 0; i < NOWN; i++) tq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the split tile's share: its weight rows, and the mask / seed rows of its batch tiles (a run-time range)
        const float* wpx = W + (((size_t)xt * NTblk) << 8) + (lane << 2);
        int xrow[NX];
        float sdx[NX][NS];
        if (XTRA) {
#pragma unroll
            for (int m = 0; m < NX; m++) {
                xrow[m] = XMODE == 2 ? TROW + (c & 3) : 16 * (xm0 + m < MT ? xm0 + m : xm0) + c;
#pragma unroll
                for (int j = 0; j < NS; j++) sdx[m][j] = seed[xrow[m] * NS + j];
            }
        }
        f32x4 b0[NOWN], b1[NOWN], bx0, bx1;
        auto loadB = [&](f32x4 (&dst)[NOWN], f32x4& dx, int ch) {
#pragma unroll
            for (int i = 0; i < NOWN; i++)
                dst[i] = *reinterpret_cast<const f32x4*>(wp + i * tst + ((size_t)ch << 8));
            if (XTRA) dx = *reinterpret_cast<const f32x4*>(wpx + ((size_t)ch << 8));
        };
        // RLC_BWD_PREFETCH: the raw mask dwords of the NEXT chunk are read before the MFMAs of this one (MT registers)
        unsigned mwn[MT];
        auto load_masks = [&](int ch) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) mwn[mt] = *reinterpret_cast<const lds_u32*>(mp + brow(mt) * MSTRIDE + 16 * ch);
        };
        auto mac = [&](const f32x4 (&bin)[NOWN], const f32x4& binx, int ch) {
            f32x4 wv[NS], b[NOWN], bx;
#pragma unroll
            for (int j = 0; j < NS; j++) wv[j] = *reinterpret_cast<const lds_f32x4*>(wvp + j * 256 + 16 * ch);
#pragma unroll
            for (int i = 0; i < NOWN; i++) b[i] = TRICK ? bin[i] * wv[0] : bin[i];
            if (XTRA) bx = TRICK ? binx * wv[0] : binx;
            f32x4 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#ifdef RLC_BWD_PREFETCH
                unsigned mw = mwn[mt];
#else
                unsigned mw = *reinterpret_cast<const lds_u32*>(mp + brow(mt) * MSTRIDE + 16 * ch);
#endif
                if (BIT >= 0) mw = (mw >> (BIT >= 0 ? BIT : 0)) & 0x01010101u;
                float mf[4];
                mask4(mw, mf, std::integral_constant<int, BIT>{});
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float f = mf[s];                                   // 0.0 or 1.0
                    if (TRICK) {
                        av[mt][s] = f;
                    } else {
                        float v = 0.0f;
#pragma unroll
                        for (int j = 0; j < NS; j++) v += sd[mt][j] * wv[j][s];
                        av[mt][s] = f * v;
                    }
                }
            }
#ifdef RLC_BWD_PREFETCH
            load_masks(ch + 1 < NTk ? ch + 1 : ch);       // consumed by the next call (clamped on the last chunk)
#endif
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (T4 && mt == MT - 1) tq[i] = mfma4(av[mt][s], b[i][s], tq[i]);
                        else acc[mt][i] = mfma16(av[mt][s], b[i][s], acc[mt][i]);
                    }
            if (XTRA) {
#pragma unroll
                for (int m = 0; m < NX; m++) {
                    unsigned mw = *reinterpret_cast<const lds_u32*>(L.mask + xrow[m] * MSTRIDE + 4 * g + 16 * ch);
                    if (BIT >= 0) mw = (mw >> (BIT >= 0 ? BIT : 0)) & 0x01010101u;
                    float mf[4];
                    mask4(mw, mf, std::integral_constant<int, BIT>{});
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        float f = mf[s];
                        if (!TRICK) {
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < NS; j++) v += sdx[m][j] * wv[j][s];
                            f *= v;
                        }
                        accx[m] = XMODE == 2 ? mfma4(f, bx[s], accx[m]) : mfma16(f, bx[s], accx[m]);
                    }
                }
            }
        };
        loadB(b0, bx0, 0);
#ifdef RLC_BWD_PREFETCH
        load_masks(0);
#endif
        int ch = 0;
        // one scheduling region per pair of chunks (see fwd_loop): next chunk's weight tile first, then per M tile
        // its mask dword read ahead of the 4*NOWN MFMAs that consume the previous one
        auto pin = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x020, NOWN + (XTRA ? 1 : 0), 0);              // VMEM read
            __builtin_amdgcn_sched_group_barrier(0x100, NS + MT + (XTRA ? NX : 0), 0);          // DS read: wvec rows + every mask dword
        };
        for (; ch + 2 <= NTk; ch += 2) {
            loadB(b1, bx1, ch + 1);
            mac(b0, bx0, ch);
            pin();
            loadB(b0, bx0, ch + 2 < NTk ? ch + 2 : NTk - 1);
            mac(b1, bx1, ch + 1);
            pin();
        }
        if (ch < NTk) mac(b0, bx0, ch);
        if (tail8) {
            // row length = 16 NTk + 8: the last 8 n's in TWO steps -- lane group g takes n = 16 NTk + 2g + s: an 8-byte
            // weight load (columns 2g, 2g+1 of the lane's row of the block), two mask bytes, two wvec entries per row
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int tofs = ((((g >> 1) << 4) + c) << 2) + 2 * (g & 1);
            const float* wt = W + (((size_t)tile0() * NTblk + NTk) << 8) + tofs;
            f32x2 bt[NOWN], wv[NS];
#pragma unroll
            for (int i = 0; i < NOWN; i++) bt[i] = *reinterpret_cast<const f32x2*>(wt + i * tst);
#pragma unroll
            for (int j = 0; j < NS; j++) wv[j] = *reinterpret_cast<const RLC_LDS f32x2*>(wvec + j * 256 + 16 * NTk + 2 * g);
            if (TRICK) {
#pragma unroll
                for (int i = 0; i < NOWN; i++) bt[i] = bt[i] * wv[0];
            }
            f32x2 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                unsigned mw = *reinterpret_cast<const RLC_LDS unsigned short*>(L.mask + brow(mt) * MSTRIDE + 16 * NTk + 2 * g);
                if (BIT >= 0) mw = (mw >> (BIT >= 0 ? BIT : 0)) & 0x0101u;
                float mf[4];
                mask4(mw, mf, std::integral_constant<int, BIT>{});
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const float f = mf[s2];
                    if (TRICK) {
                        av[mt][s2] = f;
                    } else {
                        float v = 0.0f;
#pragma unroll
                        for (int j = 0; j < NS; j++) v += sd[mt][j] * wv[j][s2];
                        av[mt][s2] = f * v;
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        if (T4 && mt == MT - 1) tq[i] = mfma4(av[mt][s2], bt[i][s2], tq[i]);
                        else acc[mt][i] = mfma16(av[mt][s2], bt[i][s2], acc[mt][i]);
                    }
            if (XTRA) {
                f32x2 btx = *reinterpret_cast<const f32x2*>(W + (((size_t)xt * NTblk + NTk) << 8) + tofs);
                if (TRICK) btx = btx * wv[0];
#pragma unroll
                for (int m = 0; m < NX; m++) {
                    unsigned mw = *reinterpret_cast<const RLC_LDS unsigned short*>(L.mask + xrow[m] * MSTRIDE + 16 * NTk + 2 * g);
                    if (BIT >= 0) mw = (mw >> (BIT >= 0 ? BIT : 0)) & 0x0101u;
                    float mf[4];
                    mask4(mw, mf, std::integral_constant<int, BIT>{});
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        float f = mf[s2];
                        if (!TRICK) {
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < NS; j++) v += sdx[m][j] * wv[j][s2];
                            f *= v;
                        }
                        accx[m] = XMODE == 2 ? mfma4(f, btx[s2], accx[m]) : mfma16(f, btx[s2], accx[m]);
                    }
                }
            }
        }
        if constexpr (T4) {
#pragma unroll
            for (int i = 0; i < NOWN; i++) acc[MT - 1][i] += tail_finish(tq[i]);
            if constexpr (XMODE == 2) accx[0] = tail_finish(accx[0]);
        }
        if (TRICK) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const f32x4 sv = *reinterpret_cast<const lds_f32x4*>(&seed[16 * mt + 4 * g]);
#pragma unroll
                for (int i = 0; i < NOWN; i++) acc[mt][i] = acc[mt][i] * sv;
            }
            if (XTRA) {
#pragma unroll
                for (int m = 0; m < NX; m++)
                    accx[m] = accx[m] * *reinterpret_cast<const lds_f32x4*>(
                                            &seed[(XMODE == 2 ? TROW : 16 * (xm0 + m < MT ? xm0 + m : xm0)) + 4 * g]);
            }
        }
    }

    // ACCUM: add onto the accumulators (a second branch that feeds the same input), else start from zero
    template <int NS, int BIT = -1, bool ACCUM = false>
    __device__ __forceinline__ void bwd_gemm(f32x4 (&acc)[MT][NTW], const float* W, int Nk /* row length = k-dim */,
                                             int Kout /* rows of W used */, const lds_f32* seed /* LDS [MB][NS] */,
                                             const lds_f32* wvec /* LDS [NS][256] */) {
        const int NT = (Kout + 15) >> 4;
        if (!ACCUM) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < NTW; i++) acc[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (ablate(5)) return;
        const int nown = nown_of(NT);
        const int NTblk = (Nk + 15) >> 4;          // blocks per row of the tile-blocked W
        const bool tail8 = (Nk & 15) == 8 && Nk > 16;
        const int NTk = tail8 ? Nk >> 4 : NTblk;
        constexpr bool TRICK = NS == 1 && !ACCUM;
        f32x4 accx[MXS];
#ifdef RLC_STAMPS
        const long long t_w0 = clock64();
#endif
        if constexpr (NTW >= 2) {
            if (split_mode(NT)) {                  // workgroup-uniform
#pragma unroll
                for (int m = 0; m < MXS; m++) accx[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (T4 && wave == 7) bwd_loop<NS, 1, BIT, TRICK, T4 ? 2 : 1>(acc, W, NTk, seed, wvec, tail8, NTblk, accx, NT - 1, share_lo(3));
                else if (wave >= 4) bwd_loop<NS, 1, BIT, TRICK, 1>(acc, W, NTk, seed, wvec, tail8, NTblk, accx, NT - 1, share_lo(wave - 4));
                else bwd_loop<NS, 2, BIT, TRICK, 0>(acc, W, NTk, seed, wvec, tail8, NTblk, accx, 0, 0);
                collect_split<ACCUM>(acc, accx);
            } else if (nown >= 2) bwd_loop<NS, 2, BIT, TRICK, 0>(acc, W, NTk, seed, wvec, tail8, NTblk, accx, 0, 0);
            else if (nown == 1) bwd_loop<NS, 1, BIT, TRICK, 0>(acc, W, NTk, seed, wvec, tail8, NTblk, accx, 0, 0);
        } else {
            if (nown >= 1) bwd_loop<NS, 1, BIT, TRICK, 0>(acc, W, NTk, seed, wvec, tail8, NTblk, accx, 0, 0);
        }
#ifdef RLC_STAMPS
        if (lane == 0 && stamp_buf) stamp_buf[56 + wave] += (float)(clock64() - t_w0);   // per-wave k-loop cycles
#endif
    }

    // epilogue of bwd_gemm: dh1 = (acc [+ extra(b, k)]) * (hbuf > 0); column-reduce into the W1 / b1 gradients of
    // this wave's first-layer units and apply Adam (+ optional Polyak) right here.  xs = the layer's input rows
    // (LDS [MB][SMAX]).  extra(b, k): further contributions to dL/dh1[b][k] (heads that hang off the first layer).
    // GONLY: only the gradient is produced (written to `tap`, which must not be null); no Adam, no Polyak -- the
    // batch-split kernel (ddpg_split_kernel.h) reduces such partial gradients over the CUs of an agent first.
    // stage (or null): the LDS copy of this first layer (stage_store layout) that receives the stepped values too
    template <class EXTRA = NoExtra, bool GONLY = false>
    __device__ __forceinline__ void trunk_grad_adam(const f32x4 (&acc)[MT][NTW], float* th, float* m, float* v,
                                                    float alpha, int oW1, int ob1, float* tap, float* tt, float tau,
                                                    const lds_f32* xs, EXTRA extra = EXTRA{}, lds_f32* stage = nullptr) {
        if constexpr (ablate(7)) return;
        if (S <= 4) trunk_grad_adam_t<4, EXTRA, GONLY>(acc, th, m, v, alpha, oW1, ob1, tap, tt, tau, xs, extra, stage);     // wave-uniform
        else trunk_grad_adam_t<SMAX, EXTRA, GONLY>(acc, th, m, v, alpha, oW1, ob1, tap, tt, tau, xs, extra, stage);
    }
    template <int SP, class EXTRA, bool GONLY = false>
    __device__ __forceinline__ void trunk_grad_adam_t(const f32x4 (&acc)[MT][NTW], float* th, float* m, float* v,
                                                      float alpha, int oW1, int ob1, float* tap, float* tt, float tau,
                                                      const lds_f32* xs, EXTRA extra, lds_f32* stage) {
        const int NT = (H1 + 15) >> 4;
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = tile_of(i);
            if (t >= NT) continue;
            const int k = 16 * t + c;
            float gb = 0.0f;
            float gw[SP];
#pragma unroll
            for (int s = 0; s < SP; s++) gw[s] = 0.0f;
            HeadExtra::f4 hcolw = {0.f, 0.f, 0.f, 0.f};
            if constexpr (std::is_same<EXTRA, HeadExtra>::value) hcolw = extra.column(k < H1 ? k : 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                // with a head term per element (NAF) the scheduler otherwise hoists every LDS read of the unrolled loop to
                // the front and spills the accumulators it still needs (550 scratch reloads in this function alone)
                if constexpr (!std::is_same<EXTRA, NoExtra>::value && !std::is_same<EXTRA, HeadExtra>::value)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int b = 16 * mt + 4 * g + r;
                    const float hv = L.hbuf[b * LDH + (k < H1 ? k : 0)];
                    float d;
                    if constexpr (std::is_same<EXTRA, NoExtra>::value) {
                        d = (k < H1 && hv > 0.0f) ? acc[mt][i][r] : 0.0f;
                    } else if constexpr (std::is_same<EXTRA, HeadExtra>::value) {
                        const float e = extra.at(hcolw, b);
                        d = (k < H1 && hv > 0.0f) ? acc[mt][i][r] + e : 0.0f;
                    } else {
                        d = (k < H1 && hv > 0.0f) ? acc[mt][i][r] + extra(b, k < H1 ? k : 0) : 0.0f;
                    }
                    gb += d;
                    const f32x4 x0 = *reinterpret_cast<const lds_f32x4*>(&xs[b * SMAX]);
#pragma unroll
                    for (int s = 0; s < 4; s++) gw[s] += x0[s] * d;
                    if (SP > 4) {
                        const f32x4 x1 = *reinterpret_cast<const lds_f32x4*>(&xs[b * SMAX + 4]);
#pragma unroll
                        for (int s = 0; s < 4; s++) gw[(SP > 4 ? 4 : 0) + s] += x1[s] * d;
                    }
                }
            }
            gb = col4_sum(gb);
#pragma unroll
            for (int s = 0; s < SP; s++) gw[s] = col4_sum(gw[s]);
            // lanes g == s' handle row s' (spread the Adam work over the 4 lane groups)
            if (k < H1) {
                for (int s = g; s <= S; s += 4) {
                    const bool is_bias = s == S;
                    float gr = gb;
#pragma unroll
                    for (int q = 0; q < SP; q++)
                        if (q == s && !is_bias) gr = gw[q];
                    const int p = is_bias ? ob1 + k : oW1 + s * H1 + k;
                    if constexpr (GONLY) {
                        tap[p] = gr;
                    } else {
                        float mm = m[p], vv = v[p];
                        const float o = tt ? tt[p] : 0.0f;          // with the other loads: one memory round trip, not two
                        const float nv = astep_small(th[p], gr, mm, vv, alpha);
                        m[p] = mm; v[p] = vv; th[p] = nv;
                        if (stage) stage[is_bias ? S * H1 + k : s * H1 + k] = nv;
                        if (tap) tap[p] = gr;
                        if (tt) tt[p] = polyak(o, nv, tau);
                    }
                }
            }
        }
    }

    // ---------------------------------------------------------------------------------------
    // weight-gradient GEMM + Adam (+Polyak) epilogue:
    //   G[k'][n] = sum_b X[b][k'] * D[b][n],  X = [hbuf | E] (E = NE extra input columns, or none), D as above.
    // TRANSPOSED tiles: acc[q][r] = G[k' = 16(m0+q) + c][n = 16t + 4g + r] (D^T on the A side, hbuf on the B
    // side), so each lane owns 4 CONSECUTIVE n of one weight row = its 16 bytes of block (m0+q, t) of the
    // tile-blocked arrays: the W / m / v / W' traffic of the Adam epilogue is one 1 KB-contiguous load and one
    // store per array per tile.  k-dim = batch, lane group g takes b = 4*ks + {0,2,1,3}[g] (conflict-free reads).
    //
    // Work items = (N tile t, chunk of <= 4 M' tiles); the items of a matrix are dealt
    // round-robin to the 8 waves (every output tile is independent: no cross-wave reduction), so all four SIMDs
    // carry the same MFMA load.  While the k-loop of one item runs, the W / m / v / W' of the wave's NEXT item are
    // already in flight into a second register set.
    // ---------------------------------------------------------------------------------------
    struct WgPre { f32x4 w[4], m[4], v[4], t[4]; };
    // the first two items of this wave (idx = wave, wave + 8) in flight before wgrad_adam is entered: wgrad_prefetch
    // issues them ahead of the backward GEMM that precedes the weight-gradient phase, so that their HBM latency hides
    // under that GEMM instead of under one k-loop (a wave has only two items per matrix at widths <= 128)
    struct WgPre2 { WgPre a, b; };

    // item idx of a [H1 x N] matrix -> N tile t, first M' tile m0, tiles in the chunk nq
    __device__ __forceinline__ void wg_item_geom(int idx, int N, int& t, int& m0, int& nq) const {
        const int NT = (N + 15) >> 4, NMT = (H1 + 15) >> 4;
        const int nch = (NMT + 3) >> 2, cbase = NMT / nch, crem = NMT % nch;
        t = idx % NT;
        const int ch = idx / NT;
        nq = cbase + (ch < crem ? 1 : 0);
        m0 = ch * cbase + (ch < crem ? ch : crem);
    }
    __device__ __forceinline__ int wg_nitems(int N) const {
        const int NT = (N + 15) >> 4, NMT = (H1 + 15) >> 4;
        return NT * ((NMT + 3) >> 2);
    }
    // Prefetch an item's W / m / v / W' NOW (addresses clamped, stores predicated).
    template <bool NOPOL>
    __device__ __forceinline__ void wg_issue(WgPre& P, int idx, int N, const float* Wp, const float* mp, const float* vp,
                                             const float* Wt, float alpha) const {
        int t, m0, nq;
        wg_item_geom(idx, N, t, m0, nq);
        const int NT = (N + 15) >> 4;
        const int lane4 = (g * 16 + c) << 2;
        const bool n4ok = 16 * t + 4 * g < N;        // N % 4 == 0: all four columns valid or none
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kp = 16 * (m0 + q) + c;
            const size_t p = (q < nq && kp < H1 && n4ok) ? ((((size_t)(m0 + q) * NT + t) << 8) + lane4) : 0;
            if constexpr (ablate(12)) {
                P.w[q] = P.m[q] = P.v[q] = P.t[q] = f32x4{0.5f, 0.25f, 0.125f, 1.0f} * alpha;
                continue;
            }
            P.w[q] = *reinterpret_cast<const f32x4*>(&Wp[p]);
            P.m[q] = ld_stream(&mp[p]);
            P.v[q] = ld_stream(&vp[p]);
            if constexpr (!NOPOL) P.t[q] = ld_target(&Wt[p]);
        }
    }
    template <bool NOPOL = false, int NPRE = 2>
    __device__ __forceinline__ void wgrad_prefetch(WgPre2& pre, int N, const float* Wp, const float* mp, const float* vp,
                                                   const float* Wt) const {
        if constexpr (ablate(0) || ablate(1)) return;
        const int nitems = wg_nitems(N);
        asm volatile("" ::: "memory");
        if (wave < nitems) wg_issue<NOPOL>(pre.a, wave, N, Wp, mp, vp, Wt, 0.0f);
        if constexpr (NPRE > 1)
            if (wave + kWaves < nitems) wg_issue<NOPOL>(pre.b, wave + kWaves, N, Wp, mp, vp, Wt, 0.0f);
        asm volatile("" ::: "memory");
    }

    // GONLY: as in trunk_grad_adam -- the gradient tiles go to `tapp` (not null), nothing else is read or written.
    // NOPOL: the matrix has no target copy (Wt unused): no target loads, no Polyak stores.
    template <int NS, int NE, int BIT = -1, bool GONLY = false, bool NOPOL = false>
    __device__ __forceinline__ void wgrad_adam(const lds_f32* seed /* LDS [MB][NS] */, const lds_f32* E /* LDS [MB][NE] or null */,
                                               int N, float* Wp, float* mp, float* vp,
                                               float alpha, float* tapp, float* Wt, float tau,
                                               const lds_f32* wvec /* LDS [NS][256] */) {
        WgPre2 none;
        wgrad_adam_pre<NS, NE, BIT, GONLY, NOPOL, 0>(seed, E, N, Wp, mp, vp, alpha, tapp, Wt, tau, wvec, none);
    }
    // NPRE > 0: `pre` holds this wave's first NPRE items already in flight (wgrad_prefetch<NOPOL, NPRE>); by reference and
    // a compile-time count, so that the registers stay registers
    template <int NS, int NE, int BIT, bool GONLY, bool NOPOL, int NPRE>
    __device__ __forceinline__ void wgrad_adam_pre(const lds_f32* seed, const lds_f32* E, int N, float* Wp, float* mp, float* vp,
                                                   float alpha, float* tapp, float* Wt, float tau, const lds_f32* wvec,
                                                   WgPre2& pre) {
        if constexpr (ablate(0)) return;
#ifdef RLC_STAMPS
        const long long t_wg0 = clock64();
#endif
        const int NT = (N + 15) >> 4;
        const int NMT = (H1 + 15) >> 4;                  // MFMA rows: the hbuf units; extra rows below
        const int nch = (NMT + 3) >> 2, cbase = NMT / nch, crem = NMT % nch;    // chunk sizes differ by at most one
        const int nitems = NT * nch;
        const int gperm = ((g & 1) << 1) | (g >> 1);     // 0,2,1,3
        const int lane4 = (g * 16 + c) << 2;
        constexpr unsigned MBITS = BIT < 0 ? 0xffu : (1u << (BIT < 0 ? 0 : BIT));

        auto item_geom = [&](int idx, int& t, int& m0, int& nq) { wg_item_geom(idx, N, t, m0, nq); };
        // Prefetch an item's W / m / v / W' NOW: their HBM latency hides under the previous item's k-loop
        auto issue = [&](WgPre& P, int idx) {
            if constexpr (GONLY || ablate(1)) return;
            wg_issue<NOPOL>(P, idx, N, Wp, mp, vp, Wt, alpha);
        };
        // EXACT: the item has exactly MCC tiles (no aliased rows): the MCC activation reads of a k-step are then
        // base + q * 64 B with a compile-time stride and pair up as ds_read2_b32
        // RLC_WG_LATE_ISSUE (per translation unit): the OTHER register set is loaded for item `nxt` after this item's
        // k-loop instead of before it, and handed back by value -- a wave with two items per matrix (widths <= 128) then
        // issues its second prefetch when the first has landed instead of queueing both behind each other up front
#ifdef RLC_WG_LATE_ISSUE
#define RLC_RUN_RET return Q
        auto run = [&](const WgPre& P, int idx, int nxt, auto mcc_tag, auto exact_tag) {
#else
#define RLC_RUN_RET return
        auto run = [&](const WgPre& P, int idx, auto mcc_tag, auto exact_tag) {
#endif
            constexpr int MCC = decltype(mcc_tag)::value;
            constexpr bool EXACT = decltype(exact_tag)::value;
            int t, m0, nq;
            item_geom(idx, t, m0, nq);
            const int n = 16 * t + c;
            float wvn[NS];
#pragma unroll
            for (int j = 0; j < NS; j++) wvn[j] = n < N ? wvec[j * 256 + n] : 0.0f;
            f32x4 acc[MCC];
            int kq[MCC];
#pragma unroll
            for (int q = 0; q < MCC; q++) {
                acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                kq[q] = (EXACT || q < nq ? 16 * (m0 + q) : 0) + c;    // rows past the chunk alias tile 0 (never stored)
            }
            const lds_u8* mrow = L.mask + 16 * t + c;
            sub_stamp(28);               // (diagnostic build) since the last item's epilogue: next prefetch issued, item geometry
            // The batch is walked in MT groups of 4 k-steps (rows 16 gi + 4 j + gperm), fully unrolled, the operands
            // of the next group in flight while the current group's 4*MCC MFMAs issue.  LDC: compile-time leading
            // dimension of hbuf (every LDS address is then base + immediate), 0 = runtime.
            struct Ops { float hf[4][MCC]; float sd[4][NS]; unsigned mk[4]; };
            auto kloop = [&](auto ldc_tag) {
                constexpr int LDC = decltype(ldc_tag)::value;
                const int ld = LDC ? LDC : LDH;
                const lds_f32* hq[MCC];
#pragma unroll
                for (int q = 0; q < MCC; q++) hq[q] = EXACT ? L.hbuf + gperm * ld + kq[0] + 16 * q : L.hbuf + gperm * ld + kq[q];
                // a second set of bases 64 rows down: with a compile-time leading dimension every activation read is
                // then base + a 16-bit immediate (row * ld * 4 bytes passes 64 KB at row 82 of 112)
                const lds_f32* hq2[MCC];
#pragma unroll
                for (int q = 0; q < MCC; q++) hq2[q] = hq[q] + 64 * ld;
                const lds_f32* sp = seed + gperm * NS;
                const lds_u8* mp = mrow + gperm * MSTRIDE;
                auto load_ops = [&](Ops& o, int gi) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int row = 16 * gi + 4 * j;
#pragma unroll
                        for (int jj = 0; jj < NS; jj++) o.sd[j][jj] = sp[row * NS + jj];
                        o.mk[j] = mp[row * MSTRIDE];
#pragma unroll
                        for (int q = 0; q < MCC; q++) o.hf[j][q] = (LDC && row >= 64) ? hq2[q][(row - 64) * ld] : hq[q][row * ld];
                    }
                };
                auto mac_ops = [&](const Ops& o, int gi) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (gi == MT - 1 && j > 0 && 16 * gi + 4 * j >= B) continue;     // wave-uniform: all-zero seeds
                        float dv = 0.0f;
#pragma unroll
                        for (int jj = 0; jj < NS; jj++) dv += o.sd[j][jj] * wvn[jj];
                        const float df = (o.mk[j] & MBITS) ? dv : 0.0f;
#pragma unroll
                        for (int q = 0; q < MCC; q++) acc[q] = mfma16(df, o.hf[j][q], acc[q]);
                    }
                };
                Ops o0, o1;
                load_ops(o0, 0);
#pragma unroll
                for (int gi = 0; gi < MT; gi++) {
                    if (gi + 1 < MT) load_ops((gi & 1) ? o0 : o1, gi + 1);
                    mac_ops((gi & 1) ? o1 : o0, gi);
                    if (gi + 1 < MT) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (NS + 1 + MCC), 0);      // next group's reads first
                        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MCC, 0);
                    }
                }
            };
            if (LDH == 200) kloop(std::integral_constant<int, 200>{});
            else if (LDH == 136) kloop(std::integral_constant<int, 136>{});
            else kloop(std::integral_constant<int, 0>{});
            sub_stamp(22);
#ifdef RLC_WG_LATE_ISSUE
            WgPre Q;
            asm volatile("" ::: "memory");
            if (nxt >= 0) issue(Q, nxt);
            else {
#pragma unroll
                for (int q = 0; q < 4; q++) Q.w[q] = Q.m[q] = Q.v[q] = Q.t[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("" ::: "memory");
#endif
            const bool n4ok = 16 * t + 4 * g < N;
            // Every prefetched register of the item is demanded HERE, before the first store of the epilogue.  gfx9's
            // vmcnt counts stores as well as loads, in issue order, and hipcc derives each tile's wait from the loads
            // alone (vmcnt(13), (9), (5), (1) for tiles 0..3): with the previous tiles' stores in the queue behind them
            // the later tiles' waits then stand for "my own stores of a moment ago have been written back" -- a store
            // round trip inside every item.  One wait up front covers only loads issued a k-loop ago.  (All four sets: a
            // three-tile item's fourth was loaded as well, wg_issue.)
            if constexpr (!GONLY && !ablate(1)) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if constexpr (NOPOL) asm volatile("" ::"v"(P.w[q]), "v"(P.m[q]), "v"(P.v[q]));
                    else asm volatile("" ::"v"(P.w[q]), "v"(P.m[q]), "v"(P.v[q]), "v"(P.t[q]));
                }
            }
            if constexpr (ablate(1)) {       // keep the accumulators live, store nothing
#pragma unroll
                for (int q = 0; q < MCC; q++)
                    if (acc[q][0] == 1.2345e33f) Wp[0] = acc[q][1] + acc[q][2] + acc[q][3];
                RLC_RUN_RET;
            }
#pragma unroll
            for (int q = 0; q < MCC; q++) {
                const int kp = 16 * (m0 + q) + c;
                if constexpr (GONLY) {
                    if (q < nq && kp < H1 && n4ok)
                        *reinterpret_cast<f32x4*>(&tapp[(((size_t)(m0 + q) * NT + t) << 8) + lane4]) = acc[q];
                    continue;
                }
                f32x4 nw, nm = P.m[q], nv = P.v[q], nt;
                if constexpr (ablate(13)) {
                    nw = P.w[q] + acc[q]; nm = nm + acc[q]; nt = P.t[q] + acc[q];
                } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float mm = nm[r], vv = nv[r];
                    nw[r] = astep_big(P.w[q][r], acc[q][r], mm, vv, alpha);
                    nm[r] = mm; nv[r] = vv;
                    if constexpr (!NOPOL) nt[r] = polyak(P.t[q][r], nw[r], tau);
                }
                }
                if constexpr (ablate(11)) {
                    if (nw[0] + nm[1] + nv[2] + nt[3] == 1.2345e33f) Wp[0] = 1.0f;
                    continue;
                }
                if (q < nq && kp < H1 && n4ok) {
                    const size_t p = (((size_t)(m0 + q) * NT + t) << 8) + lane4;
                    st_stream(&mp[p], nm);
                    st_stream(&vp[p], nv);
                    *reinterpret_cast<f32x4*>(&Wp[p]) = nw;
                    if constexpr (!NOPOL) st_target(&Wt[p], nt);
                    if (tapp) *reinterpret_cast<f32x4*>(&tapp[p]) = acc[q];
                }
            }
            sub_stamp(23);
            RLC_RUN_RET;
        };
#undef RLC_RUN_RET
#ifdef RLC_WG_LATE_ISSUE
#define RLC_RUN(P, idx, ...) return run(P, idx, nxt, __VA_ARGS__)
        auto run_any = [&](const WgPre& P, int idx, int nxt) {
#else
#define RLC_RUN(P, idx, ...) run(P, idx, __VA_ARGS__)
        auto run_any = [&](const WgPre& P, int idx) {
#endif
            const int ch = idx / NT;
            const int nq = cbase + (ch < crem ? 1 : 0);
#ifndef RLC_WG_EXACT      // per translation unit: +4 % for SoftActorCritic (widths <= 128), -1.5 % for DDPG (register pressure)
            if (nq == 4) RLC_RUN(P, idx, std::integral_constant<int, 4>{}, std::false_type{});
            else RLC_RUN(P, idx, std::integral_constant<int, 3>{}, std::false_type{});
#else
            if (nq == 4) RLC_RUN(P, idx, std::integral_constant<int, 4>{}, std::true_type{});
            else if (nq == 3) RLC_RUN(P, idx, std::integral_constant<int, 3>{}, std::true_type{});
            else RLC_RUN(P, idx, std::integral_constant<int, 3>{}, std::false_type{});      // chunks of < 3 tiles alias tile 0 (never stored)
#endif
        };
#undef RLC_RUN

        WgPre PA, PB;
        int idx = wave;
#ifdef RLC_WG_STAGGER
        // waves 4-7 (the SIMD partners of waves 0-3) start their items late: one wave of a SIMD is then in a k-loop (matrix
        // pipe) while the other streams an epilogue (memory), instead of both doing the same thing at the same time
        if (wave >= 4) __builtin_amdgcn_s_sleep(RLC_WG_STAGGER);      // units of 64 cycles
#endif
        sub_begin();
        // Compiler-level memory barriers pin the prefetch loads and the epilogue stores where they are written:
        // without them hipcc reorders the overlapped prefetch across the stores of the previous item / previous
        // update (K updates in one launch then differ from K launches; tests/test_gpu_ddpg.py pins this).
#define RLC_CBAR() asm volatile("" ::: "memory")
        RLC_CBAR();
        bool first = true;
        if constexpr (NPRE > 0) { PA = pre.a; if constexpr (NPRE > 1) PB = pre.b; }      // already in flight (wgrad_prefetch)
        else if (idx < nitems) issue(PA, idx);
        sub_stamp(21);
#ifdef RLC_WG_LATE_ISSUE
        while (idx < nitems) {
            RLC_CBAR();
            PB = run_any(PA, idx, idx + kWaves < nitems ? idx + kWaves : -1);
            RLC_CBAR();
            idx += kWaves;
            if (idx >= nitems) break;
            PA = run_any(PB, idx, idx + kWaves < nitems ? idx + kWaves : -1);
            RLC_CBAR();
            idx += kWaves;
        }
        (void)first;
#else
        while (idx < nitems) {
            RLC_CBAR();
            if (idx + kWaves < nitems && !(NPRE > 1 && first)) issue(PB, idx + kWaves);
            first = false;
            RLC_CBAR();
            run_any(PA, idx);
            RLC_CBAR();
            idx += kWaves;
            if (idx >= nitems) break;
            if (idx + kWaves < nitems) issue(PA, idx + kWaves);
            RLC_CBAR();
            run_any(PB, idx);
            RLC_CBAR();
            idx += kWaves;
        }
#endif
#undef RLC_CBAR
        sub_begin();
        // extra rows of a concat layer (rank-NE term): G[H1+j][n] = sum_b E[b][j] * D[b][n]; one N tile per
        // wave at a time
        if constexpr (NE > 0 && !ablate(2)) {
            // waves 4-7 carry six of the 52 tile chunks above, waves 0-3 seven: the extra rows go to waves 4-7 first
            for (int t = (wave + 4) & 7; t < NT; t += kWaves) {
                const int n = 16 * t + c;
                const bool nok = n < N;
                float wvn[NS];
#pragma unroll
                for (int j = 0; j < NS; j++) wvn[j] = nok ? wvec[j * 256 + n] : 0.0f;
                float ge[NE];
#pragma unroll
                for (int j = 0; j < NE; j++) ge[j] = 0.0f;
                for (int bb = 0; bb < MT * 4; bb++) {            // (forcing a full unroll here costs 1 %: registers)
                    const int b = 4 * bb + g;
                    float dv = 0.0f;
#pragma unroll
                    for (int j = 0; j < NS; j++) dv += seed[b * NS + j] * wvn[j];
                    const float dd = (L.mask[b * MSTRIDE + 16 * t + c] & MBITS) ? dv : 0.0f;
#pragma unroll
                    for (int j = 0; j < NE; j++) ge[j] += E[b * NE + j] * dd;
                }
#pragma unroll
                for (int j = 0; j < NE; j++) {
                    const float gr = col4_sum(ge[j]);
                    if (g == j && nok) {
                        const size_t p = rlc_blk_index(((H1 + 15) & ~15) + j, n, N);   // first extra block row + j
                        if constexpr (GONLY) { tapp[p] = gr; continue; }
                        float mm = mp[p], vv = vp[p];
                        float o = 0.0f;
                        if constexpr (!NOPOL) o = Wt[p];
                        const float nv = astep_small(Wp[p], gr, mm, vv, alpha);
                        mp[p] = mm; vp[p] = vv; Wp[p] = nv;
                        if (tapp) tapp[p] = gr;
                        if constexpr (!NOPOL) Wt[p] = polyak(o, nv, tau);
                    }
                }
            }
        }
        sub_stamp(24);
#ifdef RLC_STAMPS
        if (lane == 0 && stamp_buf) stamp_buf[32 + wave] += (float)(clock64() - t_wg0);   // per-wave time in this call
#endif
    }

    // the same through this block's Adam form (TADAM: torch's step)
    __device__ __forceinline__ void adam_scalar_m(float* th, float* m, float* v, float* tt, float* tap, int p, float gr,
                                                  float alpha, float tau) const {
        float mm = m[p], vv = v[p];
        const float o = tt ? tt[p] : 0.0f;          // with the other loads: one memory round trip, not two
        const float nv = astep_small(th[p], gr, mm, vv, alpha);
        m[p] = mm; v[p] = vv; th[p] = nv;
        if (tap) tap[p] = gr;
        if (tt) tt[p] = polyak(o, nv, tau);
    }
    // Adam (+ Polyak) on one scalar parameter by the calling lane
    __device__ __forceinline__ static void adam_scalar(float* th, float* m, float* v, float* tt, float* tap, int p,
                                                       float gr, float alpha, float tau) {
        float mm = m[p], vv = v[p];
        const float o = tt ? tt[p] : 0.0f;
        const float nv = adam_step(th[p], gr, mm, vv, alpha);
        m[p] = mm; v[p] = vv; th[p] = nv;
        if (tap) tap[p] = gr;
        if (tt) tt[p] = polyak(o, nv, tau);
    }
};

}  // namespace mfb

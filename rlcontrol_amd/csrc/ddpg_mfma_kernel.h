// ddpg_mfma.hip -- fused DDPG update on gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Same contract as ddpg_generic.hip (one workgroup per agent, n_updates sequential updates per
// launch, every update = sample + gather + agents/DDPG.py:74-95), but the nine [B,200]x[200,200]-class
// contractions of one update run on MFMA tiles and nothing [B,H]-sized ever leaves the CU:
//
//   LDS   hbuf   fp32 [MT*16][LDH]   the trunk activation h1 (target / online / post-critic-step), the
//                                    A operand of every forward GEMM and of both weight-gradient GEMMs
//         mask16 u16  [MT*16][16]    relu masks of g2 / h2, one bit per unit: dg2 = mask*dq*Wc3 and
//                                    dh2 = mask*(dz.Wa3) are rank-A outer products, regenerated on the
//                                    fly as MFMA operands instead of being stored as [B,H] fp32
//         per-sample vectors (x, x', a, y, q, dq, mu, dz ...), row-reduction partials, a staged Wc3/Wa3
//   VGPR  accumulators of the GEMM in flight (MT x 4 tiles of 16x16), weight fragments streamed
//         global -> VGPR (each weight element is read once per GEMM per agent; no LDS staging)
//   HBM   theta, theta', Adam m/v: Wa2/Wc2 are updated (Adam + Polyak) in the epilogue of their
//         weight-gradient GEMM straight from the accumulators
//
// Tiling: batch rows on the MFMA M axis (MT = ceil(B/16) tiles), features on N; wave w of 8 owns
// the ADJACENT N-tiles 2w and 2w+1 (one 128-byte line per weight row) for ALL M tiles, so reductions over the batch (bias / W3 / W1 gradients)
// are wave-local and only reductions over features (q, z, dQ/da) cross waves through LDS partials,
// summed in a fixed order (deterministic: K updates in one launch == K launches, bit for bit).
// fp32 in / fp32 accumulate MFMA is a k-ordered fmaf chain (exact fp32), so the 1e-5 parity bar holds.
//
// Supported shapes: S <= 8, A in {1,2}, H1/HA/HC multiples of 4 in [16,256], B <= 128.
#pragma once
#include <type_traits>

#include "rlc_common.h"
#include "ddpg_rollout_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kWaves = 8;     // two waves per SIMD: one can issue MFMA while the other does VALU / waits on loads
constexpr int NTW = 2;        // N tiles per wave  (N <= 256)
constexpr int NT16 = 16;      // N tiles per row at most (N <= 256)
// Row stride of the byte masks: 272 B = 68 dwords, so that the dword a lane reads in the backward GEMM
// (row 16mt+c, bytes nc+4g..+3) sits in bank (4c + g + const) mod 64: conflict-free for all 64 lanes.
constexpr int MSTRIDE = 16 * NT16 + 16;
constexpr int SMAX = 8;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
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

struct Smem {
    float* hbuf;
    unsigned char* mask;   // relu masks, one BYTE (0/1) per (row, unit): [MB][MSTRIDE]
    float* part;      // [kWaves][MB][AD]
    float* wvec;      // [AD][256] staged Wc3 (row 0) or Wa3 transposed
    float *x, *x2, *a, *aout, *mu, *dz, *q, *y, *dq;
    double *r, *g;
    long long* idx;
    int* pool;
    int* dups;
};

__host__ __device__ inline int ldh_for(int H1) {
    // leading dimension with (LDH/4) % 16 == 2: conflict-free ds_read_b128 rows AND b32 columns
    int q = (H1 + 3) / 4;
    while ((q & 15) != 2) q++;
    return q * 4;
}

__host__ __device__ inline size_t smem_carve(const RlcDims& d, int MT, unsigned char* base, Smem* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int MB = MT * 16, S = d.S, A = d.A, LDH = ldh_for(d.H1);
    // + 16 floats of tail: the unmasked fragment reads of the last k-chunk run up to 15 floats past a row's
    // end (into the next row, or into this zeroed tail after the last row)
    float* hbuf = (float*)take(sizeof(float) * (MB * LDH + 16));
    double* r = (double*)take(sizeof(double) * MB);
    double* g = (double*)take(sizeof(double) * MB);
    long long* idx = (long long*)take(sizeof(long long) * RLC_MAX_BATCH);
    unsigned char* mask = (unsigned char*)take((size_t)MB * MSTRIDE);
    float* part = (float*)take(sizeof(float) * kWaves * MB * A);
    float* wvec = (float*)take(sizeof(float) * A * 256);
    float* x = (float*)take(sizeof(float) * MB * SMAX);      // rows padded to 8 floats: two ds_read_b128
    float* x2 = (float*)take(sizeof(float) * MB * SMAX);
    float* a = (float*)take(sizeof(float) * MB * A);
    float* aout = (float*)take(sizeof(float) * MB * A);
    float* mu = (float*)take(sizeof(float) * MB * A);
    float* dz = (float*)take(sizeof(float) * MB * A);
    float* q = (float*)take(sizeof(float) * MB);
    float* y = (float*)take(sizeof(float) * MB);
    float* dq = (float*)take(sizeof(float) * MB);
    int* pool = (int*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    int* dups = (int*)take(sizeof(int) * 4);
    if (out) {
        out->hbuf = hbuf; out->r = r; out->g = g; out->idx = idx; out->mask = mask; out->part = part;
        out->wvec = wvec; out->x = x; out->x2 = x2; out->a = a; out->aout = aout; out->mu = mu; out->dz = dz;
        out->q = q; out->y = y; out->dq = dq; out->pool = pool; out->dups = dups;
    }
    return off;
}

template <int MT, int AD>
struct Upd {
    static constexpr int MB = MT * 16;

    // per-thread geometry
    int tid, lane, wave, c, g;
    int S, H1, HA, HC, B, LDH;
    Smem L;
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

    // ---------------------------------------------------------------------------------------
    // hbuf[b][k] = relu(b1[k] + sum_i xs[b][i] W1[i][k])   (rows >= B and columns >= H1 zeroed)
    // ---------------------------------------------------------------------------------------
    __device__ __forceinline__ void trunk(const float* W1, const float* b1, const float* xs) {
        if (S <= 4) trunk_t<4>(W1, b1, xs);      // wave-uniform: Pendulum-sized states need one 16-byte read per row
        else trunk_t<SMAX>(W1, b1, xs);
    }
    template <int SP>
    __device__ __forceinline__ void trunk_t(const float* W1, const float* b1, const float* xs) {
        // 256 column slots x 2 row halves
        const int half = tid >> 8;
        for (int k = tid & 255; k < LDH; k += 256) {
            float w[SP];
            float bias = 0.0f;
            const bool live = k < H1;
#pragma unroll
            for (int i = 0; i < SP; i++) w[i] = (live && i < S) ? W1[i * H1 + k] : 0.0f;
            if (live) bias = b1[k];
#pragma unroll 4
            for (int b = half * (MB / 2); b < (half + 1) * (MB / 2); b++) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(&xs[b * SMAX]);
                float acc = 0.0f;      // same i-order as the scalar form; padded lanes multiply by w = 0
#pragma unroll
                for (int i = 0; i < 4; i++) acc += x0[i] * w[i];
                if (SP > 4) {
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(&xs[b * SMAX + 4]);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc += x1[i] * w[(SP > 4 ? 4 : 0) + i];
                }
                acc = fmaxf(acc + bias, 0.0f);
                L.hbuf[b * LDH + k] = (live && b < B) ? acc : 0.0f;
            }
        }
    }

    // ---------------------------------------------------------------------------------------
    // forward GEMM: acc[mt][i] (tile rows 16mt.., cols 16*(2*wave+i)..) = hbuf[:, 0:K] . W[0:K, :]
    // A: one ds_read_b128 per M tile per 16-deep chunk, lane (c,g) holds k = kc+4g+s for step s;
    // B: tile-blocked W (rlc_blk_index): block (kc/16, t) holds rows kc..kc+15 of tile t; this lane needs rows
    //    4g+s of column c -> four dwords 16 B apart inside the block's 1 KB, streamed global -> VGPR.
    // No masks anywhere in the loop: blocks are zero-padded to 16x16 in memory (rows K..16*ceil(K/16)-1 are
    // zeros -- Wc2's action rows live in their own block row, RlcDims::arow0), hbuf columns >= H1 are zeros or
    // finite neighbours (times a zero weight), and the number of tiles a wave owns (2, 1 or 0) is a template
    // parameter.  Two register sets (A and B fragments of the chunk in flight / the next chunk) alternate in a
    // loop unrolled by two, so there are no register-rotation moves either: per chunk a wave issues
    // 7 ds_read_b128 + 4*NOWN global_load_dword + 28*NOWN MFMAs and little else.
    // ---------------------------------------------------------------------------------------
    template <int NOWN>
    __device__ __forceinline__ void fwd_loop(f32x4 (&acc)[MT][NTW], const float* W, int NT, int KB) {
        const float* wp = W + ((size_t)(NTW * wave) << 8) + (((((c >> 2) << 4) + 4 * g) << 2) + (c & 3));
        const size_t wstep = (size_t)NT << 8;                       // floats between block rows
        const float* ap = L.hbuf + c * LDH + 4 * g;
        f32x4 a0[MT], a1[MT];
        float b0[NOWN][4], b1[NOWN][4];
        auto loadA = [&](f32x4 (&dst)[MT], int ch) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) dst[mt] = *reinterpret_cast<const f32x4*>(ap + 16 * mt * LDH + 16 * ch);
        };
        auto loadB = [&](float (&dst)[NOWN][4], int ch) {
#pragma unroll
            for (int i = 0; i < NOWN; i++)
#pragma unroll
                for (int s = 0; s < 4; s++) dst[i][s] = wp[(size_t)ch * wstep + (i << 8) + 4 * s];
        };
        auto mac = [&](const f32x4 (&a)[MT], const float (&b)[NOWN][4]) {
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[mt][i] = mfma16(a[mt][s], b[i][s], acc[mt][i]);
        };
        loadB(b0, 0);
        loadA(a0, 0);
        int ch = 0;
        for (; ch + 2 <= KB; ch += 2) {
            loadB(b1, ch + 1);
            loadA(a1, ch + 1);
            mac(a0, b0);
            if (ch + 2 < KB) {          // wave-uniform
                loadB(b0, ch + 2);
                loadA(a0, ch + 2);
            }
            mac(a1, b1);
        }
        if (ch < KB) mac(a0, b0);      // odd chunk count: the last chunk is already loaded
    }

    __device__ __forceinline__ void fwd_gemm(f32x4 (&acc)[MT][NTW], const float* W, int N, int K) {
        const int NT = (N + 15) >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) acc[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef RLC_STAMPS
        const long long t_w0 = clock64();
#endif
        const int nown = NT - NTW * wave;          // tiles this wave owns: wave-uniform
        const int KB = (K + 15) >> 4;
        if (nown >= 2) fwd_loop<2>(acc, W, NT, KB);
        else if (nown == 1) fwd_loop<1>(acc, W, NT, KB);
#ifdef RLC_STAMPS
        if (lane == 0 && stamp_buf) stamp_buf[48 + wave] += (float)(clock64() - t_w0);   // per-wave k-loop cycles
#endif
    }

    // acc += bias[n] + sum_j E[b][j] * Wx[j][n] ; relu          (E = action rows of the critic concat)
    __device__ __forceinline__ void bias_relu(f32x4 (&acc)[MT][NTW], const float* bias, int N, const float* E,
                                              const float* Wx /* tile-blocked matrix whose rows xrow0+j multiply E, or null */,
                                              int xrow0 = 0) {
        const int NT = (N + 15) >> 4;
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = NTW * wave + i;
            const int n = 16 * t + c;
            const bool ok = t < NT && n < N;
            const float bs = ok ? bias[n] : 0.0f;
            float wx[AD];
#pragma unroll
            for (int j = 0; j < AD; j++) wx[j] = (ok && Wx) ? Wx[rlc_blk_index(xrow0 + j, n, N)] : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[mt][i][r] + bs;
                    if (Wx) {
                        const int b = 16 * mt + 4 * g + r;
#pragma unroll
                        for (int j = 0; j < AD; j++) v += E[b * AD + j] * wx[j];
                    }
                    acc[mt][i][r] = ok ? fmaxf(v, 0.0f) : 0.0f;
                }
        }
    }

    // out[b][j] partial over this wave's columns: sum_n f(acc[b][n]) * coef_j[n]; f = identity or step
    template <bool STEP>
    __device__ __forceinline__ void row_dot(const f32x4 (&acc)[MT][NTW], int N, const float* coef /* [n*cs + j*js] */,
                                            int cs, int js, const float* coef2 /* optional multiplier [n] */,
                                            int blk_row0 = -1 /* >= 0: coef is a tile-blocked matrix, rows blk_row0+j */) {
        const int NT = (N + 15) >> 4;
        float cf[NTW][AD];
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = NTW * wave + i;
            const int n = 16 * t + c;
            const bool ok = t < NT && n < N;
#pragma unroll
            for (int j = 0; j < AD; j++) {
                float v = ok ? (blk_row0 >= 0 ? coef[rlc_blk_index(blk_row0 + j, n, N)]
                                              : coef[(size_t)n * cs + (size_t)j * js])
                             : 0.0f;
                if (coef2) v *= ok ? coef2[n] : 0.0f;
                cf[i][j] = v;
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < AD; j++) {
                    float p = 0.0f;
#pragma unroll
                    for (int i = 0; i < NTW; i++) {
                        const float v = acc[mt][i][r];
                        p += STEP ? (v > 0.0f ? cf[i][j] : 0.0f) : v * cf[i][j];
                    }
                    p = row16_sum(p);
                    if (c == 0) L.part[((size_t)wave * MB + 16 * mt + 4 * g + r) * AD + j] = p;
                }
    }

    // fixed-order sum of the waves' partials
    __device__ __forceinline__ float part_sum(int b, int j) const {
        float s = L.part[((size_t)0 * MB + b) * AD + j];
#pragma unroll
        for (int w = 1; w < kWaves; w++) s += L.part[((size_t)w * MB + b) * AD + j];
        return s;
    }

    // relu masks of the accumulators -> one byte per (row, unit)
    __device__ __forceinline__ void store_masks(const f32x4 (&acc)[MT][NTW], int N) {
        const int NT = (N + 15) >> 4;
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = NTW * wave + i;
            if (t < NT) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        L.mask[(16 * mt + 4 * g + r) * MSTRIDE + 16 * t + c] = acc[mt][i][r] > 0.0f ? 1 : 0;
            }
        }
    }

    // ---------------------------------------------------------------------------------------
    // backward-to-input GEMM: acc[b][k'] = sum_n D[b][n] * W[k'][n],  D[b][n] = mask(b,n) * sum_j seed[b][j]*wv[j][n]
    // (D is never materialised).  k-dim = n in chunks of 16 with lane (c,g) taking n = nc+4g+s:
    //   B = the lane's 16 bytes of block (t, nc/16) of the tile-blocked W: 1 KB contiguous per instruction;
    //   A = the relu mask bytes of row 16mt+c (one ds_read_b32 -> four v_cvt_f32_ubyte).
    // NS == 1 (critic always, actor when A == 1): D is rank one, so the seed leaves the loop --
    //   acc[b][k'] = seed[b] * sum_n maskf(b,n) * (wv[n] W[k'][n]):  A = the 0/1 mask floats as they are, B is
    //   scaled by wv (4 multiplies per tile per chunk) and the rows are scaled by seed[b] once at the end.
    // Same structure as fwd_loop: tiles owned is a template parameter, two register sets, no masks.
    // ---------------------------------------------------------------------------------------
    template <int NS, int NOWN>
    __device__ __forceinline__ void bwd_loop(f32x4 (&acc)[MT][NTW], const float* W, int NTk, const float* seed) {
        const float* wp = W + (((size_t)(NTW * wave) * NTk) << 8) + (lane << 2);
        const unsigned char* mp = L.mask + c * MSTRIDE + 4 * g;
        const float* wvp = L.wvec + 4 * g;
        float sd[MT][NS];
        if (NS > 1) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int j = 0; j < NS; j++) sd[mt][j] = seed[(16 * mt + c) * NS + j];
        }
        f32x4 b0[NOWN], b1[NOWN];
        auto loadB = [&](f32x4 (&dst)[NOWN], int ch) {
#pragma unroll
            for (int i = 0; i < NOWN; i++)
                dst[i] = *reinterpret_cast<const f32x4*>(wp + (((size_t)i * NTk + ch) << 8));
        };
        auto mac = [&](const f32x4 (&bin)[NOWN], int ch) {
            f32x4 wv[NS], b[NOWN];
#pragma unroll
            for (int j = 0; j < NS; j++) wv[j] = *reinterpret_cast<const f32x4*>(wvp + j * 256 + 16 * ch);
#pragma unroll
            for (int i = 0; i < NOWN; i++) b[i] = NS == 1 ? bin[i] * wv[0] : bin[i];
            f32x4 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const unsigned mw = *reinterpret_cast<const unsigned*>(mp + 16 * mt * MSTRIDE + 16 * ch);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float f = (float)((mw >> (8 * s)) & 0xffu);        // v_cvt_f32_ubyte<s>: 0.0 or 1.0
                    if (NS == 1) {
                        av[mt][s] = f;
                    } else {
                        float v = 0.0f;
#pragma unroll
                        for (int j = 0; j < NS; j++) v += sd[mt][j] * wv[j][s];
                        av[mt][s] = f * v;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < NOWN; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[mt][i] = mfma16(av[mt][s], b[i][s], acc[mt][i]);
        };
        loadB(b0, 0);
        int ch = 0;
        for (; ch + 2 <= NTk; ch += 2) {
            loadB(b1, ch + 1);
            mac(b0, ch);
            if (ch + 2 < NTk) loadB(b0, ch + 2);      // wave-uniform
            mac(b1, ch + 1);
        }
        if (ch < NTk) mac(b0, ch);
        if (NS == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const f32x4 sv = *reinterpret_cast<const f32x4*>(&seed[16 * mt + 4 * g]);
#pragma unroll
                for (int i = 0; i < NOWN; i++) acc[mt][i] = acc[mt][i] * sv;
            }
        }
    }

    template <int NS>
    __device__ __forceinline__ void bwd_gemm(f32x4 (&acc)[MT][NTW], const float* W, int Nk /* row length = k-dim */,
                                             int Kout /* rows of W used = H1 */, const float* seed /* LDS [MB][NS] */) {
        const int NT = (Kout + 15) >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) acc[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nown = NT - NTW * wave;
        const int NTk = (Nk + 15) >> 4;
        if (nown >= 2) bwd_loop<NS, 2>(acc, W, NTk, seed);
        else if (nown == 1) bwd_loop<NS, 1>(acc, W, NTk, seed);
    }

    // epilogue of bwd_gemm: dh1 = acc * (hbuf > 0); column-reduce into the W1 / b1 gradients of this wave's
    // trunk units and apply Adam (+ optional Polyak) right here.
    __device__ __forceinline__ void trunk_grad_adam(const f32x4 (&acc)[MT][NTW], float* th, float* m, float* v,
                                                    float alpha, int oW1, int ob1, float* tap, float* tt, float tau) {
        if (S <= 4) trunk_grad_adam_t<4>(acc, th, m, v, alpha, oW1, ob1, tap, tt, tau);     // wave-uniform
        else trunk_grad_adam_t<SMAX>(acc, th, m, v, alpha, oW1, ob1, tap, tt, tau);
    }
    template <int SP>
    __device__ __forceinline__ void trunk_grad_adam_t(const f32x4 (&acc)[MT][NTW], float* th, float* m, float* v,
                                                      float alpha, int oW1, int ob1, float* tap, float* tt, float tau) {
        const int NT = (H1 + 15) >> 4;
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const int t = NTW * wave + i;
            if (t >= NT) continue;
            const int k = 16 * t + c;
            float gb = 0.0f;
            float gw[SP];
#pragma unroll
            for (int s = 0; s < SP; s++) gw[s] = 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int b = 16 * mt + 4 * g + r;
                    const float hv = L.hbuf[b * LDH + (k < H1 ? k : 0)];
                    const float d = (k < H1 && hv > 0.0f) ? acc[mt][i][r] : 0.0f;
                    gb += d;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&L.x[b * SMAX]);
#pragma unroll
                    for (int s = 0; s < 4; s++) gw[s] += x0[s] * d;
                    if (SP > 4) {
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&L.x[b * SMAX + 4]);
#pragma unroll
                        for (int s = 0; s < 4; s++) gw[(SP > 4 ? 4 : 0) + s] += x1[s] * d;
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
                    float mm = m[p], vv = v[p];
                    const float nv = adam_step(th[p], gr, mm, vv, alpha);
                    m[p] = mm; v[p] = vv; th[p] = nv;
                    if (tap) tap[p] = gr;
                    if (tt) { const float o = tt[p]; tt[p] = o + tau * (nv - o); }
                }
            }
        }
    }

    // ---------------------------------------------------------------------------------------
    // weight-gradient GEMM + Adam (+Polyak) epilogue:
    //   G[k'][n] = sum_b X[b][k'] * D[b][n],  X = [hbuf | E] (E = action columns, or none), D as above.
    // TRANSPOSED tiles: acc[q][r] = G[k' = 16(m0+q) + c][n = 16t + 4g + r] (D^T on the A side, hbuf on the B
    // side), so each lane owns 4 CONSECUTIVE n of one weight row = its 16 bytes of block (m0+q, t) of the
    // tile-blocked arrays: the W / m / v / W' traffic of the Adam epilogue is one 1 KB-contiguous load and one
    // store per array per tile.  k-dim = batch, lane group g takes b = 4*ks + {0,2,1,3}[g] (conflict-free reads).
    //
    // Work items = (N tile t, chunk of <= 4 M' tiles); the 13 x 4 items of a 200 x 200 matrix are dealt
    // round-robin to the 8 waves (every output tile is independent: no cross-wave reduction), so all four SIMDs
    // carry the same MFMA load.  While the k-loop of one item runs, the W / m / v / W' of the wave's NEXT item are
    // already in flight into a second register set.
    // ---------------------------------------------------------------------------------------
    struct WgPre { f32x4 w[4], m[4], v[4], t[4]; };

    template <int NS>
    __device__ __forceinline__ void wgrad_adam(const float* seed /* LDS [MB][NS] */, const float* E /* LDS [MB][AD] or null */,
                                               int Krows /* H1 (+AD if E) */, int N, float* Wp, float* mp, float* vp,
                                               float alpha, float* tapp, float* Wt, float tau) {
        (void)Krows;
        const int NT = (N + 15) >> 4;
        const int NMT = (H1 + 15) >> 4;                  // MFMA rows: the trunk units; action rows below
        const int nch = (NMT + 3) >> 2, cbase = NMT / nch, crem = NMT % nch;    // chunk sizes differ by at most one
        const int nitems = NT * nch;
        const int gperm = ((g & 1) << 1) | (g >> 1);     // 0,2,1,3
        const int lane4 = (g * 16 + c) << 2;

        auto item_geom = [&](int idx, int& t, int& m0, int& nq) {
            t = idx % NT;
            const int ch = idx / NT;
            nq = cbase + (ch < crem ? 1 : 0);
            m0 = ch * cbase + (ch < crem ? ch : crem);
        };
        // Prefetch an item's W / m / v / W' NOW: their HBM latency hides under the previous item's k-loop
        // (addresses clamped, stores predicated).
        auto issue = [&](WgPre& P, int idx) {
            int t, m0, nq;
            item_geom(idx, t, m0, nq);
            const bool n4ok = 16 * t + 4 * g < N;        // N % 4 == 0: all four columns valid or none
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int kp = 16 * (m0 + q) + c;
                const size_t p = (q < nq && kp < H1 && n4ok) ? ((((size_t)(m0 + q) * NT + t) << 8) + lane4) : 0;
                P.w[q] = *reinterpret_cast<const f32x4*>(&Wp[p]);
                P.m[q] = *reinterpret_cast<const f32x4*>(&mp[p]);
                P.v[q] = *reinterpret_cast<const f32x4*>(&vp[p]);
                P.t[q] = *reinterpret_cast<const f32x4*>(&Wt[p]);
            }
        };
        auto run = [&](const WgPre& P, int idx, auto mcc_tag) {
            constexpr int MCC = decltype(mcc_tag)::value;
            int t, m0, nq;
            item_geom(idx, t, m0, nq);
            const int n = 16 * t + c;
            float wvn[NS];
#pragma unroll
            for (int j = 0; j < NS; j++) wvn[j] = n < N ? L.wvec[j * 256 + n] : 0.0f;
            f32x4 acc[MCC];
            int kq[MCC];
#pragma unroll
            for (int q = 0; q < MCC; q++) {
                acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                kq[q] = (q < nq ? 16 * (m0 + q) : 0) + c;             // rows past the chunk alias tile 0 (never stored)
            }
            const unsigned char* mrow = L.mask + 16 * t + c;
            sub_begin();
#pragma unroll 4
            for (int ks = 0; ks < MT * 4; ks++) {
                const int b = 4 * ks + gperm;
                // D[b][n] for this lane's (b, n = 16t + c)
                float dv = 0.0f;
#pragma unroll
                for (int j = 0; j < NS; j++) dv += seed[b * NS + j] * wvn[j];
                const float df = mrow[b * MSTRIDE] ? dv : 0.0f;
                // hbuf fragments, unmasked: columns kp >= H1 (last tile only) only feed rows that are never stored
                float hf[MCC];
#pragma unroll
                for (int q = 0; q < MCC; q++) hf[q] = L.hbuf[b * LDH + kq[q]];
#pragma unroll
                for (int q = 0; q < MCC; q++) acc[q] = mfma16(df, hf[q], acc[q]);
            }
            sub_stamp(22);
            const bool n4ok = 16 * t + 4 * g < N;
#pragma unroll
            for (int q = 0; q < MCC; q++) {
                const int kp = 16 * (m0 + q) + c;
                f32x4 nw, nm = P.m[q], nv = P.v[q], nt;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float mm = nm[r], vv = nv[r];
                    nw[r] = adam_step_fast(P.w[q][r], acc[q][r], mm, vv, alpha);
                    nm[r] = mm; nv[r] = vv;
                    nt[r] = P.t[q][r] + tau * (nw[r] - P.t[q][r]);
                }
                if (q < nq && kp < H1 && n4ok) {
                    const size_t p = (((size_t)(m0 + q) * NT + t) << 8) + lane4;
                    *reinterpret_cast<f32x4*>(&mp[p]) = nm;
                    *reinterpret_cast<f32x4*>(&vp[p]) = nv;
                    *reinterpret_cast<f32x4*>(&Wp[p]) = nw;
                    *reinterpret_cast<f32x4*>(&Wt[p]) = nt;
                    if (tapp) *reinterpret_cast<f32x4*>(&tapp[p]) = acc[q];
                }
            }
            sub_stamp(23);
        };
        auto run_any = [&](const WgPre& P, int idx) {
            const int ch = idx / NT;
            if (cbase + (ch < crem ? 1 : 0) == 4) run(P, idx, std::integral_constant<int, 4>{});
            else run(P, idx, std::integral_constant<int, 3>{});
        };

        WgPre PA, PB;
        int idx = wave;
        sub_begin();
        // Compiler-level memory barriers pin the prefetch loads and the epilogue stores where they are written:
        // without them hipcc reorders the overlapped prefetch across the stores of the previous item / previous
        // update (K updates in one launch then differ from K launches; tests/test_gpu_ddpg.py pins this).
#define CBAR() asm volatile("" ::: "memory")
        CBAR();
        if (idx < nitems) issue(PA, idx);
        sub_stamp(21);
        while (idx < nitems) {
            CBAR();
            if (idx + kWaves < nitems) issue(PB, idx + kWaves);
            CBAR();
            run_any(PA, idx);
            CBAR();
            idx += kWaves;
            if (idx >= nitems) break;
            if (idx + kWaves < nitems) issue(PA, idx + kWaves);
            CBAR();
            run_any(PB, idx);
            CBAR();
            idx += kWaves;
        }
#undef CBAR
        sub_begin();
        // action rows of the critic's concat (rank-AD term): G[H1+j][n] = sum_b E[b][j] * D[b][n]; one N tile per
        // wave at a time
        if (E != nullptr) {
            for (int t = wave; t < NT; t += kWaves) {
                const int n = 16 * t + c;
                const bool nok = n < N;
                float wvn[NS];
#pragma unroll
                for (int j = 0; j < NS; j++) wvn[j] = nok ? L.wvec[j * 256 + n] : 0.0f;
                float ge[AD];
#pragma unroll
                for (int j = 0; j < AD; j++) ge[j] = 0.0f;
                for (int bb = 0; bb < MB / 4; bb++) {
                    const int b = 4 * bb + g;
                    float dv = 0.0f;
#pragma unroll
                    for (int j = 0; j < NS; j++) dv += seed[b * NS + j] * wvn[j];
                    const float dd = L.mask[b * MSTRIDE + 16 * t + c] ? dv : 0.0f;
#pragma unroll
                    for (int j = 0; j < AD; j++) ge[j] += E[b * AD + j] * dd;
                }
#pragma unroll
                for (int j = 0; j < AD; j++) {
                    const float gr = col4_sum(ge[j]);
                    if (g == j && nok) {
                        const size_t p = rlc_blk_index(((H1 + 15) & ~15) + j, n, N);   // RlcDims::arow0 + j
                        float mm = mp[p], vv = vp[p];
                        const float nv = adam_step(Wp[p], gr, mm, vv, alpha);
                        mp[p] = mm; vp[p] = vv; Wp[p] = nv;
                        if (tapp) tapp[p] = gr;
                        const float o = Wt[p];
                        Wt[p] = o + tau * (nv - o);
                    }
                }
            }
        }
        sub_stamp(24);
    }
};

template <int MT, int AD>
__global__ __launch_bounds__(kThreads) void rlc_ddpg_update_mfma_kernel(RlcDev dv, int first_agent, int n_updates,
                                                                        int source, const long long* host_idx,
                                                                        int grad_taps, const RlcRollout* rollout,
                                                                        int q8_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using U = Upd<MT, AD>;
    constexpr int MB = U::MB;
    const RlcDims d = dv.d;
    U u;
    u.tid = threadIdx.x; u.lane = u.tid & 63; u.wave = __builtin_amdgcn_readfirstlane(u.tid >> 6);
    u.c = u.lane & 15; u.g = u.lane >> 4;
    u.S = d.S; u.H1 = d.H1; u.HA = d.HA; u.HC = d.HC; u.B = d.B; u.LDH = ldh_for(d.H1);
    smem_carve(d, MT, smem, &u.L);
    Smem& L = u.L;
    const int tid = u.tid, S = d.S, H1 = d.H1, HA = d.HA, HC = d.HC, B = d.B;
    const int agent = first_agent + blockIdx.x;

    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* m_a = dv.m_a + (size_t)agent * d.Ppad;
    float* v_a = dv.v_a + (size_t)agent * d.Ppad;
    float* m_c = dv.m_c + (size_t)agent * d.Ppad;
    float* v_c = dv.v_c + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 4;
    const float lr_a = dv.actor_lr[agent], lr_c = dv.critic_lr[agent], tau = dv.tau;
#ifdef RLC_STAMPS
    float* stamp_buf = grad_taps ? dv.tap_gc + (size_t)agent * d.Ppad : nullptr;   // diagnostic build: no gradient taps
    float* tap_gc = nullptr;
    float* tap_ga = nullptr;
#else
    float* tap_gc = grad_taps ? dv.tap_gc + (size_t)agent * d.Ppad : nullptr;
    float* tap_ga = grad_taps ? dv.tap_ga + (size_t)agent * d.Ppad : nullptr;
#endif
    float amax[AD];
#pragma unroll
    for (int j = 0; j < AD; j++) amax[j] = dv.amax[j];

    // zero the padded tails of the per-sample vectors once (rows >= B never change afterwards)
    for (int i = tid; i < MB * AD; i += kThreads) { L.a[i] = 0.f; L.aout[i] = 0.f; L.mu[i] = 0.f; L.dz[i] = 0.f; }
    for (int i = tid; i < MB * SMAX; i += kThreads) { L.x[i] = 0.f; L.x2[i] = 0.f; }
    for (int i = tid; i < MB; i += kThreads) { L.q[i] = 0.f; L.y[i] = 0.f; L.dq[i] = 0.f; }
    for (int i = tid; i < MB * MSTRIDE / 4; i += kThreads) reinterpret_cast<unsigned int*>(L.mask)[i] = 0u;
    if (tid < 16) L.hbuf[MB * u.LDH + tid] = 0.0f;
    __syncthreads();

    f32x4 acc[MT][NTW];
#ifdef RLC_STAMPS
    // diagnostic build only: phase boundaries in shader cycles, written where the critic gradient tap lives
    long long t_prev = clock64();
    int stamp_i = 0;
#define STAMP()                                                                                  \
    do {                                                                                         \
        if (tid == 0 && stamp_buf) { const long long t = clock64(); stamp_buf[stamp_i] += (float)(t - t_prev); t_prev = t; } \
        stamp_i++;                                                                               \
    } while (0)
    if (stamp_buf) for (int i = tid; i < 64; i += kThreads) stamp_buf[i] = 0.0f;
    const long long t_k0 = clock64(), w_k0 = wall_clock64();
    u.stamp_buf = stamp_buf;
    __syncthreads();
#else
#define STAMP() do {} while (0)
#endif

    for (int upd = 0; upd < n_updates; upd++) {
#ifdef RLC_STAMPS
        stamp_i = 0;
#endif
        // Re-materialise lane geometry every update: without this hipcc hoists the address arithmetic of
        // all ~15 phases out of the update loop and then spills it (190 scratch stores in the prologue).
        asm volatile("" : "+v"(u.c), "+v"(u.g), "+s"(u.wave));
        if (rollout) {
            // on-device experiment loop: one environment step first; the update runs when learn() would
            // (agents/base_agent.py:65-70).  hbuf is free here (the trunk overwrites it below).
            if (!rlc_train_step_device(rollout, agent, L.hbuf, upd == 0 ? q8_first : 0)) continue;
        }
        // ================= sample + gather (utils/replaybuffer.py:32-37) =================
        u.sub_begin();
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            const unsigned long long call = dv.rep.sample_ctr[agent];
            __syncthreads();
            u.sub_stamp(25);
            rlc_sample_distinct(ring.size, B, dv.rep.seed[agent], call, L.pool, L.idx, L.dups);
            if (tid == 0) dv.rep.sample_ctr[agent] = call + 1;
            u.sub_stamp(26);
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < B; b += kThreads) L.idx[b] = host_idx[((size_t)blockIdx.x * n_updates + upd) * B + b];
        }
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + b;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot * AD; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = dv.rep.gr[slot]; L.g[b] = dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[b]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot * AD; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = dv.rep.rr[slot]; L.g[b] = dv.rep.rg[slot];
            }
            for (int i = 0; i < S; i++) {
                L.x[b * SMAX + i] = clip_state_val(ps[i], dv.clip_state, dv.smin[i], dv.smax[i]);
                L.x2[b * SMAX + i] = clip_state_val(ps2[i], dv.clip_state, dv.smin[i], dv.smax[i]);
            }
#pragma unroll
            for (int j = 0; j < AD; j++) L.a[b * AD + j] = pa[j];
        }
        u.sub_stamp(27);
        __syncthreads();
        STAMP();

        // ================= steps 1-2: target networks on s' (DDPG.py:77) =================
        u.trunk(tt + d.oW1, tt + d.ob1, L.x2);
        __syncthreads();
        STAMP();
        u.fwd_gemm(acc, tt + d.oWa2, HA, H1);
        u.bias_relu(acc, tt + d.oba2, HA, nullptr, nullptr);
        u.template row_dot<false>(acc, HA, tt + d.oWa3, AD, 1, nullptr);          // z' partials
        __syncthreads();
        STAMP();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            L.aout[i] = tanhf(u.part_sum(b, j) + tt[d.oba3 + j]) * amax[j];
        }
        __syncthreads();
        STAMP();
        u.fwd_gemm(acc, tt + d.oWc2, HC, H1);
        u.bias_relu(acc, tt + d.obc2, HC, L.aout, tt + d.oWc2, d.arow0);
        // q' partials: only column j = 0 of the partial buffer is meaningful here
        {
            // reuse row_dot with coef = Wc3' (stride 1, js 0 -> every j gets the same value)
            u.template row_dot<false>(acc, HC, tt + d.oWc3, 1, 0, nullptr);
        }
        __syncthreads();
        STAMP();
        for (int b = tid; b < B; b += kThreads) {
            const float qt = u.part_sum(b, 0) + tt[d.obc3];
            const float y = (float)(L.r[b] + L.g[b] * (double)qt);     // float64 TD glue (DDPG.py:80-84)
            L.y[b] = y;
            dv.tap_y[(size_t)agent * RLC_MAX_BATCH + b] = y;
        }
        __syncthreads();
        STAMP();

        // ================= step 3: critic step =================
        u.trunk(th + d.oW1, th + d.ob1, L.x);
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < HC ? th[d.oWc3 + n] : 0.0f;
        __syncthreads();
        STAMP();
        u.fwd_gemm(acc, th + d.oWc2, HC, H1);
        u.bias_relu(acc, th + d.obc2, HC, L.a, th + d.oWc2, d.arow0);
        u.template row_dot<false>(acc, HC, th + d.oWc3, 1, 0, nullptr);           // q partials
        __syncthreads();
        STAMP();
        for (int b = tid; b < B; b += kThreads) {
            const float q = u.part_sum(b, 0) + th[d.obc3];
            L.q[b] = q;
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = q;
            L.dq[b] = 2.0f * (q - L.y[b]) / (float)B;                  // d mean((y-q)^2)/dq
        }
        __syncthreads();
        STAMP();
        // wave-local column reductions from the live g2 accumulators: dWc3, dbc2; then the relu masks
        float g_wc3[NTW], g_bc2[NTW];
        {
            const int NT = (HC + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = NTW * u.wave + i;
                const int n = 16 * t + u.c;
                const float w3 = (t < NT && n < HC) ? L.wvec[n] : 0.0f;
                float s3 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 dq4 = *reinterpret_cast<const f32x4*>(&L.dq[16 * mt + 4 * u.g]);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float gv = acc[mt][i][r];
                        s3 += gv * dq4[r];
                        s2 += gv > 0.0f ? dq4[r] * w3 : 0.0f;
                    }
                }
                g_wc3[i] = col4_sum(s3);
                g_bc2[i] = col4_sum(s2);
            }
        }
        u.store_masks(acc, HC);
        __syncthreads();
        STAMP();
        // dh1 = (dg2 . Wc2[:H1]^T) * relu'(h1) -> W1/b1 gradients -> critic Adam on the trunk (Q1)
        const float alpha_c = adam_alpha(lr_c, pw[2], pw[3]);
        u.template bwd_gemm<1>(acc, th + d.oWc2, HC, H1, L.dq);
        __syncthreads();      // every wave has finished reading the pre-step Wc2 rows and W1
        STAMP();
        u.trunk_grad_adam(acc, th, m_c, v_c, alpha_c, d.oW1, d.ob1, tap_gc, nullptr, 0.0f);
        STAMP();
        // dWc2 = [h1|a]^T . dg2 with Adam + Polyak in the epilogue
        u.template wgrad_adam<1>(L.dq, L.a, H1 + AD, HC, th + d.oWc2, m_c + d.oWc2, v_c + d.oWc2, alpha_c,
                     tap_gc ? tap_gc + d.oWc2 : nullptr, tt + d.oWc2, tau);
        // small critic tensors: Wc3, bc2 (column owners), bc3 (one thread)
        {
            const int NT = (HC + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = NTW * u.wave + i;
                const int n = 16 * t + u.c;
                if (t < NT && n < HC && u.g < 2) {
                    const int p = (u.g == 0) ? d.oWc3 + n : d.obc2 + n;
                    const float gr = (u.g == 0) ? g_wc3[i] : g_bc2[i];
                    float mm = m_c[p], vv = v_c[p];
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_c);
                    m_c[p] = mm; v_c[p] = vv; th[p] = nv;
                    if (tap_gc) tap_gc[p] = gr;
                    const float o = tt[p];
                    tt[p] = o + tau * (nv - o);
                }
            }
            if (u.wave == 0) {            // bc3: sum_b dq[b] by one wave (fixed-order shuffle tree)
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dq[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) {
                    const int p = d.obc3;
                    float mm = m_c[p], vv = v_c[p];
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_c);
                    m_c[p] = mm; v_c[p] = vv; th[p] = nv;
                    if (tap_gc) tap_gc[p] = gr;
                    const float o = tt[p];
                    tt[p] = o + tau * (nv - o);
                }
            }
        }
        __syncthreads();
        STAMP();
        if (tid == 0) { pw[2] *= 0.9f; pw[3] *= 0.999f; }

        // ================= step 4: actor forward with the updated trunk (DDPG.py:90) =================
        u.trunk(th + d.oW1, th + d.ob1, L.x);
        for (int i = tid; i < AD * 256; i += kThreads) {
            const int j = i / 256, n = i % 256;
            L.wvec[i] = n < HA ? th[d.oWa3 + n * AD + j] : 0.0f;       // Wa3 transposed [j][n]
        }
        __syncthreads();
        STAMP();
        u.fwd_gemm(acc, th + d.oWa2, HA, H1);
        u.bias_relu(acc, th + d.oba2, HA, nullptr, nullptr);
        u.template row_dot<false>(acc, HA, th + d.oWa3, AD, 1, nullptr);          // z partials
        u.store_masks(acc, HA);
        __syncthreads();
        STAMP();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            const float mu = tanhf(u.part_sum(b, j) + th[d.oba3 + j]);
            L.mu[i] = mu;
            const float ao = mu * amax[j];
            L.aout[i] = ao;
            dv.tap_aout[(size_t)agent * RLC_MAX_BATCH * AD + i] = ao;
        }
        __syncthreads();
        STAMP();
        // the h2 accumulators are needed again for dWa3 once dz is known: park them in registers
        f32x4 h2acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) h2acc[mt][i] = acc[mt][i];

        // ================= step 5: dQ/da at the scaled action, updated critic (DDPG.py:91) =================
        u.fwd_gemm(acc, th + d.oWc2, HC, H1);
        u.bias_relu(acc, th + d.obc2, HC, L.aout, th + d.oWc2, d.arow0);
        // dqda[b][j] = sum_n step(g2[b][n]) * Wc3[n] * Wc2[H1+j][n]
        u.template row_dot<true>(acc, HC, th + d.oWc2, 1, HC, th + d.oWc3, d.arow0);
        __syncthreads();
        STAMP();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            const float dqda = u.part_sum(b, j);
            dv.tap_dqda[(size_t)agent * RLC_MAX_BATCH * AD + i] = dqda;
            const float mu = L.mu[i];
            L.dz[i] = -dqda * (1.0f - mu * mu);                         // grad_ys = -dQ/da on tanh output (Q3)
        }
        __syncthreads();
        STAMP();

        // ================= step 6: actor step =================
        float g_wa3[NTW][AD], g_ba2[NTW];
        {
            const int NT = (HA + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = NTW * u.wave + i;
                const int n = 16 * t + u.c;
                const bool ok = t < NT && n < HA;
                float w3[AD], s3[AD];
#pragma unroll
                for (int j = 0; j < AD; j++) { w3[j] = ok ? L.wvec[j * 256 + n] : 0.0f; s3[j] = 0.0f; }
                float s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int b = 16 * mt + 4 * u.g + r;
                        const float hv = h2acc[mt][i][r];
                        float dd = 0.0f;
#pragma unroll
                        for (int j = 0; j < AD; j++) {
                            const float dzb = L.dz[b * AD + j];
                            s3[j] += hv * dzb;
                            dd += dzb * w3[j];
                        }
                        s2 += hv > 0.0f ? dd : 0.0f;
                    }
#pragma unroll
                for (int j = 0; j < AD; j++) g_wa3[i][j] = col4_sum(s3[j]);
                g_ba2[i] = col4_sum(s2);
            }
        }
        const float alpha_a = adam_alpha(lr_a, pw[0], pw[1]);
        u.template bwd_gemm<AD>(acc, th + d.oWa2, HA, H1, L.dz);
        __syncthreads();
        STAMP();
        u.trunk_grad_adam(acc, th, m_a, v_a, alpha_a, d.oW1, d.ob1, tap_ga, tt, tau);
        STAMP();
        u.template wgrad_adam<AD>(L.dz, nullptr, H1, HA, th + d.oWa2, m_a + d.oWa2, v_a + d.oWa2, alpha_a,
                     tap_ga ? tap_ga + d.oWa2 : nullptr, tt + d.oWa2, tau);
        {
            const int NT = (HA + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = NTW * u.wave + i;
                const int n = 16 * t + u.c;
                if (t < NT && n < HA && u.g <= AD) {
                    // lane group 0 -> ba2[n]; groups 1..AD -> Wa3[n][j]
                    int p = d.oba2 + n;
                    float gr = g_ba2[i];
#pragma unroll
                    for (int j = 0; j < AD; j++)
                        if (u.g == j + 1) { p = d.oWa3 + n * AD + j; gr = g_wa3[i][j]; }
                    float mm = m_a[p], vv = v_a[p];
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_a);
                    m_a[p] = mm; v_a[p] = vv; th[p] = nv;
                    if (tap_ga) tap_ga[p] = gr;
                    const float o = tt[p];
                    tt[p] = o + tau * (nv - o);
                }
            }
            if (u.wave < AD) {            // ba3[j]: sum_b dz[b][j], wave j
                const int j = u.wave;
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dz[b * AD + j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) {
                    const int p = d.oba3 + j;
                    float mm = m_a[p], vv = v_a[p];
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_a);
                    m_a[p] = mm; v_a[p] = vv; th[p] = nv;
                    if (tap_ga) tap_ga[p] = gr;
                    const float o = tt[p];
                    tt[p] = o + tau * (nv - o);
                }
            }
        }
        __syncthreads();
        STAMP();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; }
        __syncthreads();
    }
#ifdef RLC_STAMPS
    if (tid == 0 && stamp_buf) {
        stamp_buf[40] = (float)(clock64() - t_k0);
        stamp_buf[41] = (float)(wall_clock64() - w_k0);
    }
#endif
}

template <int MT, int AD>
int launch_t(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source, const long long* idx_dev,
             int grad_taps, hipStream_t st, const RlcRollout* rollout, int q8_first) {
    const size_t lds = smem_carve(dv.d, MT, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "MFMA DDPG kernel needs %zu B of LDS (> 160 KiB)", lds);
    auto kern = rlc_ddpg_update_mfma_kernel<MT, AD>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates, source, idx_dev,
                       grad_taps, rollout, q8_first);
    RLC_HIP(hipGetLastError());
    return 0;
}

}  // namespace

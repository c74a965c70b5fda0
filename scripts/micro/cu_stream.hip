// Micro-benchmark (GPU box): what one CU's memory path sustains on the weight-gradient epilogue's access pattern --
// four arrays (W, m, v, W') read and written back in 1 KB blocks, 16 bytes per lane, one 512-thread workgroup per CU,
// each workgroup on its own 4 x 320 KB of state (a DDPG agent's two big matrices), repeated.  PRE = items (16 KB per
// wave) in flight per wave.  Printed: GB/s per CU (read + written) with 1 workgroup and with 256.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_stream scripts/micro/cu_stream.hip && /tmp/cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTiles = 4;                 // 1 KB blocks per array per item
constexpr size_t kFloats = 80 * 1024;     // floats per array per workgroup (320 KB)

template <int PRE, bool NT>
__global__ __launch_bounds__(512) void stream(float* base, int reps) {
    float* W = base + (size_t)blockIdx.x * 4 * kFloats;
    float *M = W + kFloats, *V = M + kFloats, *T = V + kFloats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nitems = kFloats / (256 * kTiles);         // 80 items of 4 KB per array
    for (int r = 0; r < reps; r++) {
        f32x4 w[PRE][kTiles], m[PRE][kTiles], v[PRE][kTiles], t[PRE][kTiles];
        auto issue = [&](int slot, int it) {
#pragma unroll
            for (int q = 0; q < kTiles; q++) {
                const size_t p = ((size_t)it * kTiles + q) * 256 + lane * 4;
                w[slot][q] = *reinterpret_cast<const f32x4*>(W + p);
                if (NT) {
                    m[slot][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(M + p));
                    v[slot][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(V + p));
                    t[slot][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(T + p));
                } else {
                    m[slot][q] = *reinterpret_cast<const f32x4*>(M + p);
                    v[slot][q] = *reinterpret_cast<const f32x4*>(V + p);
                    t[slot][q] = *reinterpret_cast<const f32x4*>(T + p);
                }
            }
        };
        int it = wave;
#pragma unroll
        for (int s = 0; s < PRE; s++)
            if (it + 8 * s < nitems) issue(s, it + 8 * s);
        for (int k = 0; it < nitems; it += 8, k++) {
#pragma unroll
            for (int s = 0; s < PRE; s++) {
                if ((k % PRE) != s) continue;
#pragma unroll
                for (int q = 0; q < kTiles; q++) {
                    const size_t p = ((size_t)it * kTiles + q) * 256 + lane * 4;
                    const f32x4 nm = m[s][q] * 0.9f + w[s][q] * 0.1f, nv = v[s][q] * 0.999f + 1e-3f;
                    const f32x4 nw = w[s][q] - nm * 1e-3f, nt = t[s][q] + (nw - t[s][q]) * 0.01f;
                    if (NT) {
                        __builtin_nontemporal_store(nm, reinterpret_cast<f32x4*>(M + p));
                        __builtin_nontemporal_store(nv, reinterpret_cast<f32x4*>(V + p));
                        __builtin_nontemporal_store(nt, reinterpret_cast<f32x4*>(T + p));
                    } else {
                        *reinterpret_cast<f32x4*>(M + p) = nm;
                        *reinterpret_cast<f32x4*>(V + p) = nv;
                        *reinterpret_cast<f32x4*>(T + p) = nt;
                    }
                    *reinterpret_cast<f32x4*>(W + p) = nw;
                }
                asm volatile("" ::: "memory");
                if (it + 8 * PRE < nitems) issue(s, it + 8 * PRE);
                asm volatile("" ::: "memory");
            }
        }
    }
}

template <int PRE, bool NT>
void run(float* buf, int nblocks) {
    const int reps = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    stream<PRE, NT><<<nblocks, 512>>>(buf, 5);
    hipEventRecord(e0);
    stream<PRE, NT><<<nblocks, 512>>>(buf, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * 4 * kFloats * 4 * reps;         // read + written, per workgroup
    printf("PRE %d  nt %d  workgroups %3d:  %.1f us per pass  %.1f GB/s per CU  %.2f TB/s total\n", PRE, (int)NT, nblocks,
           1e3 * ms / reps, bytes / (ms * 1e6), bytes * nblocks / (ms * 1e9));
}

int main() {
    float* buf;
    const size_t n = 256 * 4 * kFloats;
    if (hipMalloc(&buf, n * sizeof(float)) != hipSuccess) return 1;
    hipMemset(buf, 0, n * sizeof(float));
    for (int nb : {1, 256}) {
        run<1, true>(buf, nb);
        run<2, true>(buf, nb);
        run<3, true>(buf, nb);
        run<2, false>(buf, nb);
    }
    return 0;
}

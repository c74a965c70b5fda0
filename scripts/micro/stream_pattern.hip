// Diagnostic micro-benchmark (not product code): per-CU streaming rate of the weight-gradient epilogue's
// access patterns.  Every workgroup (8 waves) reads 4 arrays of [200][200] fp32 and writes them back, from
// its own 1.28 MB region (256 workgroups -> 330 MB, larger than the caches), 16 B per lane per instruction.
//   pattern 0: one instruction = 16 rows x 64 B  (tile t), the other 64 B half of each 128 B line much later
//   pattern 1: one instruction = 16 rows x 64 B, the two halves in back-to-back instructions
//   pattern 2: one instruction = 8 rows x 128 B
//   pattern 3: fully contiguous 1 KB per instruction
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_pattern.hip -o scripts/micro/stream_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float* base, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float* A[4];
    for (int a = 0; a < 4; a++) A[a] = base + ((size_t)blockIdx.x * 4 + a) * 40960;
    const int N = 200;
    for (int rep = 0; rep < reps; rep++) {
        if (PAT == 3) {
            for (int off = tid * 4; off < 40000; off += 512 * 4 * 7) {
                f32x4 v[4][7];
                for (int q = 0; q < 7; q++) {
                    const int o = off + q * 2048 < 40000 ? off + q * 2048 : 0;
                    for (int a = 0; a < 4; a++) v[a][q] = *reinterpret_cast<const f32x4*>(&A[a][o]);
                }
                for (int q = 0; q < 7; q++)
                    if (off + q * 2048 < 40000)
                        for (int a = 0; a < 4; a++) *reinterpret_cast<f32x4*>(&A[a][off + q * 2048]) = v[a][q] * 1.0001f;
            }
            continue;
        }
        // wave w owns column tiles 2w, 2w+1 (32 columns = 128 B); 13 row tiles in chunks of 7
        const int halves = (PAT == 0) ? 2 : 1;
        for (int hh = 0; hh < halves; hh++)
            for (int m0 = 0; m0 < 13; m0 += (PAT == 0 ? 7 : 4)) {
                constexpr int MC = (PAT == 0) ? 7 : 4;
                constexpr int NI = (PAT == 0) ? 1 : 2;
                f32x4 v[4][MC][NI];
                size_t p[MC][NI];
                bool ok[MC][NI];
                for (int q = 0; q < MC; q++)
                    for (int i = 0; i < NI; i++) {
                        int row, col;
                        if (PAT == 2) {       // 8 rows x 128 B per instruction: lanes 0..7 of a row are contiguous
                            const int r8 = lane >> 3, c8 = lane & 7;
                            row = 16 * (m0 + q) + 8 * i + r8;
                            col = 32 * wave + 4 * c8;
                        } else {
                            const int t = 2 * wave + (PAT == 0 ? hh : i);
                            row = 16 * (m0 + q) + c;
                            col = 16 * t + 4 * g;
                        }
                        ok[q][i] = row < 200 && col < N && (m0 + q) < 13;
                        p[q][i] = ok[q][i] ? (size_t)row * N + col : 0;
                        for (int a = 0; a < 4; a++) v[a][q][i] = *reinterpret_cast<const f32x4*>(&A[a][p[q][i]]);
                    }
                for (int q = 0; q < MC; q++)
                    for (int i = 0; i < NI; i++)
                        if (ok[q][i])
                            for (int a = 0; a < 4; a++) *reinterpret_cast<f32x4*>(&A[a][p[q][i]]) = v[a][q][i] * 1.0001f;
            }
    }
}

template <int PAT>
void run(float* buf, int blocks) {
    const int reps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(512), 0, 0, buf, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(512), 0, 0, buf, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * reps * 2.0 * 4 * 40000 * 4;
    printf("pattern %d, %3d workgroups: %.1f GB/s per CU, %.2f TB/s total, %.1f us per 1.28 MB pass\n", PAT, blocks,
           bytes / blocks / (ms * 1e6), bytes / (ms * 1e9), ms * 1e3 / reps);
}

int main() {
    float* buf;
    hipMalloc(&buf, (size_t)256 * 4 * 40960 * 4);
    hipMemset(buf, 0, (size_t)256 * 4 * 40960 * 4);
    for (int blocks : {16, 256}) { run<0>(buf, blocks); run<1>(buf, blocks); run<2>(buf, blocks); run<3>(buf, blocks); }
    return 0;
}

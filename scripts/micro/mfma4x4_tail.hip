// Micro-check (GPU box): v_mfma_f32_4x4x1_16b_f32 as the "ragged batch tile" of the k-loops (mfma_blocks.h, TAIL4).
// Lane l = 16 g + c of a wave holds, exactly as for v_mfma_f32_16x16x4_f32, the weight W[4g+s][c] of step s; the A operand
// is X[c & 3][4g+s] (four batch rows instead of sixteen).  Block b = l / 4 = (g, c / 4) of the instruction then
// accumulates rows 0..3 x columns 4(c/4)..+3 over the k's of lane group g; a butterfly over g completes the sum and
// leaves rows 0..3 of column c in every lane -- the layout lanes g == 0 of a 16x16x4 accumulator have.
// Also times both instructions back to back (cycles per instruction on one wave).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4x4_tail scripts/micro/mfma4x4_tail.hip && /tmp/mfma4x4_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void tail_tile(const float* X /* [4][16] */, const float* W /* [16][16] */, float* out /* [4][16] */) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < 4; s++)
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(X[(c & 3) * 16 + 4 * g + s], W[(4 * g + s) * 16 + c], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) {
        acc[r] += __shfl_xor(acc[r], 16, 64);
        acc[r] += __shfl_xor(acc[r], 32, 64);
    }
    if (g == 0)
        for (int r = 0; r < 4; r++) out[r * 16 + c] = acc[r];
}

template <int KIND>
__global__ void rate(float* out, long long* cyc, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = threadIdx.x * 0.001f, y = 1.0f + x;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}

int main() {
    float hX[64], hW[256], ref[64], got[64];
    srand(1);
    for (auto& v : hX) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hW) v = rand() / (float)RAND_MAX - 0.5f;
    for (int r = 0; r < 4; r++)
        for (int n = 0; n < 16; n++) {
            double s = 0;
            for (int k = 0; k < 16; k++) s += (double)hX[r * 16 + k] * hW[k * 16 + n];
            ref[r * 16 + n] = (float)s;
        }
    float *dX, *dW, *dO;
    long long* dC;
    hipMalloc(&dX, sizeof hX); hipMalloc(&dW, sizeof hW); hipMalloc(&dO, 4096); hipMalloc(&dC, 16);
    hipMemcpy(dX, hX, sizeof hX, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW, sizeof hW, hipMemcpyHostToDevice);
    tail_tile<<<1, 64>>>(dX, dW, dO);
    hipMemcpy(got, dO, sizeof got, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 64; i++) worst = fmax(worst, fabs(got[i] - ref[i]));
    printf("tail tile: max |4x4x1 path - fp64 reference| = %.3g  (%s)\n", worst, worst < 1e-5 ? "OK" : "MISMATCH");
    const int iters = 100000;
    rate<0><<<1, 64>>>(dO, dC, iters);
    rate<1><<<1, 64>>>(dO, dC, iters);
    long long cyc[2];
    hipMemcpy(cyc, dC, sizeof cyc, hipMemcpyDeviceToHost);
    printf("cycles per instruction (clock64 ticks, one wave): 16x16x4 %.2f   4x4x1 %.2f\n", cyc[0] / (4.0 * iters),
           cyc[1] / (4.0 * iters));
    return worst < 1e-5 ? 0 : 1;
}

// Diagnostic micro-benchmark (not product code): how many cycles per v_mfma_f32_16x16x4_f32 does a CU sustain
// with the fwd_gemm instruction mix?  Variant 3 = as 2 but every block streams its own weights (HBM, not L2).  Variants: 0 = MFMAs only (operands in registers), 1 = + ds_read_b128 A
// fragments per chunk, 2 = + global dword B loads per chunk (prefetched one chunk ahead).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float* W, float* out, long long* cyc, int chunks) {
    __shared__ __attribute__((aligned(16))) float hbuf[112 * 200];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 112 * 200; i += WAVES * 64) hbuf[i] = (float)(i % 13) * 0.01f;
    __syncthreads();
    f32x4 acc[7][2];
    for (int m = 0; m < 7; m++) for (int i = 0; i < 2; i++) acc[m][i] = f32x4{0, 0, 0, 0};
    float b0[2][4], b1[2][4];
    for (int i = 0; i < 2; i++) for (int s = 0; s < 4; s++) b0[i][s] = b1[i][s] = 0.001f * (lane + s + i);
    f32x4 av[7];
    for (int m = 0; m < 7; m++) av[m] = f32x4{0.1f * c, 0.2f, 0.3f, 0.4f + g};
    const int wave = tid >> 6;
    const long long t0 = clock64();
    for (int ch = 0; ch < chunks; ch++) {
        const int kc = (ch % 12) * 16;
        if (VAR >= 2) {
            for (int i = 0; i < 2; i++)
                for (int s = 0; s < 4; s++)
                    b1[i][s] = (VAR >= 3 ? W + ((size_t)blockIdx.x * 4 + (ch / 12) % 4) * 40000 : W)
                        [(size_t)(kc + 4 * g + s) * 200 + 16 * (2 * wave + i) % 200 + c];
        }
        if (VAR >= 1) {
            for (int m = 0; m < 7; m++) av[m] = *reinterpret_cast<const f32x4*>(&hbuf[(16 * m + c) * 200 + kc + 4 * g]);
        }
        for (int s = 0; s < 4; s++)
            for (int i = 0; i < 2; i++)
                for (int m = 0; m < 7; m++)
                    acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][s], b0[i][s], acc[m][i], 0, 0, 0);
        if (VAR >= 2)
            for (int i = 0; i < 2; i++) for (int s = 0; s < 4; s++) b0[i][s] = b1[i][s];
    }
    const long long t1 = clock64();
    float sum = 0;
    for (int m = 0; m < 7; m++) for (int i = 0; i < 2; i++) for (int r = 0; r < 4; r++) sum += acc[m][i][r];
    out[blockIdx.x * WAVES * 64 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR, int WAVES>
void run(const float* W, float* out, long long* cyc, int blocks) {
    const int chunks = 2000;
    hipLaunchKernelGGL((k<VAR, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, W, out, cyc, chunks);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    const double mfma_per_simd = (double)chunks * 56 * WAVES / 4.0;
    printf("variant %d, %d waves/CU, %d blocks: %.0f cycles, %.1f cycles per MFMA per SIMD (32 = pipe peak)\n", VAR,
           WAVES, blocks, avg, avg / mfma_per_simd);
}

int main() {
    float *W, *out; long long* cyc;
    hipMalloc(&W, (size_t)40000 * 4 * 4 * 256); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(W, 0, (size_t)40000 * 4 * 4 * 256);
    for (int blocks : {1, 256}) {
        run<0, 4>(W, out, cyc, blocks); run<0, 8>(W, out, cyc, blocks);
        run<1, 4>(W, out, cyc, blocks); run<1, 8>(W, out, cyc, blocks);
        run<2, 4>(W, out, cyc, blocks); run<2, 8>(W, out, cyc, blocks);
        run<3, 4>(W, out, cyc, blocks); run<3, 8>(W, out, cyc, blocks);
    }
    return 0;
}

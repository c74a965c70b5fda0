// Diagnostic micro-benchmark (not product code): the roofs bench.py prices against, measured on THIS box, and the
// calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of the fused update kernels.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/peaks.hip -o scripts/micro/peaks && scripts/micro/peaks [bytes]
// Prints one JSON line (scripts/peaks_summary.py merges it with the --pmc passes into profiles/r03_peaks.json).
//   mfma16 / mfma32   v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32, independent accumulators, operands in registers,
//                     two waves per SIMD on every CU (the update kernels' occupancy) -> TFLOP/s
//   copy_f4           float4 copy of a 1 GiB buffer (read + written bytes) -> GB/s;  read_f4: the read half alone
//   calib_kloop_dword the forward k-loop's weight stream (mfma_blocks.h fwd_loop): per 1 KB tile-blocked block every lane
//                     loads the four dwords 16 B apart that it feeds to four MFMA steps -- each byte of the buffer once
//   calib_b128        the backward loop's / Adam epilogue's stream: one 16-byte load per lane, 1 KB per instruction
//   calib_rw_b128     the Adam epilogue: 16 B per lane read, modified, written back
// Under `rocprofv3 --pmc FETCH_SIZE` (and, in its own pass, WRITE_SIZE) the counter value of each calib_* dispatch over
// its known byte count is the factor profiles/r03_peaks.json records.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void mfma16(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 0.001f * lane, b = 1.0f - 0.002f * lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; i++) s += acc[i];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(512) void mfma32(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = 0.001f * lane, b = 1.0f - 0.002f * lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_f4(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void read_f4(const f32x4* __restrict__ src, float* out, size_t n4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += src[i];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.0f;
}

// one wave per 1 KB block at a time; blocks dealt to the waves of the grid in order
__global__ __launch_bounds__(512) void calib_kloop_dword(const float* __restrict__ W, float* out, size_t nblk) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int lofs = ((((c >> 2) << 4) + 4 * g) << 2) + (c & 3);      // mfma_blocks.h fwd_loop
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 8;
    float s = 0.f;
    for (size_t b = wave; b < nblk; b += nw) {
        const float* wp = W + (b << 8) + lofs;
#pragma unroll
        for (int k = 0; k < 4; k++) s += wp[4 * k];
    }
    if (s == 123.456f) out[0] = 1.0f;
}

__global__ __launch_bounds__(512) void calib_b128(const float* __restrict__ W, float* out, size_t nblk) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 8;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t b = wave; b < nblk; b += nw) s += *reinterpret_cast<const f32x4*>(W + (b << 8) + (lane << 2));
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.0f;
}

__global__ __launch_bounds__(512) void calib_rw_b128(float* __restrict__ W, size_t nblk) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 8;
    for (size_t b = wave; b < nblk; b += nw) {
        f32x4* p = reinterpret_cast<f32x4*>(W + (b << 8) + (lane << 2));
        *p = *p * 1.0001f;
    }
}

template <class F>
static float timed(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : ((size_t)1 << 30);
    const size_t n4 = bytes / 16, nblk = bytes / 1024;
    float *a, *b, *out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 4 << 20));
    CHECK(hipMemset(a, 0x11, bytes)); CHECK(hipMemset(b, 0, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    // one 512-thread workgroup per CU = two waves per SIMD
    const float ms16 = timed([&] { hipLaunchKernelGGL(mfma16, dim3(cus), dim3(512), 0, 0, out, iters); }, 3);
    const float ms32 = timed([&] { hipLaunchKernelGGL(mfma32, dim3(cus), dim3(512), 0, 0, out, iters); }, 3);
    const double f16 = (double)cus * 8 * iters * 8 * 2.0 * 16 * 16 * 4 / (ms16 * 1e-3);
    const double f32 = (double)cus * 8 * iters * 4 * 2.0 * 32 * 32 * 2 / (ms32 * 1e-3);
    const float msc = timed([&] { hipLaunchKernelGGL(copy_f4, dim3(cus * 16), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)b, n4); }, 5);
    const float msr = timed([&] { hipLaunchKernelGGL(read_f4, dim3(cus * 16), dim3(256), 0, 0, (const f32x4*)a, out, n4); }, 5);
    const float msd = timed([&] { hipLaunchKernelGGL(calib_kloop_dword, dim3(cus * 2), dim3(512), 0, 0, a, out, nblk); }, 3);
    const float msb = timed([&] { hipLaunchKernelGGL(calib_b128, dim3(cus * 2), dim3(512), 0, 0, a, out, nblk); }, 3);
    const float msw = timed([&] { hipLaunchKernelGGL(calib_rw_b128, dim3(cus * 2), dim3(512), 0, 0, b, nblk); }, 3);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"buffer_bytes\": %zu, "
           "\"mfma_f32_16x16x4_tflops\": %.2f, \"mfma_f32_32x32x2_tflops\": %.2f, "
           "\"copy_f4_gbs\": %.1f, \"read_f4_gbs\": %.1f, "
           "\"calib_kloop_dword_gbs\": %.1f, \"calib_b128_gbs\": %.1f, \"calib_rw_b128_gbs\": %.1f, "
           "\"calib_launches\": {\"calib_kloop_dword\": 4, \"calib_b128\": 4, \"calib_rw_b128\": 4, \"copy_f4\": 6, \"read_f4\": 6}}\n",
           prop.gcnArchName, cus, prop.clockRate / 1000, bytes, f16 / 1e12, f32 / 1e12,
           2.0 * bytes / (msc * 1e6), bytes / (msr * 1e6), bytes / (msd * 1e6), bytes / (msb * 1e6), 2.0 * bytes / (msw * 1e6));
    return 0;
}

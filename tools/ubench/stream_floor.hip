// What does the memory system alone allow for the headline workload's traffic shape?  Reads 1024 x 4096 cf32
// (32 MiB) and writes 1024 x 4096 f32 (16 MiB) with no arithmetic to speak of, in the launch shapes the FFT
// kernels use.  The result is the practical floor for any fused multiply->fft->amplitude->range kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_floor.hip -o tools/ubench/stream_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// V complex samples per thread per iteration, grid-stride
template <int V>
__global__ __launch_bounds__(256) void flat(const float2* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / V; i += (size_t)gridDim.x * 256) {
        float2 v[V];
        for (int j = 0; j < V; ++j) v[j] = in[i + j * (n / V)];
        for (int j = 0; j < V; ++j) out[i + j * (n / V)] = v[j].x + v[j].y;
    }
}
// the pipe kernel's shape: one 4096-point row per 512-thread workgroup, 8 strided points per thread,
// persistent over rows with the next row's loads issued before the current row's stores
__global__ __launch_bounds__(512, 2) void rows(const float2* __restrict__ in, float* __restrict__ out, int nrows) {
    int r = blockIdx.x;
    float2 cur[8], nxt[8];
    if (r < nrows)
        for (int j = 0; j < 8; ++j) cur[j] = in[(size_t)r * 4096 + threadIdx.x + 512 * j];
    for (; r < nrows; r += gridDim.x) {
        const int rn = r + gridDim.x;
        if (rn < nrows)
            for (int j = 0; j < 8; ++j) nxt[j] = in[(size_t)rn * 4096 + threadIdx.x + 512 * j];
        for (int j = 0; j < 8; ++j) out[(size_t)r * 4096 + threadIdx.x + 512 * j] = cur[j].x + cur[j].y;
        for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    }
}
__global__ void empty_kernel() {}

template <class F>
void timeit(const char* name, F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-34s %7.2f us per launch (back to back)  -> %.2f TB/s of 48 MiB\n", name, us, 50331648.0 / us * 1e-6);
}
int main() {
    const size_t n = 1024ull * 4096;
    float2* in; float* out;
    (void)hipMalloc(&in, n * 8); (void)hipMalloc(&out, n * 4);
    (void)hipMemset(in, 0, n * 8);
    timeit("empty kernel", [&] { empty_kernel<<<1, 64>>>(); });
    timeit("flat<1> grid 16384", [&] { flat<1><<<16384, 256>>>(in, out, n); });
    timeit("flat<2> grid 8192", [&] { flat<2><<<8192, 256>>>(in, out, n); });
    timeit("flat<4> grid 4096", [&] { flat<4><<<4096, 256>>>(in, out, n); });
    timeit("flat<8> grid 2048", [&] { flat<8><<<2048, 256>>>(in, out, n); });
    timeit("flat<4> grid 2048 (2 iters)", [&] { flat<4><<<2048, 256>>>(in, out, n); });
    timeit("rows grid 1024 (1 row each)", [&] { rows<<<1024, 512>>>(in, out, 1024); });
    timeit("rows grid 512 (persistent, 2 rows)", [&] { rows<<<512, 512>>>(in, out, 1024); });
    timeit("rows grid 256 (persistent, 4 rows)", [&] { rows<<<256, 512>>>(in, out, 1024); });
    return 0;
}

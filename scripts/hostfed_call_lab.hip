// Lab harness (not product; prepared at the end of round 2, NOT YET RUN): where do the ~175 us of a small host-fed call go?
// A 128-item config-2 call moves 1.0 MB in and 1.8 MB out (~60 us of PCIe time) and runs three short kernels, yet takes
// 0.175 ms through baz_music_process on page-locked buffers (profiles/r02_flowgraph_model_rates.txt).  This times the
// ingredients one by one on page-locked host memory, 2,000 repetitions each:
//   the API calls the path makes per call (hipPointerGetAttributes x2, hipHostGetDevicePointer x3),
//   one / three back-to-back kernel launches + hipStreamSynchronize (empty kernels: pure submission + completion latency),
//   the same three launches replayed as a hipGraph,
//   a kernel that reads 1 MB of host memory and one that writes 1.8 MB of host memory (the PCIe part, zero-copy),
//   the same bytes as hipMemcpyAsync H2D / D2H + synchronize (the copy path).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/hostfed_call_lab scripts/hostfed_call_lab.hip && scripts/hostfed_call_lab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1024) *p = 1; }

__global__ __launch_bounds__(256) void read_host(const v4f* __restrict__ src, size_t n16, float* __restrict__ sink)
{
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(src + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = 1.0f;
}

__global__ __launch_bounds__(256) void write_host(v4f* __restrict__ dst, size_t n16)
{
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, dst + i);
}

template <class F>
static void timeit(const char* name, int reps, F body)
{
    for (int i = 0; i < 50; ++i) body();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) body();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%-86s %8.1f us\n", name, us);
    fflush(stdout);
}

int main()
{
    const size_t in_bytes = 128u * 8192u, out_bytes = 128u * 14400u;     // one 128-item config-2 call
    const int reps = 2000;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char *h_in, *h_out, *d_in, *d_out;
    float* sink;
    CK(hipHostMalloc((void**)&h_in, in_bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_out, out_bytes, hipHostMallocDefault));
    CK(hipMalloc((void**)&d_in, in_bytes));
    CK(hipMalloc((void**)&d_out, out_bytes));
    CK(hipMalloc((void**)&sink, 64));
    for (size_t i = 0; i < in_bytes; ++i) h_in[i] = (char)i;
    void *z_in = nullptr, *z_out = nullptr;
    CK(hipHostGetDevicePointer(&z_in, h_in, 0));
    CK(hipHostGetDevicePointer(&z_out, h_out, 0));

    timeit("hipPointerGetAttributes x2 + hipHostGetDevicePointer x3 (what a call looks up)", reps, [&] {
        hipPointerAttribute_t a;
        void* p;
        (void)hipPointerGetAttributes(&a, h_in);
        (void)hipPointerGetAttributes(&a, h_out);
        (void)hipHostGetDevicePointer(&p, h_in, 0);
        (void)hipHostGetDevicePointer(&p, h_out, 0);
        (void)hipHostGetDevicePointer(&p, h_in + 64, 0);
    });
    timeit("hipStreamSynchronize on an idle stream", reps, [&] { CK(hipStreamSynchronize(s)); });
    timeit("1 empty launch + synchronize", reps, [&] {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        CK(hipStreamSynchronize(s));
    });
    timeit("3 empty launches + synchronize", reps, [&] {
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        CK(hipStreamSynchronize(s));
    });
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        timeit("the same 3 launches as one hipGraphLaunch + synchronize", reps, [&] {
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
        });
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    timeit("kernel reads 1.0 MB of page-locked host memory + synchronize (zero-copy in)", reps, [&] {
        hipLaunchKernelGGL(read_host, dim3(256), dim3(256), 0, s, (const v4f*)z_in, in_bytes / 16, sink);
        CK(hipStreamSynchronize(s));
    });
    timeit("kernel writes 1.8 MB of page-locked host memory + synchronize (zero-copy out)", reps, [&] {
        hipLaunchKernelGGL(write_host, dim3(256), dim3(256), 0, s, (v4f*)z_out, out_bytes / 16);
        CK(hipStreamSynchronize(s));
    });
    timeit("read 1.0 MB, empty, write 1.8 MB: three launches + synchronize (the zero-copy call's skeleton)", reps, [&] {
        hipLaunchKernelGGL(read_host, dim3(256), dim3(256), 0, s, (const v4f*)z_in, in_bytes / 16, sink);
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        hipLaunchKernelGGL(write_host, dim3(256), dim3(256), 0, s, (v4f*)z_out, out_bytes / 16);
        CK(hipStreamSynchronize(s));
    });
    timeit("hipMemcpyAsync H2D 1.0 MB + synchronize", reps, [&] {
        CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
    });
    timeit("hipMemcpyAsync D2H 1.8 MB + synchronize", reps, [&] {
        CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
    });
    timeit("H2D 1.0 MB, 3 empty launches, D2H 2 KB, D2H 1.8 MB + synchronize (the copy call's skeleton)", reps, [&] {
        CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s));
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        CK(hipMemcpyAsync(h_in, d_in, 2048, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
    });
    return 0;
}

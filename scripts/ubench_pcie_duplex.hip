// ubench_pcie_duplex.hip -- does a kernel that READS page-locked host memory run beside one that WRITES page-locked host memory at the sum
// of their rates (the link is full duplex) or do they share one budget?  Round 5, VERDICT r4 task 4: decides whether overlapping the two
// directions inside a host-fed call can pay at all.  Also: the same with the copy engines (hipMemcpyAsync on two streams).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_pcie_duplex scripts/ubench_pcie_duplex.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_host(const v4f* __restrict__ src, size_t n4, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
        const v4f v = __builtin_nontemporal_load(src + i);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void write_host(v4f* __restrict__ dst, size_t n4)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
        const v4f v = {1.f, 2.f, 3.f, (float)i};
        __builtin_nontemporal_store(v, dst + i);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t MB = 64, bytes = MB << 20, n4 = bytes / 16;
    v4f *hin, *hout, *din, *dout; float* sink;
    CK(hipHostMalloc((void**)&hin, bytes, hipHostMallocDefault)); CK(hipHostMalloc((void**)&hout, bytes, hipHostMallocDefault));
    for (size_t i = 0; i < n4; ++i) hin[i] = v4f{1, 2, 3, 4};
    v4f *zin, *zout; CK(hipHostGetDevicePointer((void**)&zin, hin, 0)); CK(hipHostGetDevicePointer((void**)&zout, hout, 0));
    CK(hipMalloc((void**)&din, bytes)); CK(hipMalloc((void**)&dout, bytes)); CK(hipMalloc((void**)&sink, 64));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int blocks : {8, 32, 128}) {
        auto run = [&](bool r, bool w, bool engines) {
            double best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                if (engines) {
                    if (r) CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s1));
                    if (w) CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s2));
                } else {
                    if (r) hipLaunchKernelGGL(read_host, dim3(blocks), dim3(256), 0, s1, zin, n4, sink);
                    if (w) hipLaunchKernelGGL(write_host, dim3(blocks), dim3(256), 0, s2, zout, n4);
                }
                CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
                const double dt = now() - t0;
                if (dt < best) best = dt;
            }
            return best;
        };
        const double tr = run(true, false, false), tw = run(false, true, false), tb = run(true, true, false);
        printf("kernels, %3d workgroups each, %zu MiB per direction: read alone %.3f ms (%.1f GB/s)  write alone %.3f ms (%.1f GB/s)  both %.3f ms (%.1f GB/s in + out; sum of the two alone %.3f ms)\n",
               blocks, MB, tr * 1e3, bytes / tr / 1e9, tw * 1e3, bytes / tw / 1e9, tb * 1e3, 2.0 * bytes / tb / 1e9, (tr + tw) * 1e3);
        if (blocks == 128) {
            const double er = run(true, false, true), ew = run(false, true, true), eb = run(true, true, true);
            printf("copy engines, %zu MiB per direction: H2D alone %.3f ms (%.1f GB/s)  D2H alone %.3f ms (%.1f GB/s)  both %.3f ms (%.1f GB/s in + out; sum %.3f ms)\n",
                   MB, er * 1e3, bytes / er / 1e9, ew * 1e3, bytes / ew / 1e9, eb * 1e3, 2.0 * bytes / eb / 1e9, (er + ew) * 1e3);
        }
    }
    return 0;
}

// Lab micro-benchmark (not part of the product): what HBM3E on MI355X sustains for the access patterns of the
// MUSIC-DoA pipeline -- pure streaming reads (covariance), pure streaming writes (spectrum), and both at once.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_hbm scripts/ubench_hbm.hip
// Patterns:
//   read_x4      every lane loads 16 B, a wave-instruction covers 1 KiB contiguous
//   read_dw      every lane loads 4 B, a wave-instruction covers 256 B contiguous (the covariance kernel's loads)
//   write_x4     every lane stores 16 B, 1 KiB contiguous per wave-instruction; policies plain / nt / sc0 sc1 nt
//   write_rows   the scan's pattern: a wave owns 16 rows (items) of `pitch` bytes, per step 4 store instructions,
//                each writing 256 B contiguous into 4 different rows; pitch 14400 (cfg2: odd rows start on a 64-B
//                boundary) vs 14336 (every row 256-B aligned)
//   mixed        half of the workgroups stream-read, the other half stream-write, one launch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_x4(const v4f* __restrict__ p, size_t n16, float* sink)
{
    v4f acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = 1.0f;
}

__global__ __launch_bounds__(256) void read_dw(const float* __restrict__ p, size_t n4, float* sink)
{
    float acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += p[i];
    if (acc == 1.2345f) sink[0] = 1.0f;
}

// POLICY 0 plain, 1 nt (global_store ... nt), 2 sc0 sc1 nt, 3 sc0 sc1, 4 nt via buffer store (aux = 2)
template <int POLICY>
__device__ __forceinline__ void store16(char* base, uint32_t off, v4f v)
{
    if constexpr (POLICY == 0) *reinterpret_cast<v4f*>(base + off) = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(base + off));
    else {   // wave-uniform resource over the whole buffer, per-lane 32-bit byte offset
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFF, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, (int)off, 0,
                                               POLICY == 2 ? (1 | 2 | 16) : (POLICY == 3 ? (1 | 16) : 2));
    }
}

template <int POLICY>
__global__ __launch_bounds__(256) void write_x4(v4f* __restrict__ p, size_t n16)
{
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) store16<POLICY>(reinterpret_cast<char*>(p), (uint32_t)(i * 16), v);
}

// the scan's store pattern: block = 4 waves = 64 rows; wave w owns rows 16w..16w+15 of the block; nsplit ranges of steps
template <int POLICY>
__global__ __launch_bounds__(256) void write_rows(char* __restrict__ base, uint32_t rows, uint32_t pitch, uint32_t nsteps,
                                                  uint32_t nsplit, uint32_t spin)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const uint32_t split = blockIdx.x % nsplit;
    const uint32_t row0 = ((blockIdx.x / nsplit) * 4 + wave) * 16;
    const uint32_t s0 = (uint32_t)(((uint64_t)nsteps * split) / nsplit), s1 = (uint32_t)(((uint64_t)nsteps * (split + 1)) / nsplit);
    v4f v = {1.0f, 2.0f, 3.0f, (float)lane};
    for (uint32_t st = s0; st < s1; ++st) {
        for (uint32_t k = 0; k < spin; ++k) v[0] = __builtin_fmaf(v[0], 1.0000001f, 0.5f);   // stand-in for the step's arithmetic
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = row0 + g + 4 * r;
            if (row < rows && st * 256u + 16u * c + 16u <= pitch)
                store16<POLICY>(base, row * pitch + st * 256u + 16u * c, v);
        }
    }
}

// the scan's ROW-CLASS pattern (round 2): block = 4 waves x 16 rows of ONE class (rows nclass apart), every store a
// 256-B aligned window [256*st - 4*sh, +256) of its row (res = 3600 floats, nclass = 4, sh = 16*class floats)
// MAP (prepared at the end of round 2, not yet run): which 64-row group a workgroup writes.  Workgroups are dealt to the
// 8 XCDs round-robin (blockIdx % 8), so with MAP 0 (the scan's order) neighbouring groups are written by different XCDs
// at the same time.  MAP 1: every XCD owns one contiguous eighth of the rows; MAP 2: XCD-owned runs of 32 groups (2,048
// rows, 29 MB).  Does the order in which the chip visits the spectrum move the store rate (5.1-5.3 TB/s in this pattern
// against 5.5-6.0 for aligned 14,336-B rows)?
template <int POLICY, bool BUFFER, int MAP = 0>
__global__ __launch_bounds__(256) void write_rows_class(char* __restrict__ base, uint32_t rows, uint32_t nsplit, uint32_t spin)
{
    const uint32_t res = 3600, nclass = 4, pitch = 14400;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const uint32_t rpc = rows / nclass;
    uint32_t split = blockIdx.x % nsplit;
    uint32_t grp = blockIdx.x / nsplit;
    {
        const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;   // seq-th block of its XCD
        const uint32_t bpx = gridDim.x >> 3;                                                        // blocks per XCD
        if (MAP == 1 && (gridDim.x & 7u) == 0) {          // XCD x: blocks [x * bpx, (x+1) * bpx) of the original order
            const uint32_t b = xcd * bpx + seq;
            grp = b / nsplit;
            split = b % nsplit;
        } else if (MAP == 2 && (gridDim.x & 7u) == 0 && (bpx % (32u * nsplit)) == 0) {
            const uint32_t run = 32u * nsplit, r = seq / run, o = seq % run;
            const uint32_t b = (r * 8u + xcd) * run + o;
            grp = b / nsplit;
            split = b % nsplit;
        }
    }
    const uint32_t p0 = (grp * 4 + wave) * 16;
    const uint32_t cls = p0 / rpc, j0 = p0 - cls * rpc;
    const uint32_t sh = (res * cls) & 63u;
    const uint32_t nsteps = (res + sh + 63) >> 6;
    const uint32_t s0 = (uint32_t)(((uint64_t)nsteps * split) / nsplit), s1 = (uint32_t)(((uint64_t)nsteps * (split + 1)) / nsplit);
    const uint32_t item0 = nclass * j0 + cls;
    v4f v = {1.0f, 2.0f, 3.0f, (float)lane};
    char* wbase = base + (size_t)item0 * pitch - 4 * sh;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, 0x7FFFFFFF, 0x00020000);
    for (uint32_t st = s0; st < s1; ++st) {
        for (uint32_t k = 0; k < spin; ++k) v[0] = __builtin_fmaf(v[0], 1.0000001f, 0.5f);
        const uint32_t bin = st * 64 + 4 * c - sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t off = (uint32_t)(g + 4 * r) * nclass * pitch + 16u * c;
            if (bin < res) {
                if constexpr (BUFFER)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, (int)off, (int)(st * 256u),
                                                           POLICY == 2 ? (1 | 2 | 16) : (POLICY == 1 ? 2 : 0));
                else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(wbase + off + st * 256u));
                else *reinterpret_cast<v4f*>(wbase + off + st * 256u) = v;
            }
        }
    }
}

// the row-class pattern with the stores of W consecutive steps gathered (in the scan: a 4 x 4 exchange between lane rows and
// registers, v_permlane16/32_swap): one store instruction = 1 KiB of 4 / W rows, W x 256 B contiguous per row, instead of
// 256 B of each of 4 rows.  Same bytes, same number of instructions; does the memory system prefer the longer runs?
template <int POLICY, int W>
__global__ __launch_bounds__(256) void write_rows_gather(char* __restrict__ base, uint32_t rows, uint32_t nsplit, uint32_t spin)
{
    const uint32_t res = 3600, nclass = 4, pitch = 14400;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rpc = rows / nclass;
    const uint32_t split = blockIdx.x % nsplit, grp = blockIdx.x / nsplit;
    const uint32_t p0 = (grp * 4 + wave) * 16;
    const uint32_t cls = p0 / rpc, j0 = p0 - cls * rpc;
    const uint32_t sh = (res * cls) & 63u;
    const uint32_t nsteps = (res + sh + 63) >> 6;
    const uint32_t ngr = (nsteps + W - 1) / W;
    const uint32_t g0 = (uint32_t)(((uint64_t)ngr * split) / nsplit), g1 = (uint32_t)(((uint64_t)ngr * (split + 1)) / nsplit);
    const uint32_t item0 = nclass * j0 + cls;
    v4f v = {1.0f, 2.0f, 3.0f, (float)lane};
    char* wbase = base + (size_t)item0 * pitch - 4 * sh;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, 0x7FFFFFFF, 0x00020000);
    constexpr int LPR = 16 * W;                 // lanes per row in one instruction
    const uint32_t sub = (uint32_t)lane / LPR, lo = (uint32_t)lane % LPR;
    for (uint32_t gq = g0; gq < g1; ++gq) {
        for (uint32_t k = 0; k < spin * W; ++k) v[0] = __builtin_fmaf(v[0], 1.0000001f, 0.5f);
        const uint32_t bin = gq * (64 * W) + 4 * lo - sh;
#pragma unroll
        for (int i = 0; i < 4 * W; ++i) {
            const uint32_t row = (uint32_t)i * (4 / W) + sub;
            const uint32_t off = row * nclass * pitch + 16u * lo;
            if (bin < res) {
                if constexpr (POLICY == 2)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, (int)off, (int)(gq * 256u * W), 1 | 2 | 16);
                else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(wbase + off + gq * 256u * W));
                else *reinterpret_cast<v4f*>(wbase + off + gq * 256u * W) = v;
            }
        }
    }
}

template <int POLICY>
__global__ __launch_bounds__(256) void mixed(const v4f* __restrict__ src, size_t nr16, v4f* __restrict__ dst, size_t nw16, float* sink)
{
    const uint32_t half = gridDim.x / 2;
    if (blockIdx.x & 1) {
        const uint32_t b = blockIdx.x >> 1;
        v4f acc = {0, 0, 0, 0};
        for (size_t i = (size_t)b * 256 + threadIdx.x; i < nr16; i += (size_t)half * 256) acc += __builtin_nontemporal_load(src + i);
        if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = 1.0f;
    } else {
        const uint32_t b = blockIdx.x >> 1;
        const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
        for (size_t i = (size_t)b * 256 + threadIdx.x; i < nw16; i += (size_t)half * 256) store16<POLICY>(reinterpret_cast<char*>(dst), (uint32_t)(i * 16), v);
    }
}

template <class F>
void timeit(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 2) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("%-64s %8.3f ms (min %.3f)  %6.2f TB/s\n", name, ms[ms.size() / 2], ms[0], bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const size_t RD = 2147483648ull;                    // covariance input of the bench step: 262,144 x 8 KiB
    const uint32_t rows = 262144, pitchA = 14400, pitchB = 14336;
    const size_t WR = (size_t)rows * pitchA;            // spectrum of the bench step: 3.77 GB
    char *src, *dst; float* sink;
    CK(hipMalloc((void**)&src, RD)); CK(hipMalloc((void**)&dst, WR + 4096)); CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(src, 0, RD)); CK(hipMemset(dst, 0, WR));
    const int grids[] = {2048, 8192};
    for (int gsz : grids) {
        char nm[128];
        snprintf(nm, sizeof nm, "read_x4 2.1 GB grid %d", gsz);
        timeit(nm, (double)RD, [&] { hipLaunchKernelGGL(read_x4, dim3(gsz), dim3(256), 0, 0, (const v4f*)src, RD / 16, sink); });
        snprintf(nm, sizeof nm, "read_dw 2.1 GB grid %d (8 loads in flight per lane)", gsz);
        timeit(nm, (double)RD, [&] { hipLaunchKernelGGL(read_dw, dim3(gsz), dim3(256), 0, 0, (const float*)src, RD / 4, sink); });
        snprintf(nm, sizeof nm, "write_x4 plain 3.8 GB grid %d", gsz);
        timeit(nm, (double)WR, [&] { hipLaunchKernelGGL(write_x4<0>, dim3(gsz), dim3(256), 0, 0, (v4f*)dst, WR / 16); });
        snprintf(nm, sizeof nm, "write_x4 nt 3.8 GB grid %d", gsz);
        timeit(nm, (double)WR, [&] { hipLaunchKernelGGL(write_x4<1>, dim3(gsz), dim3(256), 0, 0, (v4f*)dst, WR / 16); });
        snprintf(nm, sizeof nm, "write_x4 sc0 sc1 nt 3.8 GB grid %d", gsz);
        timeit(nm, (double)WR, [&] { hipLaunchKernelGGL(write_x4<2>, dim3(gsz), dim3(256), 0, 0, (v4f*)dst, WR / 16); });
        snprintf(nm, sizeof nm, "write_x4 sc0 sc1 3.8 GB grid %d", gsz);
        timeit(nm, (double)WR, [&] { hipLaunchKernelGGL(write_x4<3>, dim3(gsz), dim3(256), 0, 0, (v4f*)dst, WR / 16); });
    }
    const uint32_t nsteps = 57;
    for (uint32_t nsplit : {2u}) {
        const uint32_t blocks = (rows / 64) * nsplit;
        for (uint32_t spin : {0u, 64u, 256u}) {
            char nm[160];
            snprintf(nm, sizeof nm, "write_rows pitch 14400 plain      nsplit %u spin %u", nsplit, spin);
            timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL(write_rows<0>, dim3(blocks), dim3(256), 0, 0, dst, rows, pitchA, nsteps, nsplit, spin); });
            snprintf(nm, sizeof nm, "write_rows pitch 14400 nt         nsplit %u spin %u", nsplit, spin);
            timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL(write_rows<1>, dim3(blocks), dim3(256), 0, 0, dst, rows, pitchA, nsteps, nsplit, spin); });
            snprintf(nm, sizeof nm, "write_rows pitch 14400 sc0 sc1 nt nsplit %u spin %u", nsplit, spin);
            timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL(write_rows<2>, dim3(blocks), dim3(256), 0, 0, dst, rows, pitchA, nsteps, nsplit, spin); });
            snprintf(nm, sizeof nm, "write_rows pitch 14336 sc0 sc1 nt nsplit %u spin %u", nsplit, spin);
            timeit(nm, (double)rows * pitchB, [&] { hipLaunchKernelGGL(write_rows<2>, dim3(blocks), dim3(256), 0, 0, dst, rows, pitchB, 56, nsplit, spin); });
            snprintf(nm, sizeof nm, "write_rows pitch 14336 nt         nsplit %u spin %u", nsplit, spin);
            timeit(nm, (double)rows * pitchB, [&] { hipLaunchKernelGGL(write_rows<1>, dim3(blocks), dim3(256), 0, 0, dst, rows, pitchB, 56, nsplit, spin); });
            snprintf(nm, sizeof nm, "write_rows pitch 14336 plain      nsplit %u spin %u", nsplit, spin);
            timeit(nm, (double)rows * pitchB, [&] { hipLaunchKernelGGL(write_rows<0>, dim3(blocks), dim3(256), 0, 0, dst, rows, pitchB, 56, nsplit, spin); });
        }
    }
    for (uint32_t spin : {0u, 64u}) {
        const uint32_t blocks = (rows / 64) * 2;
        char nm[160];
        snprintf(nm, sizeof nm, "write_rows_class (scan pattern) plain  global_store  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<0, false>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class (scan pattern) plain  buffer_store  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<0, true>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class (scan pattern) nt     global_store  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<1, false>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class (scan pattern) nt     buffer_store  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<1, true>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class (scan pattern) sc0 sc1 nt buffer_store spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<2, true>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class nt global_store, XCD owns a contiguous eighth  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<1, false, 1>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class nt global_store, XCD owns runs of 32 groups     spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<1, false, 2>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class plain global_store, XCD owns a contiguous eighth spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<0, false, 1>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_class plain global_store, XCD owns runs of 32 groups    spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_class<0, false, 2>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
    }
    for (uint32_t spin : {0u, 64u}) {
        const uint32_t blocks = (rows / 64) * 2;
        char nm[160];
        snprintf(nm, sizeof nm, "write_rows_gather W=1 (= the scan pattern) sc0 sc1 nt  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_gather<2, 1>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_gather W=2 (512 B per row and instruction) sc0 sc1 nt  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_gather<2, 2>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_gather W=4 (1 KiB of one row per instruction) sc0 sc1 nt  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_gather<2, 4>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_gather W=2 nt  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_gather<1, 2>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_gather W=4 nt  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_gather<1, 4>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
        snprintf(nm, sizeof nm, "write_rows_gather W=4 plain  spin %u", spin);
        timeit(nm, (double)rows * pitchA, [&] { hipLaunchKernelGGL((write_rows_gather<0, 4>), dim3(blocks), dim3(256), 0, 0, dst, rows, 2, spin); });
    }
    for (int gsz : {4096, 16384}) {
        char nm[128];
        snprintf(nm, sizeof nm, "mixed read 2.1 GB + write 3.8 GB plain grid %d", gsz);
        timeit(nm, (double)RD + (double)WR, [&] { hipLaunchKernelGGL(mixed<0>, dim3(gsz), dim3(256), 0, 0, (const v4f*)src, RD / 16, (v4f*)dst, WR / 16, sink); });
        snprintf(nm, sizeof nm, "mixed read 2.1 GB + write 3.8 GB sc0 sc1 nt grid %d", gsz);
        timeit(nm, (double)RD + (double)WR, [&] { hipLaunchKernelGGL(mixed<2>, dim3(gsz), dim3(256), 0, 0, (const v4f*)src, RD / 16, (v4f*)dst, WR / 16, sink); });
    }
    return 0;
}

// ubench_write_order.hip -- round 5: WHY does a plain fill (at::native fill: one short-lived workgroup per 4 - 16 KiB, dispatched in address
// order) write 3.78 GB at 6.9 TB/s when every persistent / grid-stride pattern of scripts/ubench_hbm.hip -- and the scan -- stays at 5.0 - 5.6?
// Variants: the fill as short-lived workgroups (U x 4 KiB each), the same bytes grid-stride, and the scan's row pattern (16 rows x 256 B per
// wave and step, rows 14,400 B apart) as short-lived workgroups of S steps in two dispatch orders.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_write_order scripts/ubench_write_order.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void fill_once(v4f* __restrict__ p, size_t n16)
{
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(v4f* __restrict__ p, size_t n16)
{
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
// rows of `pitch` bytes; a workgroup = 4 waves = 64 rows (wave w: rows 16 w .. 16 w + 15) x S steps of 256 B per row; lane: row (lane >> 4) + 4 r, r = 0 .. 3,
// 16 B at (lane & 15) * 16 of the step.  ORDER 0: consecutive workgroups walk the segments of one row group (segment fastest); 1: row group fastest.
// WPB waves per block variant: WAVES = 4 (64 rows) or 1 (16 rows).
template <int S, int ORDER, bool NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rows_once(char* __restrict__ base, uint32_t rows, uint32_t pitch, uint32_t nseg)
{
    const uint32_t ngroups = rows / (16 * WAVES);
    const uint32_t seg = ORDER == 0 ? blockIdx.x % nseg : blockIdx.x / ngroups;
    const uint32_t grp = ORDER == 0 ? blockIdx.x / nseg : blockIdx.x % ngroups;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t row0 = (grp * WAVES + wave) * 16 + (lane >> 4);
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t col = (seg * S + s) * 256 + (lane & 15) * 16;
        if (col + 16 <= pitch) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v4f* q = reinterpret_cast<v4f*>(base + (size_t)(row0 + 4 * r) * pitch + col);
                if (NT) __builtin_nontemporal_store(v, q); else *q = v;
            }
        }
    }
}

// The scan's real pattern: 256-B ALIGNED pieces.  Row r starts at r * 14400, i.e. 64 (r mod 4) bytes past a 256-B boundary: a wave takes 16 rows of ONE
// class (r mod 4 = k: rows g * 64 + 4 j + k), its pieces are [c0 + 256 (s - 1), c0 + 256 s) with c0 = (256 - 64 k) mod 256, clipped to the row.
// A workgroup = 4 waves = the 4 classes of 64 consecutive rows, S piece slots each (58 slots cover a row).  PERSIST > 0: instead of one-shot workgroups,
// PERSIST workgroups walk the (group, segment) list with a stride (the scan's persistent form).
template <int S, int ORDER, bool NT, int PERSIST>
__global__ __launch_bounds__(256) void rows_class(char* __restrict__ base, uint32_t rows, uint32_t pitch, uint32_t nseg)
{
    const uint32_t ngroups = rows / 64, total = ngroups * nseg;
    const uint32_t lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int c0 = (256 - 64 * (int)k) & 255;
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (uint32_t task = blockIdx.x; task < total; task += PERSIST ? PERSIST : total) {
        const uint32_t seg = ORDER == 0 ? task % nseg : task / ngroups;
        const uint32_t grp = ORDER == 0 ? task / nseg : task % ngroups;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int col = c0 + 256 * ((int)(seg * S + s) - 1) + (int)(lane & 15) * 16;
            if (col >= 0 && col + 16 <= (int)pitch) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t row = grp * 64 + 4 * ((lane >> 4) + 4 * r) + k;
                    v4f* q = reinterpret_cast<v4f*>(base + (size_t)row * pitch + col);
                    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
                }
            }
        }
    }
}

// A bench step's whole traffic in ONE pass of short-lived workgroups in address order: workgroup b writes 4 KiB x U of the spectrum and reads the matching share
// of the 2.1 GB input (8,192 B per 14,400 B written): what the memory system takes for these bytes when both streams are compact moving windows.
template <int U>
__global__ __launch_bounds__(256) void mixed_once(const v4f* __restrict__ src, size_t nr16, v4f* __restrict__ dst, size_t nw16, float* __restrict__ sink)
{
    const size_t wbase = (size_t)blockIdx.x * 256 * U;
    // reads: the same fraction of the input as this workgroup's fraction of the output
    const size_t r0 = (size_t)((double)wbase / (double)nw16 * (double)nr16), r1 = (size_t)((double)(wbase + 256 * U) / (double)nw16 * (double)nr16);
    float acc = 0.f;
    for (size_t i = r0 + threadIdx.x; i < r1 && i < nr16; i += 256) { const v4f v = __builtin_nontemporal_load(src + i); acc += v.x + v.y + v.z + v.w; }
    const v4f v = {1.0f, 2.0f, 3.0f, acc};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = wbase + threadIdx.x + (size_t)u * 256;
        if (i < nw16) dst[i] = v;
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <class F>
void timeit(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    if (getenv("UB_TRACE")) { printf("start %s\n", name); fflush(stdout); }
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 2) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("%-84s %8.3f ms (min %.3f)  %6.2f TB/s\n", name, ms[ms.size() / 2], ms[0], bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const uint32_t rows = 262144, pitch = 14400;
    const size_t WR = (size_t)rows * pitch, n16 = WR / 16;
    char* dst; CK(hipMalloc((void**)&dst, WR + 65536)); CK(hipMemset(dst, 0, WR));
    char nm[200];
    {
        const size_t RD = 2147483648ull;
        char* src; float* sink; CK(hipMalloc((void**)&src, RD)); CK(hipMalloc((void**)&sink, 64)); CK(hipMemset(src, 0, RD));
#define MIXED(U) snprintf(nm, sizeof nm, "a step's traffic, read 2.1 GB + write 3.8 GB, one workgroup per %d KiB written", 4 * U); \
        timeit(nm, (double)RD + (double)WR, [&] { hipLaunchKernelGGL((mixed_once<U>), dim3((unsigned)((n16 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (const v4f*)src, RD / 16, (v4f*)dst, n16, sink); });
        MIXED(1) MIXED(4) MIXED(16)
        CK(hipFree(src));
    }
#define FILL(U, NT) snprintf(nm, sizeof nm, "fill, one workgroup per %d KiB, %s", 4 * U, NT ? "nt" : "plain"); \
    timeit(nm, (double)WR, [&] { hipLaunchKernelGGL((fill_once<U, NT>), dim3((unsigned)((n16 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (v4f*)dst, n16); });
    FILL(1, false) FILL(1, true) FILL(4, false) FILL(4, true) FILL(16, false) FILL(16, true)
    for (int g : {2048, 8192}) {
        snprintf(nm, sizeof nm, "fill, grid-stride, %d workgroups, plain", g);
        timeit(nm, (double)WR, [&] { hipLaunchKernelGGL((fill_stride<false>), dim3(g), dim3(256), 0, 0, (v4f*)dst, n16); });
        snprintf(nm, sizeof nm, "fill, grid-stride, %d workgroups, nt", g);
        timeit(nm, (double)WR, [&] { hipLaunchKernelGGL((fill_stride<true>), dim3(g), dim3(256), 0, 0, (v4f*)dst, n16); });
    }
#define ROWS(S, ORDER, NT, WAVES) { const uint32_t nseg = (pitch + 256 * S - 1) / (256 * S), ngr = rows / (16 * WAVES); \
    snprintf(nm, sizeof nm, "rows: %d rows x %d steps per workgroup, %s fastest, %s", 16 * WAVES, S, ORDER ? "row group" : "segment", NT ? "nt" : "plain"); \
    timeit(nm, (double)WR, [&] { hipLaunchKernelGGL((rows_once<S, ORDER, NT, WAVES>), dim3(nseg * ngr), dim3(64 * WAVES), 0, 0, dst, rows, pitch, nseg); }); }
    ROWS(1, 0, true, 4) ROWS(4, 0, true, 4) ROWS(4, 0, false, 4) ROWS(8, 0, true, 4) ROWS(16, 0, true, 4) ROWS(57, 0, true, 4)
    ROWS(4, 1, true, 4) ROWS(16, 1, true, 4)
    ROWS(4, 0, true, 1) ROWS(16, 0, true, 1) ROWS(57, 0, true, 1) ROWS(57, 0, false, 1)
#define RC(S, ORDER, NT, PERSIST) { const uint32_t nseg = (58 + S - 1) / S, ngr = rows / 64; \
    snprintf(nm, sizeof nm, "aligned classes: 64 rows x %d pieces per task, %s fastest, %s, %s", S, ORDER ? "row group" : "segment", NT ? "nt" : "plain", PERSIST ? "persistent " #PERSIST : "one-shot"); \
    timeit(nm, (double)WR, [&] { hipLaunchKernelGGL((rows_class<S, ORDER, NT, PERSIST>), dim3(PERSIST ? PERSIST : nseg * ngr), dim3(256), 0, 0, dst, rows, pitch, nseg); }); }
    RC(1, 0, true, 0) RC(1, 0, false, 0) RC(2, 0, true, 0) RC(4, 0, true, 0) RC(4, 0, false, 0) RC(8, 0, true, 0) RC(15, 0, true, 0) RC(29, 0, true, 0) RC(58, 0, true, 0) RC(58, 0, false, 0)
    RC(1, 1, true, 0) RC(4, 1, true, 0) RC(29, 1, true, 0)
    RC(29, 0, true, 1024) RC(29, 0, true, 2048) RC(4, 0, true, 2048) RC(1, 0, true, 2048) RC(58, 0, true, 2048) RC(29, 1, true, 2048)
    return 0;
}

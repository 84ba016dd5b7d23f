// baz_music_hip.hip -- C-ABI (include/baz_music_hip.h) over the gfx950 MUSIC-DoA kernels.
//
// Host-side counterpart of baz_music_doa's state and work() body
// (/root/reference/lib/baz_music_doa.cc:35-53, 60-70, 72-161).  No torch / GNU Radio types
// cross this boundary; no CPU arithmetic fallback exists: without a usable gfx950 device
// baz_music_create() fails with BAZ_MUSIC_E_NODEVICE / BAZ_MUSIC_E_HIP.
#include "../../include/baz_music_hip.h"
#include "music_kernels.hip.h"
#include "music_wide_kernels.hip.h"
#include "scan_coarse_kernels.hip.h"
#include "scan_i8_kernels.hip.h"
#ifdef BAZ_MUSIC_LAB       // measured and NOT shipped (profiles/r05_i8p_negative.txt, r05_sort_negative.txt): the release library has none of their kernels
#include "scan_i8p_kernels.hip.h"
#include "sort_kernels.hip.h"
#endif
#include "table_kernels.hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

using namespace bazmusic;

// Environment switches.  The RELEASE library reads only the documented tuning knobs: BAZ_MUSIC_EXACT,
// BAZ_MUSIC_COARSE, BAZ_MUSIC_CHUNK_MIB, BAZ_MUSIC_PIN_LIMIT_MIB, BAZ_MUSIC_ZERO_COPY, BAZ_MUSIC_SINGLE_MIB -- none of them
// can make a result wrong.  Everything else (ablations, older kernels, geometry overrides, dumps) exists only in the lab
// build, -DBAZ_MUSIC_LAB = libbaz_music_hip_lab.so, which tests/lab and the A/B tests load explicitly
// (tests/test_abi.py::test_release_library_reads_only_the_documented_knobs lists the strings of the release .so).
#ifdef BAZ_MUSIC_LAB
#define BAZ_LAB_ENV(NAME) getenv(NAME)
#else
#define BAZ_LAB_ENV(NAME) (static_cast<const char*>(nullptr))
#endif

// ---- device allocations ---------------------------------------------------------------------------------------------------------
// Every device buffer of this library comes from dev_malloc / dev_free.  Release build: hipMalloc / hipFree.  Lab build with
// BAZ_MUSIC_GUARD=1 (round 6, after a GPU memory fault the driver saw and no test did): every buffer lies between two 64-KiB guard
// zones filled with a pattern; baz_music_debug_guard_check() -- and every dev_free -- compares the zones with the pattern, so a
// kernel that writes outside its buffer is caught at the buffer it ran over, not at whatever was allocated next to it.
namespace {
#ifdef BAZ_MUSIC_LAB
constexpr size_t GUARD_BYTES = 64u << 10;
constexpr int GUARD_PATTERN = 0xA5;
struct GuardRec { void* base; size_t bytes; };
std::mutex g_guard_mtx;
std::vector<std::pair<void*, GuardRec>> g_guards;     // user pointer -> allocation (a handful of buffers per context)
unsigned long long g_guard_damaged = 0;               // zones found overwritten so far (at a free or at a check)
int guard_level()
{
    static const int level = [] { const char* v = getenv("BAZ_MUSIC_GUARD"); return v ? atoi(v) : 0; }();
    return level;
}
bool guard_on() { return guard_level() != 0; }
// compares both zones of one allocation with the pattern (synchronises the device); returns the damaged zones (0 .. 2)
int guard_check_one(void* user, const GuardRec& g)
{
    static thread_local std::vector<unsigned char> host(GUARD_BYTES);
    int bad = 0;
    (void)hipDeviceSynchronize();
    for (int z = 0; z < 2; ++z) {
        const unsigned char* zone = static_cast<const unsigned char*>(g.base) + (z ? GUARD_BYTES + ((g.bytes + 255) & ~(size_t)255) : 0);
        if (hipMemcpy(host.data(), zone, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); ++bad; continue; }
        size_t first = GUARD_BYTES;
        for (size_t i = 0; i < GUARD_BYTES; ++i)
            if (host[i] != (unsigned char)GUARD_PATTERN) { first = i; break; }
        if (first != GUARD_BYTES) {
            ++bad;
            fprintf(stderr, "[baz_music_hip] GUARD ZONE DAMAGED: buffer %p (%zu bytes), zone %s it, first damaged byte at %+lld from the buffer's %s\n",
                    user, g.bytes, z ? "behind" : "in front of", z ? (long long)first : (long long)first - (long long)GUARD_BYTES, z ? "end (rounded up to 256)" : "start");
        }
    }
    return bad;
}
#endif

hipError_t dev_malloc(void** p, size_t bytes)
{
#ifdef BAZ_MUSIC_LAB
    if (guard_on()) {
        const size_t body = (bytes + 255) & ~(size_t)255;
        void* base = nullptr;
        hipError_t e = hipMalloc(&base, body + 2 * GUARD_BYTES);
        if (e != hipSuccess) return e;
        e = hipMemset(base, GUARD_PATTERN, body + 2 * GUARD_BYTES);     // (the body too: a read of an uninitialised buffer shows up as 0xA5A5...)
        // The fill runs on the null stream and the library's streams are non-blocking ones: a buffer allocated INSIDE a launch sequence (the peak
        // picker's private spectrum) was still being filled while the scan wrote it -- seen as a fuzz failure under the guard only.  Wait for it.
        // (BAZ_MUSIC_GUARD=2: without the wait, to show that.)
        if (e == hipSuccess && guard_level() != 2) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(base); return e; }
        *p = static_cast<unsigned char*>(base) + GUARD_BYTES;
        std::lock_guard<std::mutex> lk(g_guard_mtx);
        g_guards.emplace_back(*p, GuardRec{base, bytes});
        return hipSuccess;
    }
#endif
    return hipMalloc(p, bytes);
}

hipError_t dev_free(void* p)
{
#ifdef BAZ_MUSIC_LAB
    if (guard_on() && p) {
        std::lock_guard<std::mutex> lk(g_guard_mtx);
        for (size_t i = 0; i < g_guards.size(); ++i)
            if (g_guards[i].first == p) {
                const GuardRec g = g_guards[i].second;
                g_guard_damaged += (unsigned long long)guard_check_one(p, g);
                g_guards.erase(g_guards.begin() + (long)i);
                return hipFree(g.base);
            }
    }
#endif
    return hipFree(p);
}
}  // namespace

namespace {


struct StageProf {
    std::vector<hipEvent_t> ev;   // pairs (start, stop)
    size_t used = 0;
    double total_ms = 0.0;
    uint64_t launches = 0;
};

}  // namespace

// device-side slots of the host-fed path's pipelined copies: two when the host waits for a slot's outputs before it reuses the slot
// (pageable caller memory: the runtime blocks in those copies anyway), all four when the whole call is enqueued without the host
// waiting for anything (page-locked caller memory, round 6)
constexpr int BAZ_MUSIC_NSLOT = 4;

struct baz_music_ctx {
    uint32_t m = 0, n = 0, nsamples = 0, res = 0, K = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // the stream process_device() launches on (own or caller's)
    // steering table as the real bilinear-form table F[bin][m*m] (fp64) in MFMA B-operand order:
    // FB[step][chunk][lane] (double2), see build_FB; one padded step in front of step 0 and one behind the last
    // (the scan's row classes read shifted windows), dFB points at the allocation, step 0 is dFB + fb_step_elems
    double2* dFB = nullptr;
    uint32_t fb_steps = 0;   // 64-bin steps
    size_t fb_step_elems = 0;   // double2 elements per step of FB (2 * KS * 64)
    uint32_t nclass = 1;     // row classes of the spectrum port: 64 / gcd(res, 64) when res % 4 == 0, else 1
    uint32_t keep_mask = 0;  // low-word mask of the top-n key (bin index lives in the cleared bits)
    // per-range top-n candidate keys (scan_mfma_kernel -> topn_merge_kernel)
    double* dCand = nullptr;
    size_t cand_cap = 0;     // entries
    // workspace
    double2* dR = nullptr;
    double* dQ = nullptr;
    uint32_t cap = 0;   // items
    // host-fed path (baz_music_process): two device-side slots so that the H2D copy of chunk i+1, the kernels
    // of chunk i and the D2H copy of chunk i-1 overlap on three streams
    struct Slot {
        float* in = nullptr;
        float* al = nullptr;         // ang then lvl of the chunk in flight, back to back (2 * cap * n floats)
        float* h_al = nullptr;       // page-locked host image of `al`: ONE small D2H per chunk, then a CPU copy to the caller
        float* spec = nullptr;
        hipEvent_t h2d = nullptr, comp = nullptr, d2h = nullptr;
        bool busy = false;
        uint32_t pend_done = 0, pend_nb = 0;   // the chunk whose ang / lvl still sit in h_al
    } slot[BAZ_MUSIC_NSLOT];
    uint32_t s_cap = 0;
    bool s_has_spec = false;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   // baz_music_process_device_on: ordering against the caller's stream
    std::mutex mtx;   // serialises process* against the table SWAP of set_table, like d_mutex (.cc:67,101)
    // Retune without stalling the stream (round 5): baz_music_set_table builds every image of the new table with the kernels
    // of table_kernels.hip.h on `s_tab` into the SHADOW set and takes `mtx` only to exchange the two sets (TableSet below)
    struct TableSet {
        double2* dFB = nullptr; double2* dTB = nullptr; uint4* dCS = nullptr; uint4* dIB = nullptr; double* dA2p = nullptr;
        float2* dTA = nullptr; double* dA2 = nullptr;
#ifdef BAZ_MUSIC_LAB
        uint4* dIP = nullptr;            // level-packed int8 operands (m <= 4; shares i8 / i8_ok with dIB: a configuration has one of the two)
        float* dKT = nullptr;            // float32 table at the sort key's sample bins (sort_kernels.hip.h; m <= 8)
#endif
        CoarseParams cs = {0.0f, 0.0f, 0.0, 0.0, 1};
        I8Params i8 = {};
        bool cs_ok = false, i8_ok = false;
        double refine_below = 0.0;
    } shadow;
    std::mutex tab_mtx;              // retunes among themselves (held for a whole baz_music_set_table; never while `mtx` is wanted by work())
    hipStream_t s_tab = nullptr;     // side stream of the table builders (highest priority: they are tiny)
    hipEvent_t ev_swap = nullptr;    // recorded on the launch stream at every swap: batches that still read the retired set
    bool swap_recorded = false;
    float* dRaw = nullptr;           // the raw fp32 table on the device (resolution x m complex64)
    float* hRaw = nullptr;           // its page-locked staging copy (the caller's vector may be pageable and short-lived)
    void* dTabStats = nullptr;       // baztab::TableStats
    void* hTabStats = nullptr;       // ... page-locked
    float* h_al_big = nullptr;                          // host-fed zero-copy calls: page-locked ang / lvl image of a whole call
    size_t h_al_big_cap = 0;                            // floats
    double last_retune_ms = 0.0;     // wall time of the last baz_music_set_table, of which ...
    double last_swap_wait_ms = 0.0;  // ... waiting for and holding `mtx`
    int profiling = 0;      // 0 off, 1 every stage, 2 only the dominant (scan) stage
    int lab_variant = 0;
    // literal-form refinement of near-null tiles (literal_tile() inside the scan)
    double* dG = nullptr;          // noise eigenvectors, item-minor like dQ (cap * m*m * 2 doubles)
    double2* dTB = nullptr;        // raw steering table as fp64 MFMA B-operand image (build_TB), padded like dFB
    size_t tb_step_elems = 0;      // double2 elements per step of TB (2 * ceil(2m/4) * 64)
    unsigned long long* dRefined = nullptr;   // statistic: (item, bin) values recomputed, [2]: double-buffered by API call
    int stat_parity = 0;                      // the scan adds to dRefined[stat_parity]; the merge clears the other one
    bool stat_next_clean = true;              // false after a call that failed before its merge ran
    double refine_below = 0.0;     // threshold on d = a^H Q a
    int refine_off = 0;            // lab (BAZ_MUSIC_NO_REFINE=1): projector form everywhere
    int lab_cov_old = 0;           // lab (BAZ_MUSIC_COV_OLD=1): the round-1 covariance kernel at m = 4
    int force_nsplit = 0;          // tests / lab (BAZ_MUSIC_NSPLIT=k): bin ranges per row in the scan, 0 = by batch size
    // wide arrays (17 <= m <= BAZ_MUSIC_MAX_M): the run-time-m kernels of music_wide_kernels.hip.h
    bool wide = false;
    float2* dTA = nullptr;         // steering table transposed, [m][res] complex64
    double2* dGw = nullptr;        // noise eigenvectors, [items][m - n][m]
    double* dWS = nullptr;         // fp64 strengths, [items][res] (the top-n's input)
    double2* dSw = nullptr;        // signal eigenvectors, [items][n][m] (the scan's short form where 2n <= m)
    double* dA2 = nullptr;         // ||a||^2 per bin
    int wide_literal_only = 0;     // lab (BAZ_MUSIC_WIDE_LITERAL=1): no short form in scan_wide_kernel
    uint32_t wide_cov_blocks = 512;
    int wide_cov_mfma = 0;         // 17 <= m <= 32: cov_wide_mfma_kernel, 33 <= m <= 64: cov_wide_pairs_kernel (BAZ_MUSIC_WIDE_COV_MFMA=0: lab)
    int wide_mfma = 0;             // 17 <= m <= 64, n <= 8: the scan on the fp64 matrix core (scan_wide_mfma_kernel; BAZ_MUSIC_WIDE_MFMA=0: lab)
    uint32_t wide_cap = 0;         // items the three buffers above (and dR) hold
    double* dSs = nullptr;         // short_form_applies(): coefficient vectors of the scan's short form, [2n * 2m][q_stride]
    double* dA2p = nullptr;        // ... and ||a||^2 per bin, padded like dFB (fb_steps + 2 steps of 64)
    int sig_scan = 1;              // lab / tests: BAZ_MUSIC_SIG_SCAN=0 keeps the projector GEMM
    uint8_t* dRedo = nullptr;      // [cap] items evd_sub_kernel hands back to the Jacobi
    int sub_evd = 1;               // signal subspace by orthogonal iteration where n <= 4 (run-time-m kernels: n <= 8) (lab: BAZ_MUSIC_SUB_EVD=0)
    int fused_covevd = 0;          // m = 4, K % 256 == 0: covariance + EVD in one kernel (BAZ_MUSIC_FUSE=0: lab, two kernels)
    uint32_t covevd_blocks = 512u; // grid of cov4_evd_kernel: the workgroups resident at once (2 per CU)
    int covevd_task_items = 0;     // lab (BAZ_MUSIC_COVEVD_TASK_ITEMS = 64 / 32 / 16): items per wave task of cov4_evd_kernel, 0 = by batch size
    uint32_t cov4_resident_blocks = 256u;        // grid of cov4_x4_kernel (persistent waves): one workgroup per CU
    // coarse-gated scan (scan_coarse_kernels.hip.h): m <= 8, spectrum port not wired
    uint4* dCS = nullptr;          // per 16-bin tile: f16 hi/lo pieces of the scaled table (B32, B16) + the fp64 B operand (X)
    uint32_t cs_tiles = 0;         // tiles in the image (a multiple of 8)
    CoarseParams cs = {0.0f, 0.0f, 0.0, 0.0, 1};
    bool cs_ok = false;            // image built and its scales representable
    uint32_t last_nsplit = 1;      // bin ranges per item the last scan launch produced candidates for (the merge folds them)
    int coarse = 1;                // BAZ_MUSIC_COARSE=0: the full fp64 scan also without the spectrum port (A/B, tests)
    int coarse_rg = 4;             // BAZ_MUSIC_COARSE_RG: row groups (x 16 items) per wave, 2 or 4 (lab)
    int coarse_lab = 0;            // BAZ_MUSIC_COARSE_LAB=1: never run an exact tile (cost of the coarse passes alone; wrong results)
    int coarse_stats = 0;          // BAZ_MUSIC_COARSE_STATS=1: count exact tile evaluations (baz_music_debug_coarse_fired)
    unsigned long long* dMargin = nullptr;   // baz_music_debug_coarse_margin: worst error / allowance (float bits << 32 | where)
    // int8-matrix-core scan (scan_i8_kernels.hip.h): 6 <= m <= 16, n <= 4
    uint4* dIB = nullptr;          // digit image of the table (build_i8_image)
#ifdef BAZ_MUSIC_LAB
    uint4* dIP = nullptr;          // level-packed digit operands, 2 .. 4 antennas (build_i8p_kernel); parameters in `i8` as well
    float* dKT = nullptr;          // sort-key table (sort_kernels.hip.h)
    // Sorting the items of a batch by their nulls in front of the gated scan (sort_kernels.hip.h), while that scan reports many fired tiles
    uint16_t* dKeys = nullptr;
    uint32_t* dHist = nullptr;     // items per key (KEY_BUCKETS)
    uint32_t* dCursor = nullptr;   // items per range of keys (KEY_BLOCKS)
    uint32_t* dPerm = nullptr;
    uint32_t sort_cap = 0;         // items dKeys / dPerm hold
    unsigned long long* dFire = nullptr;        // [2] exact evaluations / (row group, tile) pairs walked by the gated scan of the call in flight
    unsigned long long* hFire = nullptr;        // page-locked [4]: the same of a FINISHED call, [2] = its tag (call number << 1 | sorted), written by its merge
    unsigned long long* hFireDev = nullptr;     // ... its device address
    unsigned long long fire_calls = 0, fire_seen = 0;
    int sort_mode = 0;             // 0 never (the product), 1 always, -1 adaptive -- LAB builds only (BAZ_MUSIC_SORT).  Measured and not shipped:
                                   // the order cuts the exact evaluations of an incoherent batch from 22 % to 4.6 % of the tile pairs, but key + sort cost
                                   // 0.17 ms (0.08 with a library radix sort) of the 0.22 ms the scan gains (profiles/r05_sort_negative.txt)
    bool sort_on = false;
    uint32_t sort_streak = 0;
    uint64_t sort_clock = 0, sort_retry_at = 0;
    double rate_unsorted = 0.0;
    uint64_t sorted_calls = 0, unsorted_calls = 0;     // (tap: baz_music_debug_sort_state)
    bool last_gated_sorted = false;
#endif
    uint32_t scan_lds_pad = 0;     // lab (BAZ_MUSIC_SCAN_LDS_PAD bytes): dynamic LDS the fp64 scan asks for and never touches -- caps its workgroups per CU
                                   // (tests/lab/two_ctx_split.py: does a low-register covariance of ANOTHER context fit beside three of them?)
    int seq_walk = 0;              // lab (BAZ_MUSIC_SEQ_WALK=1): scan_mfma_kernel walks its steps left to right (round 4's order; A/B of the strided walk)
#ifdef BAZ_MUSIC_LAB
    int i8p_on = 0;                // LAB builds only (BAZ_MUSIC_I8P=1): the level-packed int8 scan at m <= 4.  Measured and not shipped
                                   // (profiles/r05_i8p_negative.txt): its arithmetic is 0.31 ms against the fp64 scan's 0.56, but the spectrum
                                   // stores alone take what the fp64 scan takes (0.59 - 0.73 ms by box), and incoherent batches run 2.4 x slower
#endif
    I8Params i8 = {};
    bool i8_ok = false;            // image built (finite table, scale representable, size within I8_IMAGE_LIMIT)
    int i8_on = 1;                 // BAZ_MUSIC_EXACT=1: every value on the fp64 matrix core (A/B; the round-3 scan)
    int i8_abl = 0;                // lab (BAZ_MUSIC_I8_ABL): ablation mask of scan_i8_kernel (timing only)
    int refine_nocount = 0;        // lab (BAZ_MUSIC_NO_REFINE_COUNT): the scans do not count the values they recompute
    uint32_t num_cus = 256;        // compute units of the device (launch geometry of scan_i8_kernel)
    int i8_wgs_per_cu[2][2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};   // [NMAX > 2][SPEC][VEC4]: resident workgroups per CU (occupancy API, on first use)
    unsigned long long* dI8Stat = nullptr;   // [0] wave tiles that ran the refined form, [1] wave tiles walked, [2] .. [4] VAL margins
    int peak_mode = 0;      // 0: the reference's n strongest bins; 1 (opt-in extension): n strongest local maxima
    float* dPeakSpec = nullptr;   // internal spectrum when peak mode runs without the spectrum port
    size_t peak_spec_cap = 0;     // floats
    size_t chunk_bytes = 0;   // host-fed path: traffic per pipelined chunk (BAZ_MUSIC_CHUNK_MIB); 0 = by buffer kind
    // host-fed path: page ranges of the caller's buffers this context has page-locked (baz_music_host_register)
    struct HostPin { uintptr_t lo, hi; uint32_t asked; };   // [lo, hi): the caller's exact bytes
    std::vector<HostPin> pins;                           // the registrations we hold a share of (g_pins): disjoint, not touching
    std::vector<HostPin> refused;                        // requests the runtime refused (`asked` again so many times since)
    uint64_t pinned_bytes = 0;
    uint64_t pin_limit = 4096ull << 20;                  // BAZ_MUSIC_PIN_LIMIT_MIB
    int auto_pin = 0;                                    // baz_music_set_host_pinning
    int zero_copy = 1;                                   // small calls on page-locked memory: no copies (BAZ_MUSIC_ZERO_COPY=0: lab)
    int single_limit_mib = 64;                           // page-locked calls below this much traffic run as ONE chunk (BAZ_MUSIC_SINGLE_MIB)
    StageProf prof[BAZ_MUSIC_NUM_STAGES];
    std::string stage_name[BAZ_MUSIC_NUM_STAGES];
    char scan_names[4][64] = {{0}, {0}, {0}, {0}};   // baz_music_stage_name(SCAN) by scan_kind, written once by baz_music_create
    int scan_kind = -1;     // the scan kernel the LAST launch took: 0 scan_mfma_kernel, 1 scan_i8_kernel, 2 scan_coarse_kernel, 3 scan_i8p_kernel (-1: none yet)
    char hip_err[256] = {0};
};

namespace {

int hip_fail(baz_music_ctx* c, hipError_t e, const char* what)
{
    if (c) snprintf(c->hip_err, sizeof(c->hip_err), "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();   // the runtime's per-thread "last error" is sticky: do not let the next call find this one
    return BAZ_MUSIC_E_HIP;
}

#define HIP_TRY(ctx, call)                                         \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return hip_fail((ctx), e__, #call); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

// F[bin][i*m+j]: i==j |a_i|^2 ; i<j Re(conj(a_i) a_j) ; i>j Im(conj(a_j) a_i)  (pair (j,i), j<i).
// The fp32 table entries are widened exactly (.cc:110-112); the products are fp64.
void build_F(const float* table_ri, uint32_t m, uint32_t res, std::vector<double>& F)
{
    const size_t mm = (size_t)m * m;
    F.assign((size_t)res * mm, 0.0);
    for (uint32_t s = 0; s < res; ++s) {
        const float* a = table_ri + (size_t)s * m * 2;
        double* f = F.data() + (size_t)s * mm;
        for (uint32_t i = 0; i < m; ++i) {
            const double air = a[2 * i], aii = a[2 * i + 1];
            f[i * m + i] = air * air + aii * aii;
            for (uint32_t j = i + 1; j < m; ++j) {
                const double ajr = a[2 * j], aji = a[2 * j + 1];
                // conj(a_i) a_j = (air - i aii)(ajr + i aji)
                f[i * m + j] = air * ajr + aii * aji;
                f[j * m + i] = air * aji - aii * ajr;
            }
        }
    }
}

// MFMA B-operand image of the table (scan_mfma_kernel): FB[step][c2][lane] (double2), c2 = 2*s + (t>>1),
// component t&1, lane = (g = lane>>4, c = lane&15):
//     value = F[bin = 64*step + 4*c + t][e = 4*s + g]          (0 for the K padding e >= m*m)
// i.e. tile t of a 64-bin step carries the bins 4c + t in its columns, so that a lane's accumulator
// registers of the 4 tiles are 4 consecutive bins.  step runs from -1 to `steps` inclusive (array index step + 1):
// the scan's row classes read windows shifted by up to 60 bins.  Bins outside [0, res) get a huge
// diagonal: d = BIG * trace(Q) = BIG * (m - n), finite, never stored, never among the top n.
void build_FB(const std::vector<double>& F, uint32_t m, uint32_t res, uint32_t steps, std::vector<double>& FB)
{
    const uint32_t mm = m * m;
    const uint32_t ks = (mm + 3) / 4;
    const double BIG = 1e300;
    FB.assign((size_t)(steps + 2) * 2 * ks * 64 * 2, 0.0);
    for (uint32_t sti = 0; sti < steps + 2; ++sti)
        for (uint32_t s = 0; s < ks; ++s)
            for (uint32_t t = 0; t < 4; ++t)
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const uint32_t g = lane >> 4, c = lane & 15;
                    const int64_t bin = 64 * ((int64_t)sti - 1) + 4 * c + t;
                    const uint32_t e = 4 * s + g;
                    double v = 0.0;
                    if (e < mm) {
                        if (bin >= 0 && bin < (int64_t)res) v = F[(size_t)bin * mm + e];
                        else v = ((e / m) == (e % m)) ? BIG : 0.0;
                    }
                    FB[(((size_t)sti * 2 * ks + 2 * s + (t >> 1)) * 64 + lane) * 2 + (t & 1)] = v;
                }
}

// The raw table in the same B-operand order, for the literal form ||G^H a||^2 of near-null tiles (literal_tile):
// K dimension = the 2m real coordinates (re a_0, im a_0, re a_1, ...), fp32 entries widened exactly (.cc:110-112):
//     TB[step][2*s + (t>>1)][lane].{x,y}[t&1] = (e & 1 ? imag : real)(table[bin = 64*step + 4*c + t][e >> 1]),  e = 4*s + g
// zero outside [0, res) (those bins keep their projector value) and for the K padding e >= 2m.
void build_TB(const float* table_ri, uint32_t m, uint32_t res, uint32_t steps, std::vector<double>& TB)
{
    const uint32_t ks2 = (2 * m + 3) / 4;
    TB.assign((size_t)(steps + 2) * 2 * ks2 * 64 * 2, 0.0);
    for (uint32_t sti = 0; sti < steps + 2; ++sti)
        for (uint32_t s = 0; s < ks2; ++s)
            for (uint32_t t = 0; t < 4; ++t)
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const uint32_t g = lane >> 4, c = lane & 15;
                    const int64_t bin = 64 * ((int64_t)sti - 1) + 4 * c + t;
                    const uint32_t e = 4 * s + g;
                    double v = 0.0;
                    if (e < 2 * m && bin >= 0 && bin < (int64_t)res) v = (double)table_ri[((size_t)bin * m + (e >> 1)) * 2 + (e & 1)];
                    TB[(((size_t)sti * 2 * ks2 + 2 * s + (t >> 1)) * 64 + lane) * 2 + (t & 1)] = v;
                }
}

uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

using baztab::f16_bits;
using baztab::f16_value;

// Table images of the coarse-gated scan (scan_coarse_kernel): the C array (1,024 B per 16-bin tile) followed by the X
// array (2,048 B per tile):
//   C    Fh then Fl, each 32 entries x 8 f16: entry (gb, c), j < 8 -> piece[bin = 16 tile + c][e = 8 gb + j]
//        (lane (g, c) of the K = 32 B operand [F | F] reads entry (g & 1, c))
//   X    k-step pair p, lane (g, c), component s & 1: F[bin][e = 4 s + g], s = 2 p + (s & 1)   (fp64 B operand)
// Fs = F * FS, Fh = f16(Fs), Fl = f16(Fs - Fh); bins outside the table: a huge diagonal (never selected), like build_FB.
// Returns false when the table's scale cannot be represented (the full scan is used then).
// The coarse form's scales and thresholds from max|F| (shared by the device builders and the host checker); false when the
// table's scale cannot be represented.
bool coarse_params_from(double fmax, uint32_t m, CoarseParams& cp, double& FS)
{
    if (!(fmax > 0.0) || !std::isfinite(fmax)) return false;
    int ex = 0;
    (void)std::frexp(fmax, &ex);                    // fmax = f * 2^ex, f in [0.5, 1)
    const int fs_exp = 14 - ex;                     // fmax * 2^fs_exp in [2^13, 2^14)
    if (fs_exp < -90 || fs_exp > 90) return false;  // SC must stay a comfortable float
    FS = std::ldexp(1.0, fs_exp);
    const double SC = 1024.0 * FS;
    cp.sc = SC;
    cp.fmax = fmax;
    const double nga = (double)cs_ng((int)m);             // the allowance E = nga 2^-16 (S + D)
    cp.sc_up = std::nextafter((float)(SC * (1.0 + nga * 0x1p-16) * (1.0 + 0x1p-20)), INFINITY);
    cp.es_factor = std::nextafter((float)(nga * 0x1p-16 * fmax * SC), INFINITY);
    return true;
}

bool build_coarse_image(const std::vector<double>& F, uint32_t m, uint32_t res, uint32_t tiles, std::vector<uint8_t>& img,
                        CoarseParams& cp)
{
    const uint32_t mm = m * m;
    double fmax = 0.0;
    for (double v : F) {
        if (!std::isfinite(v)) return false;
        fmax = std::max(fmax, std::fabs(v));
    }
    double FS = 0.0;
    if (!coarse_params_from(fmax, m, cp, FS)) return false;
    const uint32_t ng = (uint32_t)cs_groups((int)m);      // groups of 16 terms (fp64 operand)
    const bool wide = m > 4;                              // coarse operands in groups of 32 terms, Fh and Fl 1 KiB each
    const size_t cbytes = (size_t)cs_c_units((int)m) * 16, xbytes = (size_t)ng * CS_X_UNITS * 16;
    // [C of every tile and of one more (the kernel stages the first tile of the NEXT phase with every phase)][X of every tile]
    img.assign((size_t)(tiles + 1) * cbytes + (size_t)tiles * xbytes, 0);
    for (uint32_t t = 0; t <= tiles; ++t) {
        uint8_t* T = img.data() + (size_t)t * cbytes;
        std::vector<double> xpad(256 * ng);
        double* X = (t < tiles) ? reinterpret_cast<double*>(img.data() + (size_t)(tiles + 1) * cbytes + (size_t)t * xbytes) : xpad.data();
        for (uint32_t c = 0; c < 16; ++c) {
            const uint32_t bin = 16 * t + c;
            for (uint32_t e = 0; e < 16 * ng; ++e) {
                if (wide && e >= 32u * (uint32_t)cs_groups32((int)m)) continue;
                uint16_t* fh = reinterpret_cast<uint16_t*>(T + (wide ? (size_t)(e >> 5) * 2048 : 0));          // group: Fh then Fl
                uint16_t* fl = reinterpret_cast<uint16_t*>(T + (wide ? (size_t)(e >> 5) * 2048 + 1024 : 512));
                uint16_t hi = 0, lo = 0;
                if (e < mm) {
                    if (bin >= res) hi = ((e / m) == (e % m)) ? f16_bits(32768.0f) : 0;
                    else {
                        const float fs = (float)(F[(size_t)bin * mm + e] * FS);
                        hi = f16_bits(fs);
                        lo = f16_bits(fs - f16_value(hi));
                    }
                }
                const uint32_t gb = wide ? ((e >> 3) & 3u) : (e >> 3), j = e & 7u;       // entry (gb, c), j: k = 8 gb + j
                fh[(gb * 16 + c) * 8 + j] = hi;
                fl[(gb * 16 + c) * 8 + j] = lo;
            }
            for (uint32_t g = 0; g < 4; ++g)
                for (uint32_t sidx = 0; sidx < 4 * ng; ++sidx) {
                    const uint32_t e = 4 * sidx + g, lane = g * 16 + c;
                    double v = 0.0;
                    if (e < mm) v = (bin < res) ? F[(size_t)bin * mm + e] : (((e / m) == (e % m)) ? 1e300 : 0.0);
                    X[(sidx >> 1) * 128 + lane * 2 + (sidx & 1)] = v;
                }
        }
    }
    return true;
}


// Digit images of the table for the int8-matrix-core scan (scan_i8_kernels.hip.h): Fi = rint(F 2^54 / Fscale) cut into ND = 7
// balanced base-256 digits, most significant first (digits 1.. in [-128, 127], the first what is left: |.| <= 65), laid out as
// B operands of v_mfma_i32_16x16x64_i8 -- the five leading digits in `img` (staged through LDS by every tile), digits 5 and 6
// behind them (read from L2 by the tiles that refine):
//     img [(((st*4 + t)*NKB + kb)*5 + s)*1024 + lane*16 + j]          = digit s     of Fi[bin = 64 st + 4 c + t][e = 64 kb + 16 g + j]
//     img2[(((st*4 + t)*NKB + kb)*2 + s)*1024 + lane*16 + j]          = digit 5 + s   (img2 = img + i8_image_bytes5)
// (lane = 16 g + c; 0 for e >= m^2 and for bins outside the table, which the kernel gives a huge d).  Fills the kernel's
// parameters.  false: table not finite / all zero / scale out of range (the fp64 scan runs then).
constexpr size_t I8_IMAGE_LIMIT = (size_t)384 << 20;
size_t i8_image_bytes5(uint32_t m, uint32_t steps) { return (size_t)steps * 4 * i8_nkb((int)m) * I8_NS * 1024; }
size_t i8_image_bytes(uint32_t m, uint32_t steps) { return (size_t)steps * 4 * i8_nkb((int)m) * I8_ND * 1024; }

// The int8 forms' fixed-point scale, level weights and a-priori bounds from max|F| (shared by the device builders and the host
// checker); sf = the factor that turns F into its 2^54-scaled integer.  false: scale out of range.
bool i8_params_from(double fmax, uint32_t m, I8Params& ip, double& sf)
{
    const uint32_t mm = m * m;
    constexpr int NS = I8_NS, ND = I8_ND;
    if (!(fmax > 0.0) || !std::isfinite(fmax)) return false;
    int ex = 0;
    (void)std::frexp(fmax / I8_QMAX, &ex);          // fmax / QMAX = f 2^ex, f in [0.5, 1)  ->  fmax / 2^ex < QMAX
    // (float) wt[3] = Fscale 2^-36 must scale the bulk form's integer V in [1, 2^48) without leaving the normal float range
    if (ex < -88 || ex > 100) return false;
    const double fscale = std::ldexp(1.0, ex);
    const double sq = std::ldexp(1.0, 8 * ND - 2);
    sf = sq / fscale;
    for (int l = 0; l < ND; ++l) ip.wt[l] = std::ldexp(fscale, -12 - 8 * l);
    ip.sq = sq;
    // digits cut off + levels dropped + the low byte of the level-4 sum (scan_i8_kernels.hip.h, "Error bound")
    ip.e_bound = (double)mm * fscale * (double)NS * 1.01 * std::ldexp(1.0, 2 - 8 * NS) + fscale * std::ldexp(1.0, -12 - 8 * (NS - 2));
    ip.t_acc = ip.e_bound * (1.0 + 1.0 / I8_EPS);
    // first tier: four leading digits, levels 0 .. 3
    ip.e4_bound = (double)mm * fscale * (double)(NS - 1) * 1.01 * std::ldexp(1.0, 2 - 8 * (NS - 1));
    const double t4 = ip.e4_bound * (1.0 + 1.0 / I8_EPS);
    ip.t4_f = (float)t4;
    if ((double)ip.t4_f < t4) ip.t4_f = std::nextafterf(ip.t4_f, INFINITY);
    ip.ws_f = (float)ip.wt[NS - 2];
    ip.t_acc_f = (float)ip.t_acc;
    if ((double)ip.t_acc_f < ip.t_acc) ip.t_acc_f = std::nextafterf(ip.t_acc_f, INFINITY);
    // (VAL) the refined form against the fp64 form it is compared with: its own digits (7.07 2^-54 MM Fscale) plus the fp64
    // form's accumulation error, <= MM 2^-53 sum_e |q_e F_e| <= MM^2 2^-53 Fscale in the worst case
    ip.e_refined = (double)mm * fscale * std::ldexp(1.0, -54) * (7.07 + 2.0 * (double)mm);
    return true;
}

bool build_i8_image(const std::vector<double>& F, uint32_t m, uint32_t res, uint32_t steps, std::vector<uint8_t>& img,
                    I8Params& ip)
{
    const uint32_t mm = m * m, nkb = (uint32_t)i8_nkb((int)m);
    constexpr int NS = I8_NS, ND = I8_ND;
    double fmax = 0.0;
    for (double v : F) {
        if (!std::isfinite(v)) return false;
        fmax = std::max(fmax, std::fabs(v));
    }
    double sf = 0.0;
    if (!i8_params_from(fmax, m, ip, sf)) return false;
    img.assign(i8_image_bytes(m, steps), 0);
    uint8_t* img2 = img.data() + i8_image_bytes5(m, steps);
    for (uint32_t bin = 0; bin < res; ++bin) {
        const uint32_t st = bin >> 6, w = bin & 63u, c = w >> 2, t = w & 3u;
        for (uint32_t e = 0; e < mm; ++e) {
            const uint32_t kb = e >> 6, g = (e >> 4) & 3u, j = e & 15u;
            long long v = std::llrint(F[(size_t)bin * mm + e] * sf);          // |v| <= 2^54 (1 + 2^-10); the product is exact
            const size_t tile = (size_t)(st * 4 + t) * nkb + kb, in_lane = (size_t)(g * 16 + c) * 16 + j;
            for (int s = ND - 1; s >= 1; --s) {
                const long long h = (v + 128) >> 8;          // floor((v + 128) / 256)  (arithmetic shift)
                const uint8_t dg = (uint8_t)((v - h * 256) & 255);
                if (s >= NS) img2[(tile * (ND - NS) + (size_t)(s - NS)) * 1024 + in_lane] = dg;
                else img[(tile * NS + (size_t)s) * 1024 + in_lane] = dg;
                v = h;
            }
            img[(tile * NS) * 1024 + in_lane] = (uint8_t)(v & 255);
        }
    }
    return true;
}

#ifdef BAZ_MUSIC_LAB
// Level-packed digit operands for 2 .. 4 antennas (scan_i8p_kernels.hip.h), host checker of baztab::build_i8p_kernel:
//     B [((st + 1) * 4 + t) * 64 + 16 s + c][j]  = digit s (0 .. 3) of Fi[bin = 64 st + 4 c + t][e = j]       (16 bytes per entry)
//     B'[ ...                     16 s + c][j]  = digit 4 + s (s = 0 .. 2), slot 3 zero;  B' follows B (i8p_operand_units each)
bool build_i8p_image(const std::vector<double>& F, uint32_t m, uint32_t res, uint32_t steps, std::vector<uint8_t>& img, I8Params& ip)
{
    const uint32_t mm = m * m;
    constexpr int ND = I8_ND;
    double fmax = 0.0;
    for (double v : F) {
        if (!std::isfinite(v)) return false;
        fmax = std::max(fmax, std::fabs(v));
    }
    double sf = 0.0;
    if (!i8_params_from(fmax, m, ip, sf)) return false;
    img.assign(i8p_image_bytes(steps), 0);
    uint8_t* img2 = img.data() + i8p_operand_units(steps) * 16;
    for (uint32_t bin = 0; bin < res; ++bin) {
        const uint32_t st = bin >> 6, w = bin & 63u, c = w >> 2, t = w & 3u;
        for (uint32_t e = 0; e < mm; ++e) {
            long long v = std::llrint(F[(size_t)bin * mm + e] * sf);
            const size_t unit = ((size_t)(st + 1) * 4 + t) * 64 + c;
            int dgs[ND];
            for (int s = ND - 1; s >= 1; --s) {
                const long long h = (v + 128) >> 8;
                dgs[s] = (int)((v - h * 256) & 255);
                v = h;
            }
            dgs[0] = (int)(v & 255);
            for (int s = 0; s < ND; ++s) {
                uint8_t* base = (s < 4) ? img.data() + (unit + (size_t)s * 16) * 16 : img2 + (unit + (size_t)(s - 4) * 16) * 16;
                base[e] = (uint8_t)dgs[s];
            }
        }
    }
    return true;
}
#endif

// the scan's short form (scan_mfma_kernel, SIG) needs fewer MFMAs than the projector GEMM
bool short_form_applies(uint32_t m, uint32_t n) { return (n == 2 && m >= 9 && m <= 16) || (n == 1 && m >= 6 && m <= 16); }
// The int8-matrix-core scan applies: 6 .. 16 antennas (row classes below, run-time-m kernels above), lists of <= 4 keys.
bool i8_active(const baz_music_ctx* c)
{
    return c->i8_on && c->i8_ok && c->dIB && c->m >= 6 && c->m <= 16 && c->n <= 4 && !c->lab_variant;
}
// ... and its level-packed form for 2 .. 4 antennas (scan_i8p_kernels.hip.h), with the spectrum port: lab builds only
#ifdef BAZ_MUSIC_LAB
bool i8p_active(const baz_music_ctx* c)
{
    return c->i8_on && c->i8p_on && c->i8_ok && c->dIP && c->m <= 4 && c->n <= 4 && !c->lab_variant;
}
#else
constexpr bool i8p_active(const baz_music_ctx*) { return false; }
#endif
bool short_form_in_use(const baz_music_ctx* c)
{
    // (the integer form evaluates the projector form: the EVD must write its coefficients)
    return short_form_applies(c->m, c->n) && c->sig_scan && c->dSs && c->dA2p && !c->lab_variant && !i8_active(c);
}

int ensure_workspace(baz_music_ctx* c, uint32_t batch)
{
    if (batch <= c->cap) return BAZ_MUSIC_OK;
    const uint32_t cap = round_up(batch, 64);
    const size_t mm = (size_t)c->m * c->m;
    // batches launched earlier may still use the buffers freed below: drain them first (hipFree happens to synchronise the device
    // today; nothing here relies on it)
    if (c->cap) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->dR) { (void)dev_free(c->dR); c->dR = nullptr; }
    if (c->dQ) { (void)dev_free(c->dQ); c->dQ = nullptr; }
    if (c->dG) { (void)dev_free(c->dG); c->dG = nullptr; }
    if (c->dRedo) { (void)dev_free(c->dRedo); c->dRedo = nullptr; }
    if (c->dSs) { (void)dev_free(c->dSs); c->dSs = nullptr; }
    c->cap = 0;
    HIP_TRY(c, dev_malloc((void**)&c->dR, (size_t)cap * mm * sizeof(double2)));
    HIP_TRY(c, dev_malloc((void**)&c->dQ, (size_t)cap * mm * sizeof(double)));
    HIP_TRY(c, dev_malloc((void**)&c->dG, (size_t)cap * mm * 2 * sizeof(double)));
    if (c->dRedo) { (void)dev_free(c->dRedo); c->dRedo = nullptr; }
    HIP_TRY(c, dev_malloc((void**)&c->dRedo, (size_t)cap));
    if (c->dSs) { (void)dev_free(c->dSs); c->dSs = nullptr; }
    if (short_form_applies(c->m, c->n)) HIP_TRY(c, dev_malloc((void**)&c->dSs, (size_t)cap * 4 * c->n * c->m * sizeof(double)));
    c->cap = cap;
    return BAZ_MUSIC_OK;
}

struct ProfScope {
    baz_music_ctx* c;
    int stage;
    hipEvent_t stop = nullptr;
    ProfScope(baz_music_ctx* ctx, int st) : c(ctx), stage(st)
    {
        if (!c->profiling || (c->profiling == 2 && stage != BAZ_MUSIC_STAGE_SCAN)) return;
        StageProf& p = c->prof[stage];
        if (p.used + 2 > p.ev.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                p.ev.push_back(e);
            }
        }
        (void)hipEventRecord(p.ev[p.used], c->stream);
        stop = p.ev[p.used + 1];
        p.used += 2;
    }
    ~ProfScope()
    {
        if (stop) (void)hipEventRecord(stop, c->stream);
    }
};

void prof_collect(baz_music_ctx* c)
{
    for (int s = 0; s < BAZ_MUSIC_NUM_STAGES; ++s) {
        StageProf& p = c->prof[s];
        for (size_t k = 0; k + 1 < p.used; k += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.ev[k], p.ev[k + 1]) == hipSuccess) {
                p.total_ms += ms;
                p.launches += 1;
            }
        }
        p.used = 0;
    }
}

// ---- kernel dispatch -------------------------------------------------------------------

template <int M>
int launch_cov_t(baz_music_ctx* c, const float* d_in, uint32_t batch, double2* dR)
{
    if constexpr (M == 4) {
        if ((c->K % 256u) == 0 && !c->lab_cov_old) {   // dwordx4 stream + LDS transpose + 4x4x4 MFMA blocks
            // persistent waves, one workgroup per CU (see baz_music_create)
            const uint32_t blocks = std::min<uint32_t>((batch + 3) / 4, c->cov4_resident_blocks);
            hipLaunchKernelGGL(cov4_x4_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, dR, batch, c->K);
            HIP_TRY(c, hipGetLastError());
            return BAZ_MUSIC_OK;
        }
    }
    if constexpr (M <= 8) {    // one 16x16 Gram tile holds 16/(2m) items
        constexpr int IPT = 16 / (2 * M);
        const uint32_t ntiles = (batch + IPT - 1) / IPT;
        const uint32_t blocks = std::min<uint32_t>((ntiles + 3) / 4, 256u * 32u);
        hipLaunchKernelGGL((cov_mfma_kernel<M>), dim3(blocks), dim3(256), 0, c->stream, d_in, dR, batch, c->K);
    } else {                   // 2x2 tiles per item
        const uint32_t blocks = std::min<uint32_t>((batch + 3) / 4, 256u * 32u);
        hipLaunchKernelGGL((cov_mfma2_kernel<M>), dim3(blocks), dim3(256), 0, c->stream, d_in, dR, batch, c->K);
    }
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

#ifdef BAZ_MUSIC_QUICK      // lab: a library that only knows m = 4, 8, 16 (a sixth of the compile time; tests/lab iterations)
#define BAZ_M_CASES(CALL) case 4: return CALL(4); case 8: return CALL(8); case 16: return CALL(16);
#else
#define BAZ_M_CASES(CALL)                                                                          \
    case 2: return CALL(2); case 3: return CALL(3); case 4: return CALL(4); case 5: return CALL(5); \
    case 6: return CALL(6); case 7: return CALL(7); case 8: return CALL(8); case 9: return CALL(9); \
    case 10: return CALL(10); case 11: return CALL(11); case 12: return CALL(12); case 13: return CALL(13); \
    case 14: return CALL(14); case 15: return CALL(15); case 16: return CALL(16);
#endif

int launch_cov(baz_music_ctx* c, const float* d_in, uint32_t batch, double2* dR)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_COV);
#define BAZ_CALL(MV) launch_cov_t<MV>(c, d_in, batch, dR)
    switch (c->m) {
        BAZ_M_CASES(BAZ_CALL)
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
#undef BAZ_CALL
}

template <int M>
int launch_evd_t(baz_music_ctx* c, const double2* dR, uint32_t batch, double* dQ, uint32_t qstride, double* dG)
{
    if constexpr (M <= 4) {   // one item per lane, register resident
        const uint32_t blocks = (batch + 63) / 64;
        hipLaunchKernelGGL((evd_proj_kernel<M>), dim3(blocks), dim3(64), 0, c->stream, dR, dQ, batch, c->n, qstride, dG);
    } else {                  // M lanes per item, matrices in LDS
        constexpr uint32_t IPW = 64 / M;
        const uint32_t blocks = (batch + IPW - 1) / IPW;
        // few emitters: the signal subspace by orthogonal iteration (evd_sub_kernel), then the Jacobi only for the items
        // it hands back (gap too small, zero / non-finite R)
        // (the scan's short form wants the signal vectors of the product's own workspace only: dQ == c->dQ)
        double* ss = (dQ == c->dQ && batch <= c->cap) ? c->dSs : nullptr;
        // ... and when that scan runs, nothing reads the projector coefficients: they are not written either
        double* const dQw = (ss && short_form_in_use(c)) ? nullptr : dQ;
        const uint8_t* only = nullptr;
        if (c->sub_evd && c->n <= 4 && 2 * c->n <= (uint32_t)M && c->dRedo && batch <= c->cap) {
            constexpr uint32_t GS = M <= 8 ? 8 : 16, IPS = 64 / GS;
            const uint32_t sblocks = (batch + IPS - 1) / IPS;
            if (c->n == 1) hipLaunchKernelGGL((evd_sub_kernel<M, 1>), dim3(sblocks), dim3(64), 0, c->stream, dR, dQw, batch, qstride, dG, c->dRedo, ss);
            else if (c->n == 2) hipLaunchKernelGGL((evd_sub_kernel<M, 2>), dim3(sblocks), dim3(64), 0, c->stream, dR, dQw, batch, qstride, dG, c->dRedo, ss);
            else if (c->n == 3) { if constexpr (M >= 6) hipLaunchKernelGGL((evd_sub_kernel<M, 3>), dim3(sblocks), dim3(64), 0, c->stream, dR, dQw, batch, qstride, dG, c->dRedo); }
            else { if constexpr (M >= 8) hipLaunchKernelGGL((evd_sub_kernel<M, 4>), dim3(sblocks), dim3(64), 0, c->stream, dR, dQw, batch, qstride, dG, c->dRedo); }
            HIP_TRY(c, hipGetLastError());
            only = c->dRedo;
        }
        hipLaunchKernelGGL((evd_proj_lds_kernel<M>), dim3(blocks), dim3(64), 0, c->stream, dR, dQw, batch, c->n, qstride, dG, only, ss);
    }
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

// m = 4, K % 256 == 0: covariance and EVD in one kernel (cov4_evd_kernel); d_R_dbg optionally receives R (test tap)
int launch_covevd(baz_music_ctx* c, const float* d_in, uint32_t batch, double* dQ, uint32_t qstride, double* dG,
                  double2* d_R_dbg = nullptr)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_COV);
    // items per wave task: 64 where that still makes >= 256 tasks (one per CU), else 32 / 16 -- a small batch needs more waves reading than
    // lanes rotating (a host-fed 1,024-item call read its input over PCIe at 40 GB/s with 16 waves; cov4_evd_kernel)
    const uint32_t ti = c->covevd_task_items ? (uint32_t)c->covevd_task_items : (batch >= 16384u ? 64u : (batch >= 8192u ? 32u : 16u));
    const uint32_t ntasks = (batch + ti - 1) / ti;
    const uint32_t blocks = std::min<uint32_t>((ntasks + 3) / 4, c->covevd_blocks);
    hipLaunchKernelGGL(cov4_evd_kernel, dim3(blocks), dim3(256), 0, c->stream, d_in, dQ, dG, d_R_dbg, batch, c->K, c->n,
                       qstride, ti);
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

int launch_evd(baz_music_ctx* c, const double2* dR, uint32_t batch, double* dQ, uint32_t qstride, double* dG)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_EVD);
#define BAZ_CALL(MV) launch_evd_t<MV>(c, dR, batch, dQ, qstride, dG)
    switch (c->m) {
        BAZ_M_CASES(BAZ_CALL)
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
#undef BAZ_CALL
}

// Launch geometry of the scan: rows are taken class by class (class k = items nclass*j + k, see the kernel's ROW
// CLASSES note), every class padded to a multiple of 64 rows (one block = 4 waves x 16 rows of ONE class), times
// `nsplit` ranges of 64-bin steps, chosen so that a launch has >= ~8 waves/SIMD worth of wave tasks even for small
// batches / long tables (config 3: 4,096 items x 36,000 bins).  Every range of every item yields NMAX candidate keys
// for topn_merge_kernel.
struct ScanGeom {
    uint32_t groups, nsplit, blocks, rows_per_class;
};

ScanGeom scan_geometry(uint32_t batch, uint32_t nsteps, uint32_t nclass, int force_nsplit, uint32_t m)
{
    ScanGeom G;
    G.rows_per_class = round_up((batch + nclass - 1) / nclass, 64);
    G.groups = nclass * (G.rows_per_class / 16);
    const uint32_t live_groups = (batch + 15) / 16;
    // Wave tasks wanted = 2 x the waves the chip holds at once (256 CUs x 4 SIMDs x 4): enough to balance the tail,
    // and no more -- every extra range restarts the top-n lists (the gate fires until they fill), writes another
    // candidate list per item and multiplies the row streams written at once: 262,144 cfg2 items ran the scan in 0.716 /
    // 0.758 / 0.826 ms at 1 / 2 / 4 ranges (round 1 asked for 8 waves/SIMD worth = 32,768 tasks, i.e. 2 ranges there).
    // From 9 antennas on the kernel holds 2 waves per SIMD (256 registers) and a task costs more to set up (the short form's
    // 32 coefficient registers): ONE round of the 2,048 wave slots is best there -- 16 antennas, 3,600 bins: 16,384 items ran
    // the scan in 0.358 / 0.278 / 0.294 / 0.321 ms at 1 / 2 / 4 / 8 ranges, 4,096 items in 0.093 / 0.088 / 0.095 / 0.107 ms at
    // 4 / 8 / 16 / 32 (profiles/r03_bin_ranges_by_shape.txt; 5..8 antennas do not care between 2 and 8).
    const uint32_t want_tasks = (m >= 9) ? 256u * 4u * 2u : 256u * 4u * 4u * 2u;
    uint32_t ns = (want_tasks + live_groups - 1) / live_groups;
    G.nsplit = std::max<uint32_t>(1u, std::min<uint32_t>(ns, std::min<uint32_t>(nsteps, 64u)));
    if (force_nsplit > 0) G.nsplit = std::min<uint32_t>((uint32_t)force_nsplit, std::min<uint32_t>(nsteps, 64u));   // tests / lab
    G.blocks = (G.groups / 4) * G.nsplit;
    return G;
}

// Bin ranges per item of scan_i8_kernel: its workgroups (4 waves x 16 items, one range of 64-bin steps) should fill the
// slots the chip holds at once -- `slots` = CUs x resident workgroups per CU -- a whole number of times: 16,384 config-3 items
// are 256 groups; 8 ranges made 2,048 workgroups for 768 slots, 2.67 rounds of which the last ran a third empty, where 3 ranges
// fill every slot exactly once (and restart the top-n lists 3 times instead of 8).  Up to 8 rounds are tried; the fewest
// rounds within 3 % of the best filling win.  Ranges of fewer than 4 steps are not made.
constexpr uint32_t I8_MAX_NSPLIT = 16;
uint32_t i8_nsplit(uint32_t batch, uint32_t nsteps, uint32_t slots, int force_nsplit)
{
    const uint32_t groups = std::max<uint32_t>(1u, (batch + 63) / 64);
    const uint32_t cap = std::max<uint32_t>(1u, std::min<uint32_t>(I8_MAX_NSPLIT, nsteps / 4));
    if (force_nsplit > 0) return std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)force_nsplit, std::min<uint32_t>(nsteps, 64u)));
    uint32_t best = 1;
    double best_fill = 0.0;
    for (uint32_t r = 1; r <= 8; ++r) {
        const uint32_t ns = std::max<uint32_t>(1u, std::min<uint32_t>(cap, (uint32_t)(((uint64_t)r * slots) / groups)));
        const uint64_t blocks = (uint64_t)groups * ns;
        const uint64_t rounds = (blocks + slots - 1) / slots;
        const double fill = (double)blocks / (double)(rounds * slots);
        if (fill > best_fill * 1.03) { best_fill = fill; best = ns; }
        if (ns == cap) break;
    }
    return best;
}

int ensure_candidates(baz_music_ctx* c, size_t entries)
{
    if (entries <= c->cand_cap) return BAZ_MUSIC_OK;
    if (c->dCand) {                                        // (see ensure_workspace: drain what may still write the old lists)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)dev_free(c->dCand);
        c->dCand = nullptr;
    }
    c->cand_cap = 0;
    HIP_TRY(c, dev_malloc((void**)&c->dCand, entries * sizeof(double)));
    c->cand_cap = entries;
    return BAZ_MUSIC_OK;
}

// The coarse-gated scan applies: spectrum port not wired, m <= 8, n <= 4, the table's scale representable.
bool coarse_applies(const baz_music_ctx* c)
{
    // (one emitter from 6 antennas on: the full scan is the SHORT form, a quarter of the projector GEMM's work, and the
    // gated scan's exact tiles are the projector form -- not the same bits; that case keeps the full scan)
    return c->coarse && c->cs_ok && c->dCS && c->m <= 8 && c->n <= 4 && !c->lab_variant && !short_form_applies(c->m, c->n);
}

constexpr int coarse_rg_wide(int m, int nmax) { return (cs_groups(m) == 4 && nmax > 2) ? 1 : 2; }   // (what the register file holds without spilling)

// its launch geometry: a workgroup = 4 waves x RG x 16 items; ranges of table phases so that small batches still fill the chip
constexpr uint32_t COARSE_WANT_BLOCKS = 1024;
struct CoarseGeom {
    uint32_t groups, nsplit, nphases, tpp;
};
CoarseGeom coarse_geometry(const baz_music_ctx* c, uint32_t batch)
{
    CoarseGeom G;
    // four row groups per wave where the lists fit the register file beside them (n <= 2), else two
    // (m >= 5: 2 .. 4 operand groups per tile -- two row groups while the register file holds them, else one)
    const uint32_t rg = (c->m > 4) ? (uint32_t)coarse_rg_wide((int)c->m, c->n <= 2 ? 2 : 4)
                                   : ((c->coarse_rg == 2 || c->n > 2) ? 2u : 4u);
    G.tpp = (rg <= 2) ? 4u : 8u;
    G.nphases = c->cs_tiles / G.tpp;
    G.groups = (batch + 64 * rg - 1) / (64 * rg);
    const uint32_t ns = (COARSE_WANT_BLOCKS + G.groups - 1) / G.groups;
    G.nsplit = std::max(1u, std::min(ns, std::min(G.nphases, 16u)));
    if (c->force_nsplit > 0) G.nsplit = std::max(1u, std::min((uint32_t)c->force_nsplit, std::min(G.nphases, 16u)));
    return G;
}

// ---- sorting in front of the gated scan (sort_kernels.hip.h): LAB builds only --------------------------------------------------------
#ifdef BAZ_MUSIC_LAB
constexpr uint32_t SORT_MIN_BATCH = 4096;      // below this a launch is a handful of workgroups either way
constexpr double SORT_ON_RATE = 0.06;          // share of (row group, tile) pairs an UNSORTED call evaluated exactly above which sorting pays
constexpr uint32_t SORT_PROBE_EVERY = 64;      // while sorting: every so many calls one call unsorted, whose statistic decides anew

int ensure_sort_workspace(baz_music_ctx* c, uint32_t batch)
{
    if (c->sort_mode == 0) return BAZ_MUSIC_OK;                   // (nothing of the sorting exists unless it was asked for)
    if (!c->dFire) {
        HIP_TRY(c, dev_malloc((void**)&c->dFire, 2 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemsetAsync(c->dFire, 0, 2 * sizeof(unsigned long long), c->stream));
        HIP_TRY(c, hipHostMalloc((void**)&c->hFire, 4 * sizeof(unsigned long long), hipHostMallocDefault));
        std::memset(c->hFire, 0, 4 * sizeof(unsigned long long));
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->hFireDev, c->hFire, 0));
    }
    if (!c->dKT) return BAZ_MUSIC_OK;
    if (!c->dHist) {
        HIP_TRY(c, dev_malloc((void**)&c->dHist, bazsort::KEY_BUCKETS * sizeof(uint32_t)));
        HIP_TRY(c, dev_malloc((void**)&c->dCursor, bazsort::KEY_BUCKETS * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->dHist, 0, bazsort::KEY_BUCKETS * sizeof(uint32_t), c->stream));
    }
    if (batch > c->sort_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->dKeys) (void)dev_free(c->dKeys);
        if (c->dPerm) (void)dev_free(c->dPerm);
        c->dKeys = nullptr; c->dPerm = nullptr; c->sort_cap = 0;
        const uint32_t cap = round_up(batch, 4096);
        HIP_TRY(c, dev_malloc((void**)&c->dKeys, (size_t)cap * sizeof(uint16_t)));
        HIP_TRY(c, dev_malloc((void**)&c->dPerm, (size_t)cap * sizeof(uint32_t)));
        c->sort_cap = cap;
    }
    return BAZ_MUSIC_OK;
}

// Sort this call's items?  The gated scan's merge leaves, in page-locked memory, how many of its (row group, tile) pairs were evaluated exactly and
// whether that call was sorted; the answer of some EARLIER call is there when this one is launched (nothing waits for it).
//   * an unsorted call with a high share starts a trial (sorting on);
//   * a sorted call must bring the share clearly below the unsorted one (x 0.6), else sorting goes off again and is not tried for a while
//     (a small coherent batch cut into many bin ranges has a high share that no order improves);
//   * while sorting, every SORT_PROBE_EVERY-th call runs unsorted: a low share there (the stream turned coherent) switches it off.
// Up to 4 antennas (the rows' coefficients sit in registers; from 5 on the exact tiles fetch them from L2, where an index list costs gathers).
bool sort_decide(baz_music_ctx* c, uint32_t batch)
{
    if (batch < SORT_MIN_BATCH || !c->dKT || !c->hFire || c->m > 4 || c->n > 2 || c->coarse_rg == 2 || c->coarse_lab) return false;
    if (c->sort_mode >= 0) return c->sort_mode != 0;
    ++c->sort_clock;
    const unsigned long long tag = reinterpret_cast<volatile unsigned long long*>(c->hFire)[2];
    if (tag != c->fire_seen) {
        c->fire_seen = tag;
        const unsigned long long fired = reinterpret_cast<volatile unsigned long long*>(c->hFire)[0];
        const unsigned long long walked = reinterpret_cast<volatile unsigned long long*>(c->hFire)[1];
        const double rate = walked ? (double)fired / (double)walked : 0.0;
        if (tag & 1ull) {                              // a sorted call
            if (c->sort_on && rate > 0.6 * c->rate_unsorted) {
                c->sort_on = false;
                c->sort_retry_at = c->sort_clock + 512;
            }
        } else {
            c->rate_unsorted = rate;
            if (rate <= SORT_ON_RATE) c->sort_on = false;
            else if (!c->sort_on && c->sort_clock >= c->sort_retry_at) c->sort_on = true;
        }
    }
    if (c->sort_on && ++c->sort_streak >= SORT_PROBE_EVERY) {
        c->sort_streak = 0;
        return false;                                  // the probe
    }
    return c->sort_on;
}

template <int M>
int launch_sort_t(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch)
{
    if constexpr (M <= 8) {
        const uint32_t ns = bazsort::key_samples(c->res);
        const uint32_t padded = round_up(batch, 256);          // whole workgroups of the key kernel = whole 8-key loads of the sweeps
        hipLaunchKernelGGL((bazsort::coarse_key_kernel<M>), dim3(padded / 256), dim3(256), 0, c->stream, dQ, c->dKT, ns, batch, qstride, c->dKeys);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL((bazsort::key_sweep_kernel<false>), dim3(bazsort::KEY_BLOCKS), dim3(256), 0, c->stream, c->dKeys, padded / 8, c->dHist,
                           c->dCursor, c->dPerm);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL((bazsort::key_sweep_kernel<true>), dim3(bazsort::KEY_BLOCKS), dim3(256), 0, c->stream, c->dKeys, padded / 8, c->dHist,
                           c->dCursor, c->dPerm);
        HIP_TRY(c, hipGetLastError());
    }
    return BAZ_MUSIC_OK;
}
#endif   // BAZ_MUSIC_LAB

template <int M, int NMAX>
int launch_scan_t(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                  float* d_lvl, float* d_spec)
{
    if constexpr (M <= 8 && NMAX <= 4) {
        if (!d_spec && coarse_applies(c) && dQ == c->dQ) {
            const CoarseGeom CG = coarse_geometry(c, batch);
            if ((size_t)batch * CG.nsplit * NMAX > c->cand_cap) return BAZ_MUSIC_E_INVALID;
            ScanRefine rf;
            rf.Gs = c->refine_off ? nullptr : c->dG;
            rf.TB = c->dTB + c->tb_step_elems;
            rf.below = c->refine_below;
            rf.count = c->refine_nocount ? nullptr : c->dRefined + c->stat_parity;
            rf.A2 = nullptr;
            unsigned long long* stats = c->coarse_stats ? c->dMargin : nullptr;     // lab: exact tile evaluations, summed over launches
#ifdef BAZ_MUSIC_LAB
            // lab (BAZ_MUSIC_SORT): the items in an order in which neighbours share their nulls, while the scan's own statistic says that pays
            int sr = ensure_sort_workspace(c, batch);
            if (sr) return sr;
            const bool sorted = sort_decide(c, batch);
            if (sorted) {
                sr = launch_sort_t<M>(c, dQ, qstride, batch);
                if (sr) return sr;
                ++c->sorted_calls;
            } else {
                ++c->unsorted_calls;
            }
            c->last_gated_sorted = sorted;
            const uint32_t* perm = sorted ? c->dPerm : nullptr;
            unsigned long long* fstat = c->dFire;
#else
            const uint32_t* perm = nullptr;                 // the product walks the items in their own order and keeps no fire statistic
            unsigned long long* fstat = nullptr;
#endif
#define BAZ_COARSE_ARGS dim3(CG.groups * CG.nsplit), dim3(256), 0, c->stream, dQ, c->dCS, c->dCS + (size_t)(c->cs_tiles + 1) * cs_c_units(M), \
                        c->dCand, batch, c->res, qstride, CG.nphases, CG.nsplit, c->keep_mask, c->n, rf, c->cs, stats, nullptr, perm, fstat
#ifdef BAZ_MUSIC_LAB
            if (c->sort_mode != 0 && M <= 4 && CG.tpp != 4 && !c->coarse_lab) {        // lab (BAZ_MUSIC_SORT): the index list and the fire statistic
                if constexpr (M <= 4) hipLaunchKernelGGL((scan_coarse_kernel<M, NMAX, 4, 8, false, 0, true>), BAZ_COARSE_ARGS);
            } else
#endif
            if constexpr (M > 4) hipLaunchKernelGGL((scan_coarse_kernel<M, NMAX, coarse_rg_wide(M, NMAX), 4>), BAZ_COARSE_ARGS);
            else if (CG.tpp == 4) hipLaunchKernelGGL((scan_coarse_kernel<M, NMAX, 2, 4>), BAZ_COARSE_ARGS);
#ifdef BAZ_MUSIC_LAB
            else if (c->coarse_lab == 1) hipLaunchKernelGGL((scan_coarse_kernel<M, NMAX, 4, 8, false, 1>), BAZ_COARSE_ARGS);
            else if (c->coarse_lab == 2) hipLaunchKernelGGL((scan_coarse_kernel<M, NMAX, 4, 8, false, 2>), BAZ_COARSE_ARGS);
#endif
            else hipLaunchKernelGGL((scan_coarse_kernel<M, NMAX, 4, 8>), BAZ_COARSE_ARGS);
#undef BAZ_COARSE_ARGS
            HIP_TRY(c, hipGetLastError());
            c->last_nsplit = CG.nsplit;
            c->scan_kind = 2;
            return BAZ_MUSIC_OK;
        }
    }
    ScanGeom G = scan_geometry(batch, c->fb_steps, c->nclass, c->force_nsplit, c->m);
    c->last_nsplit = G.nsplit;
    double* cand = c->dCand;
    if ((size_t)batch * G.nsplit * NMAX > c->cand_cap) return BAZ_MUSIC_E_INVALID;   // reserve_candidates() sized it
    const bool spec = d_spec != nullptr;
    const bool vec4 = (c->res % 4u) == 0 && (reinterpret_cast<uintptr_t>(d_spec) % 16u) == 0;
#ifdef BAZ_MUSIC_LAB
    if constexpr (M <= 4 && NMAX <= 4) {
        // LAB (BAZ_MUSIC_I8P=1): 2 .. 4 antennas with the spectrum port: the int8 matrix core with level-packed operands (scan_i8p_kernels.hip.h); same
        // geometry (row classes, bin ranges) and candidate lists as the fp64 scan below.  Without the port the coarse-gated scan above.
        if (spec && i8p_active(c) && dQ == c->dQ) {
            ScanRefine rf;
            rf.Gs = c->refine_off ? nullptr : c->dG;
            rf.TB = c->dTB + c->tb_step_elems;
            rf.below = c->refine_below;
            rf.count = c->refine_nocount ? nullptr : c->dRefined + c->stat_parity;
            rf.A2 = nullptr;
            const uint4* p1 = c->dIP + I8P_STEP_UNITS;                            // step 0 (a padded step lies in front)
            const uint4* p2 = p1 + i8p_operand_units(c->fb_steps);
#define BAZ_I8P_ARGS dim3(G.blocks), dim3(256), 0, c->stream, dQ, p1, p2, c->dFB + c->fb_step_elems, d_spec, cand, batch, c->res, qstride, \
                     G.nsplit, c->nclass, G.rows_per_class, c->keep_mask, c->n, rf, c->i8, c->dI8Stat, nullptr
#ifdef BAZ_MUSIC_LAB
            if constexpr (M == 4 && NMAX == 2) {       // lab: timing ablations (wrong results)
                if (vec4 && c->i8_abl) {
                    if (c->i8_abl == 1) hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, true, false, 1>), BAZ_I8P_ARGS);
                    else if (c->i8_abl == 2) hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, true, false, 2>), BAZ_I8P_ARGS);
                    else if (c->i8_abl == 4) hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, true, false, 4>), BAZ_I8P_ARGS);
                    else if (c->i8_abl == 5) hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, true, false, 5>), BAZ_I8P_ARGS);
                    else hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, true, false, 3>), BAZ_I8P_ARGS);
                    HIP_TRY(c, hipGetLastError());
                    c->scan_kind = 3;
                    return BAZ_MUSIC_OK;
                }
            }
#endif
            if (vec4) hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, true>), BAZ_I8P_ARGS);
            else hipLaunchKernelGGL((scan_i8p_kernel<M, NMAX, true, false>), BAZ_I8P_ARGS);
#undef BAZ_I8P_ARGS
            HIP_TRY(c, hipGetLastError());
            c->scan_kind = 3;
            return BAZ_MUSIC_OK;
        }
    }
#endif
    if constexpr (M >= 6 && NMAX <= 4) {
        // the bulk of the values on the int8 matrix core, exactly accumulated; steps with a value under the accuracy
        // threshold in this kernel's own fp64 form (scan_i8_kernels.hip.h).  Same launch geometry (nclass = 1 from m = 6 on).
        // (5 .. 8 antennas without the spectrum port belong to the coarse-gated scan above; BAZ_MUSIC_COARSE=0 is its A/B and
        // must stay bit-identical to it: the fp64 scan below.  One emitter from 6 antennas on has no gated scan: here.)
        const bool gated_shape = M <= 8 && NMAX <= 4 && !short_form_applies(c->m, c->n);
        if (i8_active(c) && dQ == c->dQ && (spec || !gated_shape)) {
            ScanRefine rf;
            rf.Gs = c->refine_off ? nullptr : c->dG;
            rf.TB = c->dTB + c->tb_step_elems;
            rf.below = c->refine_below;
            rf.count = c->refine_nocount ? nullptr : c->dRefined + c->stat_parity;
            rf.A2 = nullptr;
            // its own bin ranges: whole rounds of the resident workgroup slots (i8_nsplit)
            int& per_cu = c->i8_wgs_per_cu[NMAX > 2 ? 1 : 0][spec ? 1 : 0][vec4 ? 1 : 0];
            if (per_cu == 0) {
                int occ = 0;
                hipError_t oe;
                if (spec && vec4) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_i8_kernel<M, NMAX, true, true>, 256, 0);
                else if (spec) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_i8_kernel<M, NMAX, true, false>, 256, 0);
                else oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_i8_kernel<M, NMAX, false, false>, 256, 0);
                per_cu = (oe == hipSuccess && occ > 0) ? occ : 2;
                (void)hipGetLastError();
            }
            G.nsplit = i8_nsplit(batch, c->fb_steps, c->num_cus * (uint32_t)per_cu, c->force_nsplit);
            G.blocks = (G.groups / 4) * G.nsplit;
            c->last_nsplit = G.nsplit;
            if ((size_t)batch * G.nsplit * NMAX > c->cand_cap) return BAZ_MUSIC_E_INVALID;
#define BAZ_I8_LAUNCH(SPEC, VEC4)                                                                                       \
    hipLaunchKernelGGL((scan_i8_kernel<M, NMAX, SPEC, VEC4>), dim3(G.blocks), dim3(256), 0, c->stream, dQ, c->dIB,      \
                       c->dIB + i8_image_bytes5(c->m, c->fb_steps) / 16, c->dFB + c->fb_step_elems, d_spec, cand, batch, c->res, qstride, G.nsplit, c->keep_mask, c->n, rf, \
                       c->i8, c->dI8Stat, nullptr)
#ifdef BAZ_MUSIC_LAB
            if constexpr ((M == 8 || M == 16) && NMAX == 2) {        // lab: ablations of the bulk loop (timing only, wrong results)
                if (spec && vec4 && c->i8_abl) {
#define BAZ_I8_ABL(ABLV)                                                                                                  \
    hipLaunchKernelGGL((scan_i8_kernel<M, NMAX, true, true, false, ABLV>), dim3(G.blocks), dim3(256), 0, c->stream, dQ, c->dIB, \
                       c->dIB + i8_image_bytes5(c->m, c->fb_steps) / 16, c->dFB + c->fb_step_elems, d_spec, cand, batch, c->res, \
                       qstride, G.nsplit, c->keep_mask, c->n, rf, c->i8, c->dI8Stat, nullptr)
                    switch (c->i8_abl) {
                        case 1: BAZ_I8_ABL(1); break;      // no spectrum stores
                        case 4: BAZ_I8_ABL(4); break;      // no MFMAs
                        case 5: BAZ_I8_ABL(5); break;
                        case 33: BAZ_I8_ABL(33); break;    // staging and barriers alone
                        case 256: BAZ_I8_ABL(256); break;  // the steps left to right (round 4's walk): A/B of the strided walk
                        case 257: BAZ_I8_ABL(257); break;  // ... without the stores
                        case 9: BAZ_I8_ABL(9); break;      // first phase staged only, no stores: the tiles' arithmetic alone
                        default: BAZ_I8_ABL(32); break;    // staging, barriers and stores alone
                    }
#undef BAZ_I8_ABL
                    HIP_TRY(c, hipGetLastError());
                    return BAZ_MUSIC_OK;
                }
            }
#endif
            if (spec && vec4) BAZ_I8_LAUNCH(true, true);
            else if (spec) BAZ_I8_LAUNCH(true, false);
            else BAZ_I8_LAUNCH(false, false);
#undef BAZ_I8_LAUNCH
            HIP_TRY(c, hipGetLastError());
            c->scan_kind = 1;
            return BAZ_MUSIC_OK;
        }
    }
    c->scan_kind = 0;
    ScanRefine rf;
    rf.Gs = c->refine_off ? nullptr : c->dG;
    rf.TB = c->dTB + c->tb_step_elems;
    rf.below = c->refine_below;
    rf.count = c->refine_nocount ? nullptr : c->dRefined + c->stat_parity;
    rf.A2 = c->dA2p ? c->dA2p + 64 : nullptr;
    const double2* fb0 = c->dFB + c->fb_step_elems;   // step 0 (a padded step lies in front)
    if constexpr (M >= 6 && NMAX == 2) {
        // one or two emitters: the short form ||a||^2 - sum_c |s_c^H a|^2 (scan_mfma_kernel, SIG) where it needs fewer
        // MFMAs than the projector GEMM: n = 2 from 9 antennas, n = 1 from 6
        if (short_form_in_use(c) && dQ == c->dQ) {
            const double2* tb0 = c->dTB + c->tb_step_elems;
#define BAZ_SIG_LAUNCH(SPEC, VEC4, SIGV)                                                                                    \
    hipLaunchKernelGGL((scan_mfma_kernel<M, NMAX, SPEC, VEC4, 0, (1 | 2 | 16), SIGV>), dim3(G.blocks), dim3(256), 0, c->stream, \
                       c->dSs, tb0, d_spec, cand, batch, c->res, qstride, G.nsplit, c->nclass, G.rows_per_class, c->keep_mask, c->n, rf, (uint32_t)c->seq_walk)
            if (c->n == 2) {
                if constexpr (M >= 9) {
                    if (spec && vec4) BAZ_SIG_LAUNCH(true, true, 2);
                    else if (spec) BAZ_SIG_LAUNCH(true, false, 2);
                    else BAZ_SIG_LAUNCH(false, false, 2);
                }
            } else {
                if (spec && vec4) BAZ_SIG_LAUNCH(true, true, 1);
                else if (spec) BAZ_SIG_LAUNCH(true, false, 1);
                else BAZ_SIG_LAUNCH(false, false, 1);
            }
#undef BAZ_SIG_LAUNCH
            HIP_TRY(c, hipGetLastError());
            return BAZ_MUSIC_OK;
        }
    }
#define BAZ_SCAN_ARGS dQ, fb0, d_spec, cand, batch, c->res, qstride, G.nsplit, c->nclass, G.rows_per_class, c->keep_mask, c->n, rf, (uint32_t)c->seq_walk
#define BAZ_SCAN_LAUNCH(SPEC, VEC4, ABLV, AUXV)                                                                    \
    hipLaunchKernelGGL((scan_mfma_kernel<M, NMAX, SPEC, VEC4, ABLV, AUXV>), dim3(G.blocks), dim3(256), (size_t)c->scan_lds_pad, c->stream, \
                       BAZ_SCAN_ARGS)
#ifdef BAZ_MUSIC_LAB
    if constexpr (M == 4 && NMAX == 2) {   // lab switches for the A/Bs in profiles/HISTORY_r01_r02.md 5.3 (BAZ_MUSIC_SCAN_VARIANT)
        if (spec && vec4 && c->lab_variant) {
            switch (c->lab_variant) {
                case 2: BAZ_SCAN_LAUNCH(true, true, 64, (1 | 2 | 16)); break;   // ungated top-n network (round 1)
                case 3: BAZ_SCAN_LAUNCH(true, true, 0, 0); break;               // plain cached spectrum stores
                case 4: BAZ_SCAN_LAUNCH(true, true, 0, 2); break;               // nt
                case 5: BAZ_SCAN_LAUNCH(true, true, 0, (1 | 16)); break;        // sc0 sc1
                // (6-8: timing only, wrong results -- tests/lab/scan_ablate.py, profiles/r03_scan_ablation_real_inputs.txt)
                case 6: BAZ_SCAN_LAUNCH(true, true, 1, (1 | 2 | 16)); break;              // everything but the spectrum stores
                case 7: BAZ_SCAN_LAUNCH(true, true, (8 | 2 | 4), (1 | 2 | 16)); break;    // stores + staging + barriers only
                case 8: BAZ_SCAN_LAUNCH(true, true, (1 | 2), (1 | 2 | 16)); break;        // MFMA + conversions, no top-n, no stores
                // (9 - 11, round 5: A/B of the rotating LDS-DMA loader; results are right)
                case 9: BAZ_SCAN_LAUNCH(true, true, 1024, (1 | 2 | 16)); break;           // rotating loader (negative: profiles/r05_loader_ab.txt)
                case 10: BAZ_SCAN_LAUNCH(true, true, (1024 | 2048), (1 | 2 | 16)); break;  // rotating loader at 3 waves per SIMD (168 registers: no spills)
                case 11: BAZ_SCAN_LAUNCH(true, true, 2048, (1 | 2 | 16)); break;          // register staging (the product's) at 3 waves per SIMD
                case 12: BAZ_SCAN_LAUNCH(true, true, 4096, (1 | 2 | 16)); break;          // blocks in the compact order (negative: profiles/r05_write_order.txt)
                case 13: BAZ_SCAN_LAUNCH(true, true, (8 | 2 | 4 | 4096), (1 | 2 | 16)); break;   // stores + staging + barriers only, compact order
                case 14: BAZ_SCAN_LAUNCH(true, true, 8192, (1 | 2 | 16)); break;          // (round 6) s_setprio 3 around the MFMAs of a step
                case 15: BAZ_SCAN_LAUNCH(true, true, 16384, (1 | 2 | 16)); break;         // ... s_setprio 3 around epilogue, staging and stores instead
                default: BAZ_SCAN_LAUNCH(true, true, 0, (1 | 2 | 16)); break;
            }
            HIP_TRY(c, hipGetLastError());
            return BAZ_MUSIC_OK;
        }
    }
#endif
    if (spec && vec4) BAZ_SCAN_LAUNCH(true, true, 0, (1 | 2 | 16));
    else if (spec) BAZ_SCAN_LAUNCH(true, false, 0, (1 | 2 | 16));
    else BAZ_SCAN_LAUNCH(false, false, 0, (1 | 2 | 16));
#undef BAZ_SCAN_LAUNCH
#undef BAZ_SCAN_ARGS
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

template <int NMAX>
int launch_merge_t(baz_music_ctx* c, uint32_t batch, float* d_ang, float* d_lvl, float* d_spec)
{
#ifdef BAZ_MUSIC_LAB
    // lab (BAZ_MUSIC_SORT): the gated scan just ran with its fire statistic on -- it travels to page-locked memory with this merge
    const bool gated = c->scan_kind == 2 && c->sort_mode != 0 && c->dFire && c->hFireDev;
    const unsigned long long tag = gated ? ((++c->fire_calls) << 1) | (c->last_gated_sorted ? 1ull : 0ull) : 0ull;
    hipLaunchKernelGGL((topn_merge_kernel<NMAX>), dim3((batch + 255) / 256), dim3(256), 0, c->stream, c->dCand,
                       d_spec, d_ang, d_lvl, batch, c->res, c->n, c->last_nsplit, c->keep_mask, c->dRefined + (c->stat_parity ^ 1),
                       gated ? c->dFire : nullptr, gated ? c->hFireDev : nullptr, tag);
#else
    hipLaunchKernelGGL((topn_merge_kernel<NMAX>), dim3((batch + 255) / 256), dim3(256), 0, c->stream, c->dCand,
                       d_spec, d_ang, d_lvl, batch, c->res, c->n, c->last_nsplit, c->keep_mask, c->dRefined + (c->stat_parity ^ 1),
                       nullptr, nullptr, 0ull);
#endif
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

template <int M>
int launch_scan_m(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                  float* d_lvl, float* d_spec)
{
    if (c->n <= 2) return launch_scan_t<M, 2>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
    if (c->n <= 4) return launch_scan_t<M, 4>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
    if constexpr (M > 5) {
        if (c->n <= 8) return launch_scan_t<M, 8>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
    }
    if constexpr (M > 9) return launch_scan_t<M, 16>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
    return BAZ_MUSIC_E_UNSUPPORTED;   // unreachable: n < m
}

uint32_t topn_list_len(uint32_t n) { return n <= 2 ? 2u : (n <= 4 ? 4u : (n <= 8 ? 8u : 16u)); }

// candidate keys one scan launch over `nb` items produces (mirrors launch_scan_t's geometry)
size_t cand_entries(const baz_music_ctx* c, uint32_t nb)
{
    size_t per_item = scan_geometry(nb, c->fb_steps, c->nclass, c->force_nsplit, c->m).nsplit;
    if (c->m <= 8 && c->dCS) per_item = std::max<size_t>(per_item, coarse_geometry(c, nb).nsplit);
    if (c->dIB) per_item = std::max<size_t>(per_item, c->force_nsplit > 0 ? (size_t)std::min<uint32_t>((uint32_t)c->force_nsplit, 64u) : I8_MAX_NSPLIT);
    return (size_t)nb * per_item * topn_list_len(c->n);
}

// nb * nsplit(nb) is not monotonic in nb (nsplit = ceil(want_tasks / groups) while that is <= 64 and <= nsteps), so
// a reservation for `batch` items is sized for the worst launch of ANY nb <= batch: nb * nsplit(nb) < nb *
// (want_tasks / ceil(nb/16) + 1) <= 16 * want_tasks + nb, and <= nb * min(64, nsteps).  A smaller batch or the short
// tail chunk of baz_music_process then never re-allocates in the middle of the pipeline.
size_t cand_entries_upto(const baz_music_ctx* c, uint32_t batch)
{
    const size_t want_tasks = 256u * 4u * 4u * 2u;   // scan_geometry()
    const size_t cap_split = std::min<size_t>(64u, std::max<uint32_t>(1u, c->fb_steps));
    const size_t worst = std::min<size_t>((size_t)batch * cap_split, 16u * want_tasks + (size_t)batch);
    const size_t forced = c->force_nsplit > 0 ? (size_t)batch * std::min<size_t>((size_t)c->force_nsplit, cap_split) : 0;
    // the coarse-gated scan: nsplit = ceil(COARSE_WANT_BLOCKS / ceil(nb / (64 rg))) <= 16, rg <= 4 row groups per wave
    // ->  nb * nsplit <= 256 * WANT + nb
    const size_t coarse = (c->m <= 8) ? std::min<size_t>((size_t)batch * 16u, 256u * COARSE_WANT_BLOCKS + (size_t)batch) : 0;
    // the int8 scan: nsplit <= r slots / ceil(nb / 64) with r <= 8 rounds of <= 4 x 256-CU slots, and <= I8_MAX_NSPLIT
    const size_t i8 = c->dIB ? std::min<size_t>((size_t)batch * I8_MAX_NSPLIT, (size_t)64u * 8u * 4u * c->num_cus + (size_t)batch) : 0;
    return std::max(std::max(std::max(std::max(worst, forced), coarse), i8), (size_t)batch) * topn_list_len(c->n);
}

int reserve_candidates(baz_music_ctx* c, uint32_t batch)
{
    return ensure_candidates(c, std::max(cand_entries(c, batch), cand_entries_upto(c, batch)));
}

int launch_merge(baz_music_ctx* c, uint32_t batch, float* d_ang, float* d_lvl, float* d_spec)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_MERGE);
    switch (topn_list_len(c->n)) {
        case 2: return launch_merge_t<2>(c, batch, d_ang, d_lvl, d_spec);
        case 4: return launch_merge_t<4>(c, batch, d_ang, d_lvl, d_spec);
        case 8: return launch_merge_t<8>(c, batch, d_ang, d_lvl, d_spec);
        default: return launch_merge_t<16>(c, batch, d_ang, d_lvl, d_spec);
    }
}

template <int NMAX>
int launch_peaks_t(baz_music_ctx* c, uint32_t batch, float* d_ang, float* d_lvl, const float* d_spec)
{
    hipLaunchKernelGGL((peak_pick_kernel<NMAX>), dim3((batch + 3) / 4), dim3(256), 0, c->stream, d_spec, d_ang, d_lvl,
                       batch, c->res, c->n);
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

int launch_peaks(baz_music_ctx* c, uint32_t batch, float* d_ang, float* d_lvl, const float* d_spec)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_MERGE);
    switch (topn_list_len(c->n)) {
        case 2: return launch_peaks_t<2>(c, batch, d_ang, d_lvl, d_spec);
        case 4: return launch_peaks_t<4>(c, batch, d_ang, d_lvl, d_spec);
        case 8: return launch_peaks_t<8>(c, batch, d_ang, d_lvl, d_spec);
        default: return launch_peaks_t<16>(c, batch, d_ang, d_lvl, d_spec);
    }
}

int launch_scan(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                float* d_lvl, float* d_spec)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_SCAN);
#define BAZ_CALL(MV) launch_scan_m<MV>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec)
    switch (c->m) {
        BAZ_M_CASES(BAZ_CALL)
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
#undef BAZ_CALL
}

// ---------------------------------------------------------------------------------------------------------------
// Wide arrays (m > BAZ_MUSIC_FAST_M): the run-time-m kernels of music_wide_kernels.hip.h, one workgroup per item.
// ---------------------------------------------------------------------------------------------------------------
size_t wide_evd_lds(uint32_t m)
{
    const size_t ld = m + 1, np = (m + (m & 1u)) / 2;
    return 2 * (size_t)m * ld * sizeof(double2) + np * 6 * sizeof(double) + bazwide::WB * sizeof(double) + 2 * np * sizeof(int) +
           m * sizeof(int);
}

// items per pass: bounds the fp64 strength scratch (res doubles per item) to ~256 MiB
uint32_t wide_pass_items(const baz_music_ctx* c)
{
    const size_t per_item = (size_t)c->res * sizeof(double);
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(8192, ((size_t)256 << 20) / per_item));
}

// launch geometry of scan_wide_mfma_kernel for a pass of nb items (one place: process_wide_locked launches with it, baz_music_reserve sizes for it)
struct WideScanGeom { uint32_t ipw, nk, groups, nsplit; };
WideScanGeom wide_scan_geometry(const baz_music_ctx* c, uint32_t nb)
{
    WideScanGeom W;
    W.ipw = (c->n <= 2) ? 4u : (c->n <= 4 ? 2u : 1u);                          // items per wave (tile = ipw items x 16 / ipw outputs)
    W.nk = 8u / W.ipw;                                                         // list length
    W.groups = (nb + 4u * W.ipw - 1) / (4u * W.ipw);                           // workgroups of 4 waves
    W.nsplit = std::max(1u, std::min((1024u + W.groups - 1) / W.groups, std::min(c->fb_steps, 16u)));
    return W;
}
// candidate keys of ANY pass of up to `pass` items: nb * nsplit(nb) is not monotonic in nb (see cand_entries_upto)
size_t wide_cand_entries(const baz_music_ctx* c, uint32_t pass)
{
    const WideScanGeom W = wide_scan_geometry(c, pass);
    const size_t cap_split = std::min<uint32_t>(c->fb_steps, 16u);
    // nsplit <= 1024 / groups + 1 and groups >= nb / (4 ipw): nb * nsplit <= 4096 ipw + nb; and <= nb * cap_split
    const size_t worst = std::min<size_t>((size_t)pass * cap_split, (size_t)4096u * W.ipw + (size_t)pass);
    return std::max<size_t>(worst, (size_t)pass * W.nsplit) * W.nk;
}

int ensure_wide_workspace(baz_music_ctx* c, uint32_t items)
{
    if (items <= c->wide_cap) return BAZ_MUSIC_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->dR) { (void)dev_free(c->dR); c->dR = nullptr; }
    if (c->dGw) { (void)dev_free(c->dGw); c->dGw = nullptr; }
    if (c->dWS) { (void)dev_free(c->dWS); c->dWS = nullptr; }
    if (c->dSw) { (void)dev_free(c->dSw); c->dSw = nullptr; }
    c->wide_cap = 0;
    const size_t mm = (size_t)c->m * c->m;
    HIP_TRY(c, dev_malloc((void**)&c->dR, (size_t)items * mm * sizeof(double2)));
    HIP_TRY(c, dev_malloc((void**)&c->dGw, (size_t)items * (c->m - c->n) * c->m * sizeof(double2)));
    HIP_TRY(c, dev_malloc((void**)&c->dWS, (size_t)items * c->res * sizeof(double)));
    HIP_TRY(c, dev_malloc((void**)&c->dSw, (size_t)items * c->n * c->m * sizeof(double2)));
    if (c->dRedo) { (void)dev_free(c->dRedo); c->dRedo = nullptr; }
    HIP_TRY(c, dev_malloc((void**)&c->dRedo, (size_t)items));
    c->wide_cap = items;
    return BAZ_MUSIC_OK;
}

int launch_cov_wide(baz_music_ctx* c, const float* d_in, uint32_t nb, double2* dR)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_COV);
    if (c->wide_cov_mfma && c->m > 32) {   // 33 <= m <= 64: one wave per (item, pair of 16-antenna blocks)
        const uint32_t nblk = (c->m + 15u) / 16u, npairs = nblk * (nblk + 1u) / 2u;
        const uint64_t tasks = (uint64_t)nb * npairs;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((tasks + 3) / 4, 8u * (uint64_t)c->wide_cov_blocks);
        hipLaunchKernelGGL(bazwide::cov_wide_pairs_kernel, dim3(blocks), dim3(256), 0, c->stream,
                           reinterpret_cast<const float2*>(d_in), dR, nb, c->m, c->K, npairs);
        HIP_TRY(c, hipGetLastError());
        return BAZ_MUSIC_OK;
    }
    if (c->wide_cov_mfma) {         // 17 <= m <= 32: one wave per item on the fp64 matrix core, 2 workgroups per CU at most
        const uint32_t blocks = std::min<uint32_t>((nb + 3) / 4, c->wide_cov_blocks);
        hipLaunchKernelGGL(bazwide::cov_wide_mfma_kernel, dim3(blocks), dim3(256), 0, c->stream,
                           reinterpret_cast<const float2*>(d_in), dR, nb, c->m, c->K);
        HIP_TRY(c, hipGetLastError());
        return BAZ_MUSIC_OK;
    }
    hipLaunchKernelGGL(bazwide::cov_wide_kernel, dim3(nb), dim3(bazwide::WB), (size_t)bazwide::COV_TC * c->m * sizeof(float2),
                       c->stream, reinterpret_cast<const float2*>(d_in), dR, c->m, c->K);
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

int check_launch_pointers(baz_music_ctx* c);       // (below, next to process_device_locked)
int refuse_null(baz_music_ctx* c, const void* p, const char* name);

int process_wide_locked(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_ang, void* d_lvl, void* d_spec)
{
    const uint32_t pass = std::min(batch, wide_pass_items(c));
    int r = ensure_wide_workspace(c, pass);
    if (r) return r;
    r = check_launch_pointers(c);
    if (r) return r;
    const float* in = static_cast<const float*>(d_in);
    float* ang = static_cast<float*>(d_ang);
    float* lvl = static_cast<float*>(d_lvl);
    float* spec = static_cast<float*>(d_spec);
    const uint32_t nn = c->m - c->n;
    for (uint32_t off = 0; off < batch; off += pass) {
        const uint32_t nb = std::min(pass, batch - off);
        r = launch_cov_wide(c, in + (size_t)off * c->nsamples * 2, nb, c->dR);
        if (r) return r;
        {
            ProfScope ps(c, BAZ_MUSIC_STAGE_EVD);
            const uint8_t* only = nullptr;
            if (c->sub_evd && c->n <= 8) {     // few emitters: signal subspace by orthogonal iteration, Jacobi for what it hands back
                // from 50 antennas on the lower triangle of R only (sub_wide_kernel, TRI): four workgroups per CU instead of two or three
                const bool tri = c->m >= 50;
                const size_t lds = ((tri ? (size_t)c->m * (c->m + 1) / 2 : (size_t)c->m * (c->m + 1)) + (size_t)c->n * 64) * sizeof(double2);
#define BAZ_SUB_WIDE(P)                                                                                                          \
    do {                                                                                                                         \
        if (tri) hipLaunchKernelGGL((bazwide::sub_wide_kernel<P, true>), dim3(nb), dim3(64), lds, c->stream, c->dR, c->dGw, c->dSw, c->dRedo, c->m);  \
        else hipLaunchKernelGGL((bazwide::sub_wide_kernel<P, false>), dim3(nb), dim3(64), lds, c->stream, c->dR, c->dGw, c->dSw, c->dRedo, c->m);     \
    } while (0)
                switch (c->n) {
                    case 1: BAZ_SUB_WIDE(1); break;
                    case 2: BAZ_SUB_WIDE(2); break;
                    case 3: BAZ_SUB_WIDE(3); break;
                    case 4: BAZ_SUB_WIDE(4); break;
                    case 5: BAZ_SUB_WIDE(5); break;
                    case 6: BAZ_SUB_WIDE(6); break;
                    case 7: BAZ_SUB_WIDE(7); break;
                    default: BAZ_SUB_WIDE(8); break;
                }
#undef BAZ_SUB_WIDE
                HIP_TRY(c, hipGetLastError());
                only = c->dRedo;
            }
            hipLaunchKernelGGL(bazwide::evd_wide_kernel, dim3(nb), dim3(bazwide::WB), wide_evd_lds(c->m), c->stream, c->dR,
                               c->dGw, c->dSw, c->m, c->n, only);
            HIP_TRY(c, hipGetLastError());
        }
        if (c->wide_mfma && !c->wide_literal_only) {
            // 17 <= m <= 64, n <= 8: the short form on the fp64 matrix core, candidates per bin range, bazmusic's merge
            const WideScanGeom W = wide_scan_geometry(c, nb);
            const uint32_t nk = W.nk, groups = W.groups, nsplit = W.nsplit;
            r = ensure_candidates(c, std::max(wide_cand_entries(c, pass), (size_t)nb * nsplit * nk));
            if (r) return r;
            r = refuse_null(c, c->dCand, "dCand");
            if (r) return r;
            float* sp = spec ? spec + (size_t)off * c->res : nullptr;
            {
                ProfScope ps(c, BAZ_MUSIC_STAGE_SCAN);
                const bool vec4 = sp && (c->res % 4u) == 0 && (reinterpret_cast<uintptr_t>(sp) % 16u) == 0;
#define BAZ_WIDE_ARGS dim3(groups * nsplit), dim3(256), 0, c->stream, c->dSw, c->dGw, c->dTB + c->tb_step_elems, c->dA2p + 64, c->dTA, sp, \
                      c->dCand, nb, c->m, c->n, c->res, nsplit, c->keep_mask, c->refine_below, c->dRefined + c->stat_parity
#define BAZ_WIDE_LAUNCH(PMAX, NOUT)                                                                                       \
    do {                                                                                                                  \
        if (sp && vec4) hipLaunchKernelGGL((bazwide::scan_wide_mfma_kernel<true, true, PMAX, NOUT>), BAZ_WIDE_ARGS);      \
        else if (sp) hipLaunchKernelGGL((bazwide::scan_wide_mfma_kernel<true, false, PMAX, NOUT>), BAZ_WIDE_ARGS);        \
        else hipLaunchKernelGGL((bazwide::scan_wide_mfma_kernel<false, false, PMAX, NOUT>), BAZ_WIDE_ARGS);               \
    } while (0)
                // PMAX: staged phases per step (2 to 32 antennas, 4 to 64); NOUT: outputs per item (4: n <= 2, 8: n = 3, 4)
                if (c->m <= 32 && c->n <= 2) BAZ_WIDE_LAUNCH(2, 4);
                else if (c->m <= 32 && c->n <= 4) BAZ_WIDE_LAUNCH(2, 8);
                else if (c->m <= 32) BAZ_WIDE_LAUNCH(2, 16);
                else if (c->n <= 2) BAZ_WIDE_LAUNCH(4, 4);
                else if (c->n <= 4) BAZ_WIDE_LAUNCH(4, 8);
                else BAZ_WIDE_LAUNCH(4, 16);
#undef BAZ_WIDE_LAUNCH
#undef BAZ_WIDE_ARGS
                HIP_TRY(c, hipGetLastError());
            }
            c->last_nsplit = nsplit;
            {
                ProfScope ps(c, BAZ_MUSIC_STAGE_MERGE);
                // the statistic counter of the NEXT call is cleared by the last pass's merge only (the passes of one call add up)
                unsigned long long* next_stat = (off + nb >= batch) ? c->dRefined + (c->stat_parity ^ 1) : nullptr;
#define BAZ_WIDE_MERGE(NK)                                                                                               \
    hipLaunchKernelGGL((topn_merge_kernel<NK>), dim3((nb + 255) / 256), dim3(256), 0, c->stream, c->dCand, sp,          \
                       ang + (size_t)off * c->n, lvl ? lvl + (size_t)off * c->n : nullptr, nb, c->res, c->n, nsplit,    \
                       c->keep_mask, next_stat)
                if (nk == 2) BAZ_WIDE_MERGE(2);
                else if (nk == 4) BAZ_WIDE_MERGE(4);
                else BAZ_WIDE_MERGE(8);
#undef BAZ_WIDE_MERGE
                HIP_TRY(c, hipGetLastError());
            }
            continue;
        }
        {
            ProfScope ps(c, BAZ_MUSIC_STAGE_SCAN);
            // enough workgroups for a small batch: split the bins of an item over several
            const uint32_t max_y = (c->res + bazwide::WB - 1) / bazwide::WB;
            const uint32_t ny = std::max(1u, std::min(max_y, (2048u + nb - 1) / nb));
            const uint32_t bpb = round_up((c->res + ny - 1) / ny, bazwide::WB);
            const bool short_form = 2 * c->n <= c->m && !c->wide_literal_only;     // few emitters: ||a||^2 - ||S^H a||^2 away from the nulls
            hipLaunchKernelGGL(bazwide::scan_wide_kernel, dim3(nb, (c->res + bpb - 1) / bpb), dim3(bazwide::WB),
                               (size_t)(nn + (short_form ? c->n : 0u)) * c->m * sizeof(double2), c->stream, c->dGw,
                               short_form ? c->dSw : nullptr, c->dTA, c->dA2, c->refine_below, c->dWS,
                               spec ? spec + (size_t)off * c->res : nullptr, c->m, c->n, c->res, bpb);
            HIP_TRY(c, hipGetLastError());
        }
        {
            ProfScope ps(c, BAZ_MUSIC_STAGE_MERGE);
            hipLaunchKernelGGL(bazwide::topn_wide_kernel, dim3(nb), dim3(bazwide::WB), 0, c->stream, c->dWS,
                               ang + (size_t)off * c->n, lvl ? lvl + (size_t)off * c->n : nullptr, c->res, c->n);
            HIP_TRY(c, hipGetLastError());
        }
    }
    return BAZ_MUSIC_OK;
}

// ---- the table's device images: allocation, device-side construction, exchange -----------------------------------
using TableSet = baz_music_ctx::TableSet;

size_t coarse_image_bytes(const baz_music_ctx* c)
{
    return ((size_t)(c->cs_tiles + 1) * cs_c_units((int)c->m) + (size_t)c->cs_tiles * CS_X_UNITS * cs_groups((int)c->m)) * 16;
}
bool wants_i8_image(const baz_music_ctx* c)
{
    return !c->wide && c->m >= 6 && c->n <= 4 && i8_image_bytes(c->m, c->fb_steps) <= I8_IMAGE_LIMIT;
}

// one set of buffers for the configuration of `c` (what baz_music_create used to allocate in place)
int alloc_table_set(baz_music_ctx* c, TableSet& T)
{
    const size_t pad_steps = (size_t)c->fb_steps + 2;
    if (c->wide) {
        if (dev_malloc((void**)&T.dTA, (size_t)c->m * c->res * sizeof(float2)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
        if (dev_malloc((void**)&T.dA2, (size_t)c->res * sizeof(double)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
        if (c->wide_mfma) {
            if (dev_malloc((void**)&T.dTB, pad_steps * c->tb_step_elems * sizeof(double2)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
            if (dev_malloc((void**)&T.dA2p, pad_steps * 64 * sizeof(double)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
        }
        return BAZ_MUSIC_OK;
    }
    if (dev_malloc((void**)&T.dFB, pad_steps * c->fb_step_elems * sizeof(double2)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (dev_malloc((void**)&T.dTB, pad_steps * c->tb_step_elems * sizeof(double2)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (c->m <= 8 && dev_malloc((void**)&T.dCS, coarse_image_bytes(c)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
#ifdef BAZ_MUSIC_LAB
    if (c->m <= 8 && c->sort_mode != 0 &&            // (lab: the sort key's table)
        dev_malloc((void**)&T.dKT, (size_t)bazsort::key_samples(c->res) * c->m * c->m * sizeof(float)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (c->i8p_on && c->m <= 4 && c->n <= 4 && dev_malloc((void**)&T.dIP, i8p_image_bytes(c->fb_steps)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
#endif
    if (wants_i8_image(c) && dev_malloc((void**)&T.dIB, i8_image_bytes(c->m, c->fb_steps)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (short_form_applies(c->m, c->n) && dev_malloc((void**)&T.dA2p, pad_steps * 64 * sizeof(double)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    return BAZ_MUSIC_OK;
}

void free_table_set(TableSet& T)
{
    if (T.dFB) (void)dev_free(T.dFB);
    if (T.dTB) (void)dev_free(T.dTB);
    if (T.dCS) (void)dev_free(T.dCS);
    if (T.dIB) (void)dev_free(T.dIB);
#ifdef BAZ_MUSIC_LAB
    if (T.dIP) (void)dev_free(T.dIP);
    if (T.dKT) (void)dev_free(T.dKT);
#endif
    if (T.dA2p) (void)dev_free(T.dA2p);
    if (T.dTA) (void)dev_free(T.dTA);
    if (T.dA2) (void)dev_free(T.dA2);
    T = TableSet();
}

// the set the kernels are launched with lives in the context's own fields (every launch site reads c->dFB, c->cs, ...)
TableSet active_table_set(const baz_music_ctx* c)
{
    TableSet T;
    T.dFB = c->dFB; T.dTB = c->dTB; T.dCS = c->dCS; T.dIB = c->dIB; T.dA2p = c->dA2p; T.dTA = c->dTA; T.dA2 = c->dA2;
#ifdef BAZ_MUSIC_LAB
    T.dIP = c->dIP;
    T.dKT = c->dKT;
#endif
    T.cs = c->cs; T.i8 = c->i8; T.cs_ok = c->cs_ok; T.i8_ok = c->i8_ok; T.refine_below = c->refine_below;
    return T;
}
void install_table_set(baz_music_ctx* c, const TableSet& T)
{
    c->dFB = T.dFB; c->dTB = T.dTB; c->dCS = T.dCS; c->dIB = T.dIB; c->dA2p = T.dA2p; c->dTA = T.dTA; c->dA2 = T.dA2;
#ifdef BAZ_MUSIC_LAB
    c->dIP = T.dIP;
    c->dKT = T.dKT;
#endif
    c->cs = T.cs; c->i8 = T.i8; c->cs_ok = T.cs_ok; c->i8_ok = T.i8_ok; c->refine_below = T.refine_below;
}

inline dim3 grid_for(size_t threads) { return dim3((unsigned)((threads + 255) / 256)); }

// Waits for a short piece of work on `s` by polling: a blocking hipStreamSynchronize sleeps on an interrupt after a brief spin, and on this stack
// such a sleep now and then lasts 5 - 11 ms when another thread of the process is waiting for its own stream at the same time (1 retune in ~100
// at config 3).  The table builders take 0.1 - 0.3 ms: poll for up to 3 ms, then fall back to the blocking wait.
hipError_t wait_stream_polling(hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) {
            (void)hipGetLastError();
            return hipStreamSynchronize(s);
        }
        std::this_thread::yield();
    }
}

// Builds every image of the table in c->hRaw into `T` on c->s_tab and WAITS for them (the caller holds c->tab_mtx, not c->mtx):
// raw table H2D, the three scalars (table_stats_kernel, one small D2H), the parameters on the host, the image kernels.
int build_tables_device(baz_music_ctx* c, TableSet& T)
{
    hipStream_t s = c->s_tab;
    const uint32_t m = c->m, res = c->res, steps = c->fb_steps;
    const size_t raw_bytes = (size_t)res * m * 2 * sizeof(float);
    // (no copy engines on this path: copy_words_kernel reads the page-locked staging copy over the link, see table_kernels.hip.h)
    void *zRaw = nullptr, *zStats = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer(&zRaw, c->hRaw, 0));
    HIP_TRY(c, hipHostGetDevicePointer(&zStats, c->hTabStats, 0));
    hipLaunchKernelGGL(baztab::copy_words_kernel, dim3((unsigned)std::min<size_t>(512, (raw_bytes / 8 + 255) / 256)), dim3(256), 0, s,
                       static_cast<const unsigned long long*>(zRaw), reinterpret_cast<unsigned long long*>(c->dRaw), raw_bytes / 8);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemsetAsync(c->dTabStats, 0, sizeof(baztab::TableStats), s));
    hipLaunchKernelGGL(baztab::table_stats_kernel, grid_for(res), dim3(256), 0, s, c->dRaw, m, res, c->wide ? 0 : 1,
                       static_cast<baztab::TableStats*>(c->dTabStats));
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(baztab::copy_words_kernel, dim3(1), dim3(256), 0, s, static_cast<const unsigned long long*>(c->dTabStats),
                       static_cast<unsigned long long*>(zStats), sizeof(baztab::TableStats) / 8);
    HIP_TRY(c, hipGetLastError());
    // (meanwhile: the images that need no scalar)
    const size_t pad_steps = (size_t)steps + 2;
    if (T.dFB) {
        hipLaunchKernelGGL(baztab::build_fb_kernel, grid_for(pad_steps * c->fb_step_elems), dim3(256), 0, s, c->dRaw, m, res, steps, T.dFB);
        HIP_TRY(c, hipGetLastError());
    }
    if (T.dTB) {
        hipLaunchKernelGGL(baztab::build_tb_kernel, grid_for(pad_steps * c->tb_step_elems), dim3(256), 0, s, c->dRaw, m, res, steps, T.dTB);
        HIP_TRY(c, hipGetLastError());
    }
    if (T.dA2p) {   // ||a||^2 per bin, huge outside the table like FB's diagonal there
        hipLaunchKernelGGL(baztab::build_a2_kernel, grid_for(pad_steps * 64), dim3(256), 0, s, c->dRaw, m, res, (uint32_t)(pad_steps * 64), 64u, T.dA2p);
        HIP_TRY(c, hipGetLastError());
    }
    if (T.dA2) {
        hipLaunchKernelGGL(baztab::build_a2_kernel, grid_for(res), dim3(256), 0, s, c->dRaw, m, res, res, 0u, T.dA2);
        HIP_TRY(c, hipGetLastError());
    }
    if (T.dTA) {
        hipLaunchKernelGGL(baztab::build_ta_kernel, grid_for((size_t)m * res), dim3(256), 0, s, reinterpret_cast<const float2*>(c->dRaw), m, res, T.dTA);
        HIP_TRY(c, hipGetLastError());
    }
#ifdef BAZ_MUSIC_LAB
    if (T.dKT) {
        const uint32_t ns = bazsort::key_samples(res);
        hipLaunchKernelGGL(bazsort::build_key_table_kernel, grid_for((size_t)ns * m * m), dim3(256), 0, s, c->dRaw, m, res, ns, T.dKT);
        HIP_TRY(c, hipGetLastError());
    }
#endif
    if (T.dCS) HIP_TRY(c, hipMemsetAsync(T.dCS, 0, coarse_image_bytes(c), s));
    if (T.dIB) HIP_TRY(c, hipMemsetAsync(T.dIB, 0, i8_image_bytes(m, steps), s));
    HIP_TRY(c, wait_stream_polling(s));                  // the scalars are on the host
    const baztab::TableStats st = *static_cast<const baztab::TableStats*>(c->hTabStats);
    double fmax, amax2;
    std::memcpy(&fmax, &st.fmax_bits, sizeof(double));
    std::memcpy(&amax2, &st.amax2_bits, sizeof(double));
    // projector-form accuracy: |error(d)| ~ m^2 eps ||a||^2 absolute; a d at or below m 1e-8 max||a||^2 (relative
    // error there <~ 1e-7) is recomputed in the reference's literal form (literal_tile() in the scan)
    T.refine_below = amax2 * (double)m * 1e-8;
    T.cs_ok = false;
    T.i8_ok = false;
    const bool finite = st.nonfinite == 0;
    if (T.dCS && finite) {    // coarse-gated scan: f16 pieces of the scaled table + the fp64 operand, per 16-bin tile
        double FS = 0.0;
        if (coarse_params_from(fmax, m, T.cs, FS)) {
            T.cs.lazy = 1;
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COARSE_LAZY")) T.cs.lazy = atoi(v);            // lab
            uint8_t* img = reinterpret_cast<uint8_t*>(T.dCS);
            hipLaunchKernelGGL(baztab::build_coarse_c_kernel, grid_for((size_t)(c->cs_tiles + 1) * 16 * m * m), dim3(256), 0, s,
                               c->dRaw, m, res, c->cs_tiles, FS, img);
            HIP_TRY(c, hipGetLastError());
            double* X0 = reinterpret_cast<double*>(img + (size_t)(c->cs_tiles + 1) * cs_c_units((int)m) * 16);
            hipLaunchKernelGGL(baztab::build_coarse_x_kernel, grid_for((size_t)c->cs_tiles * 256 * cs_groups((int)m)), dim3(256), 0, s,
                               c->dRaw, m, res, c->cs_tiles, X0);
            HIP_TRY(c, hipGetLastError());
            T.cs_ok = true;
        }
    }
#ifdef BAZ_MUSIC_LAB
    if (T.dIP) HIP_TRY(c, hipMemsetAsync(T.dIP, 0, i8p_image_bytes(steps), s));
    if (T.dIP && finite) {    // 2 .. 4 antennas: level-packed digit operands
        double sf = 0.0;
        if (i8_params_from(fmax, m, T.i8, sf)) {
            hipLaunchKernelGGL(baztab::build_i8p_kernel, grid_for(res), dim3(256), 0, s, c->dRaw, m, res, steps, sf, T.dIP);
            HIP_TRY(c, hipGetLastError());
            T.i8_ok = true;
        }
    }
#endif
    if (T.dIB && finite) {    // int8-matrix-core scan: the table's digit image
        double sf = 0.0;
        if (i8_params_from(fmax, m, T.i8, sf)) {
            uint8_t* img = reinterpret_cast<uint8_t*>(T.dIB);
            hipLaunchKernelGGL(baztab::build_i8_kernel, grid_for((size_t)res * i8_nkb((int)m) * 4), dim3(256), 0, s, c->dRaw, m, res,
                               steps, sf, img, img + i8_image_bytes5(m, steps));
            HIP_TRY(c, hipGetLastError());
            T.i8_ok = true;
        }
    }
    HIP_TRY(c, wait_stream_polling(s));
    return BAZ_MUSIC_OK;
}

// Replaces set_array_response's body (.cc:60-70).  The caller holds c->tab_mtx.  `first`: called from baz_music_create (no
// batch can be in flight, the images go straight into the active set).
int retune(baz_music_ctx* c, const float* table_ri, bool first)
{
    const auto t0 = std::chrono::steady_clock::now();
    std::memcpy(c->hRaw, table_ri, (size_t)c->res * c->m * 2 * sizeof(float));
    if (first) {
        TableSet T = active_table_set(c);
        const int r = build_tables_device(c, T);
        install_table_set(c, T);
        return r;
    }
    // the shadow set was the active one until the previous swap: batches launched before that swap may still read it
    if (c->swap_recorded) HIP_TRY(c, hipStreamWaitEvent(c->s_tab, c->ev_swap, 0));
    const int r = build_tables_device(c, c->shadow);
    if (r != BAZ_MUSIC_OK) return r;                      // the old table stays in force
    const auto t1 = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> lk(c->mtx);           // .cc:67 -- for the exchange of a few pointers only
        const TableSet old = active_table_set(c);
        install_table_set(c, c->shadow);
        c->shadow = old;
        c->swap_recorded = hipEventRecord(c->ev_swap, c->stream) == hipSuccess;
        if (!c->swap_recorded) {                          // cannot order the next retune behind the batches in flight: drain them now
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->stream);
        }
    }
    const auto t2 = std::chrono::steady_clock::now();
    c->last_retune_ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
    c->last_swap_wait_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return BAZ_MUSIC_OK;
}

void free_slots(baz_music_ctx* c)
{
    for (auto& sl : c->slot) {
        if (sl.in) (void)dev_free(sl.in);
        if (sl.al) (void)dev_free(sl.al);
        if (sl.h_al) (void)hipHostFree(sl.h_al);
        if (sl.spec) (void)dev_free(sl.spec);
        if (sl.h2d) (void)hipEventDestroy(sl.h2d);
        if (sl.comp) (void)hipEventDestroy(sl.comp);
        if (sl.d2h) (void)hipEventDestroy(sl.d2h);
        sl = baz_music_ctx::Slot();
    }
    c->s_cap = 0;
    c->s_has_spec = false;
}

int ensure_slots(baz_music_ctx* c, uint32_t chunk, bool want_spec)
{
    if (chunk <= c->s_cap && (!want_spec || c->s_has_spec)) return BAZ_MUSIC_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const bool spec = want_spec || c->s_has_spec;
    const uint32_t cap = std::max(chunk, c->s_cap);
    free_slots(c);
    if (!c->s_h2d) HIP_TRY(c, hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
    if (!c->s_d2h) HIP_TRY(c, hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking));
    for (auto& sl : c->slot) {
        HIP_TRY(c, dev_malloc((void**)&sl.in, (size_t)cap * c->nsamples * 8));
        HIP_TRY(c, dev_malloc((void**)&sl.al, (size_t)cap * c->n * 8));
        HIP_TRY(c, hipHostMalloc((void**)&sl.h_al, (size_t)cap * c->n * 8, hipHostMallocDefault));
        if (spec) HIP_TRY(c, dev_malloc((void**)&sl.spec, (size_t)cap * c->res * 4));
        HIP_TRY(c, hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&sl.comp, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&sl.d2h, hipEventDisableTiming));
    }
    c->s_cap = cap;
    c->s_has_spec = spec;
    return BAZ_MUSIC_OK;
}

// Is this caller pointer page-locked (hipHostMalloc / hipHostRegister / torch pinned memory)?
bool is_pinned_host(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();     // plain pageable memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Is the WHOLE range [p, p + bytes) page-locked and addressable from the device as one piece?  Both ends must be
// page-locked host memory and their device addresses must be `bytes - 1` apart (one mapping, or mappings that continue
// each other): a range of which only the head lies inside a registration -- two blocks fanned out from one upstream
// buffer, one of them owning the registration -- would otherwise hand the zero-copy kernels a device pointer that runs
// into unmapped host memory (a GPU fault, not an error code).  Such a range takes the pageable path.
bool range_is_pinned(const void* p, size_t bytes)
{
    if (!p || !bytes) return false;
    const char* last = static_cast<const char*>(p) + bytes - 1;
    if (!is_pinned_host(p) || !is_pinned_host(last)) return false;
    void *d0 = nullptr, *d1 = nullptr;
    if (hipHostGetDevicePointer(&d0, const_cast<void*>(p), 0) != hipSuccess ||
        hipHostGetDevicePointer(&d1, const_cast<char*>(last), 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return static_cast<char*>(d1) - static_cast<char*>(d0) == (ptrdiff_t)(bytes - 1);
}

// hipMemcpyAsync between a device buffer and a caller range that may be only PARTLY inside someone's page-lock
// registration (measured: the runtime answers such a copy with "invalid argument", also when the range straddles two
// registrations).  The whole range is tried first -- the only call the ordinary cases ever make --; a refused range
// is halved at a page boundary of the host address until the pieces are homogeneous, and a piece of at most one page
// that is still refused (a registration boundary inside it: registrations are byte-exact) crosses through a bounce
// buffer with a synchronize.  Only the refused (rare) case costs anything.
hipError_t copy_host_range(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, stream);
    if (e != hipErrorInvalidValue || bytes == 0) return e;
    (void)hipGetLastError();
    const bool h2d = kind == hipMemcpyHostToDevice;
    const uintptr_t host = (uintptr_t)(h2d ? src : dst);
    if (bytes > 4096) {
        uintptr_t mid = (host + bytes / 2) & ~(uintptr_t)4095;
        if (mid <= host || mid >= host + bytes) mid = host + bytes / 2;
        const size_t first = mid - host;
        e = copy_host_range(dst, src, first, kind, stream);
        if (e != hipSuccess) return e;
        return copy_host_range((char*)dst + first, (const char*)src + first, bytes - first, kind, stream);
    }
    char bounce[4096];
    if (h2d) {
        memcpy(bounce, src, bytes);
        e = hipMemcpyAsync(dst, bounce, bytes, kind, stream);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(stream);      // the bounce buffer dies with this frame
    }
    e = hipMemcpyAsync(bounce, src, bytes, kind, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e == hipSuccess) memcpy(dst, bounce, bytes);
    return e;
}

// Page-lock [p, p + bytes) for this context: see include/baz_music_hip.h.  Caller holds c->mtx.
// The runtime rejects a copy whose host range is only PARTLY inside a registration (hipMemcpyAsync: invalid argument --
// measured, also when the range straddles two registrations).  So (i) registrations are the caller's exact byte
// ranges, never rounded out to pages (a neighbouring heap object must not end up half inside one); (ii) a request that
// overlaps or touches earlier registrations replaces them by ONE registration of the union (a circular stream buffer
// is covered after its first few calls); (iii) a request that cannot be locked as a whole (limit, refusal) leaves
// nothing of itself locked: the registrations it touches are dropped.
//
// Registrations are PROCESS-wide objects with a count of the contexts that use them (g_pins): two blocks reading the
// same stream buffer (a fan-out) share one registration, and it is released when the LAST of them lets go -- a context
// that stops must not unmap memory another context's copies or kernels are still addressing.
struct SharedPin { uintptr_t lo, hi; int refs; };
std::mutex g_pin_mtx;
std::vector<SharedPin> g_pins;
constexpr uint32_t REFUSED_RETRY_AFTER = 1024;    // a refused range is tried again after this many requests for it

void shared_pin_release(uintptr_t lo, uintptr_t hi)         // caller holds g_pin_mtx
{
    for (size_t i = 0; i < g_pins.size(); ++i)
        if (g_pins[i].lo == lo && g_pins[i].hi == hi) {
            if (--g_pins[i].refs <= 0) {
                if (hipHostUnregister((void*)lo) != hipSuccess) (void)hipGetLastError();   // already unmapped
                g_pins.erase(g_pins.begin() + i);
            }
            return;
        }
}

int host_register_locked(baz_music_ctx* c, const void* p, size_t bytes)
{
    if (!p || !bytes) return BAZ_MUSIC_OK;
    uintptr_t lo = (uintptr_t)p, hi = (uintptr_t)p + bytes;
    for (const auto& pin : c->pins)
        if (pin.lo <= lo && hi <= pin.hi) return BAZ_MUSIC_OK;                       // known (the per-call case)
    for (size_t i = 0; i < c->refused.size(); ++i) {
        auto& r = c->refused[i];
        if (r.lo < hi && lo < r.hi) {
            if (++r.asked < REFUSED_RETRY_AFTER) return BAZ_MUSIC_E_HIP;             // refused not long ago
            c->refused.erase(c->refused.begin() + i);                                // (a transient refusal must not last until stop())
            break;
        }
    }
    std::lock_guard<std::mutex> g(g_pin_mtx);
    for (auto& sp : g_pins)
        if (sp.lo <= lo && hi <= sp.hi) {                                            // another context's registration: share it
            if (c->pinned_bytes + (sp.hi - sp.lo) > c->pin_limit) return BAZ_MUSIC_E_UNSUPPORTED;
            ++sp.refs;
            c->pins.push_back({sp.lo, sp.hi, 0});
            c->pinned_bytes += sp.hi - sp.lo;
            return BAZ_MUSIC_OK;
        }
    std::vector<size_t> touch;
    uint64_t held = 0;
    for (size_t i = 0; i < c->pins.size(); ++i)
        if (c->pins[i].hi >= lo && c->pins[i].lo <= hi) {
            touch.push_back(i);
            held += c->pins[i].hi - c->pins[i].lo;
        }
    if (touch.empty() && range_is_pinned(p, bytes))
        return BAZ_MUSIC_OK;                   // its owner's page-locked memory (hipHostMalloc, torch): not ours to manage
    for (size_t i : touch) {
        lo = std::min(lo, c->pins[i].lo);
        hi = std::max(hi, c->pins[i].hi);
    }
    // the union replaces what it touches -- possible only where nobody else holds those registrations, and where no
    // registration of another context lies inside the union (the runtime would refuse the overlap)
    for (const auto& sp : g_pins) {
        if (!(sp.lo < hi && lo < sp.hi)) continue;
        bool ours = false;
        for (size_t i : touch) ours = ours || (c->pins[i].lo == sp.lo && c->pins[i].hi == sp.hi);
        if (!ours || sp.refs > 1) return BAZ_MUSIC_E_UNSUPPORTED;                    // (the copy path splits at the boundaries)
    }
    const bool fits = c->pinned_bytes - held + (hi - lo) <= c->pin_limit;
    for (size_t k = touch.size(); k-- > 0;) {                                         // back to front: indices stay valid
        shared_pin_release(c->pins[touch[k]].lo, c->pins[touch[k]].hi);
        c->pins.erase(c->pins.begin() + touch[k]);
    }
    c->pinned_bytes -= held;
    if (!fits) return BAZ_MUSIC_E_UNSUPPORTED;
    const hipError_t e = hipHostRegister((void*)lo, hi - lo, hipHostRegisterDefault);
    if (e == hipSuccess) {
        g_pins.push_back({lo, hi, 1});
        c->pins.push_back({lo, hi, 0});
        c->pinned_bytes += hi - lo;
        return BAZ_MUSIC_OK;
    }
    c->refused.push_back({lo, hi, 0});
    return hip_fail(c, e, "hipHostRegister");
}

void host_unregister_all_locked(baz_music_ctx* c)
{
    {
        std::lock_guard<std::mutex> g(g_pin_mtx);
        for (const auto& pin : c->pins) shared_pin_release(pin.lo, pin.hi);
    }
    c->pins.clear();
    c->refused.clear();
    c->pinned_bytes = 0;
}

// One pass of the hot path over `batch` device-resident items: covariance, EVD, scan (with the literal-form
// refinement of near-null tiles inside) and the tiny top-n merge, back to back on the context's stream.  (Cutting the batch into sub-batches and overlapping covariance/EVD of
// sub-batch i+1 with the scan of sub-batch i on two extra streams was measured and is SLOWER -- 0.53 ms ->
// 0.60 / 0.70 / 1.05 ms at 2 / 4 / 8 sub-batches, profiles/r01_two_stream_pipeline_negative.txt: the
// cross-stream event dependencies cost more than the overlap buys.)
// Start the statistic of one API call: flip to the counter the previous call's merge kernel cleared (a memset only
// after a call that failed before its merge).
int begin_statistic(baz_music_ctx* c)
{
    c->stat_parity ^= 1;
    if (!c->stat_next_clean)
        HIP_TRY(c, hipMemsetAsync(c->dRefined + c->stat_parity, 0, sizeof(unsigned long long), c->stream));
    c->stat_next_clean = false;
    return BAZ_MUSIC_OK;
}

// A launch sequence is REFUSED (BAZ_MUSIC_E_HIP, message in baz_music_last_hip_error) when a device pointer it is about to hand to a
// kernel -- or to offset -- is null: a bug of that kind must come back as an error code, never as a GPU memory fault that takes the
// process down (BENCH_r05: "Memory access fault ... on address 0x10000").  Host-side, a dozen compares per call.
int refuse_null(baz_music_ctx* c, const void* p, const char* name)
{
    if (p) return BAZ_MUSIC_OK;
    snprintf(c->hip_err, sizeof(c->hip_err), "internal: device pointer %s is null at launch (m=%u n=%u res=%u): launch refused", name, c->m, c->n, c->res);
    return BAZ_MUSIC_E_HIP;
}
#define BAZ_REQUIRE(c, ptr)                                  \
    do {                                                     \
        if (const int rn__ = refuse_null((c), (c)->ptr, #ptr)) return rn__; \
    } while (0)

int check_launch_pointers(baz_music_ctx* c)
{
    BAZ_REQUIRE(c, dRefined);
    BAZ_REQUIRE(c, dR);
    BAZ_REQUIRE(c, dRedo);
    if (c->wide) {
        BAZ_REQUIRE(c, dGw); BAZ_REQUIRE(c, dWS); BAZ_REQUIRE(c, dSw); BAZ_REQUIRE(c, dTA); BAZ_REQUIRE(c, dA2);
        if (c->wide_mfma) { BAZ_REQUIRE(c, dTB); BAZ_REQUIRE(c, dA2p); }
        return BAZ_MUSIC_OK;
    }
    BAZ_REQUIRE(c, dQ); BAZ_REQUIRE(c, dG); BAZ_REQUIRE(c, dCand); BAZ_REQUIRE(c, dFB); BAZ_REQUIRE(c, dTB);
    if (short_form_applies(c->m, c->n)) { BAZ_REQUIRE(c, dSs); BAZ_REQUIRE(c, dA2p); }
    if (c->cs_ok) BAZ_REQUIRE(c, dCS);
    if (c->i8_ok && wants_i8_image(c)) BAZ_REQUIRE(c, dIB);
    return BAZ_MUSIC_OK;
}

int process_device_locked(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_ang, void* d_lvl,
                          void* d_spec)
{
    if (c->wide) {
        const int rw = process_wide_locked(c, d_in, batch, d_ang, d_lvl, d_spec);
        if (rw == BAZ_MUSIC_OK) c->stat_next_clean = true;    // (no scan statistic on this path: the counters stay 0)
        return rw;
    }
    int r = ensure_workspace(c, batch);
    if (r) return r;
    r = reserve_candidates(c, batch);
    if (r) return r;
    r = check_launch_pointers(c);
    if (r) return r;
    const uint32_t qstride = baz_music_q_stride(batch);
    if (c->fused_covevd) {
        r = launch_covevd(c, static_cast<const float*>(d_in), batch, c->dQ, qstride, c->dG);
        if (r) return r;
    } else {
        r = launch_cov(c, static_cast<const float*>(d_in), batch, c->dR);
        if (r) return r;
        r = launch_evd(c, c->dR, batch, c->dQ, qstride, c->dG);
        if (r) return r;
    }
    float* spec = static_cast<float*>(d_spec);
    if (c->peak_mode && !spec) {   // the peak picker reads the spectrum: keep a private one when port 2 is not wired
        const size_t need = (size_t)batch * c->res;
        if (need > c->peak_spec_cap) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->dPeakSpec) (void)dev_free(c->dPeakSpec);
            c->dPeakSpec = nullptr; c->peak_spec_cap = 0;
            HIP_TRY(c, dev_malloc((void**)&c->dPeakSpec, need * sizeof(float)));
            c->peak_spec_cap = need;
        }
        spec = c->dPeakSpec;
    }
    r = launch_scan(c, c->dQ, qstride, batch, static_cast<float*>(d_ang), static_cast<float*>(d_lvl), spec);
    if (r) return r;
    r = launch_merge(c, batch, static_cast<float*>(d_ang), static_cast<float*>(d_lvl), spec);
    if (r) return r;
    c->stat_next_clean = true;     // the merge cleared the next call's statistic counter
    if (c->peak_mode) return launch_peaks(c, batch, static_cast<float*>(d_ang), static_cast<float*>(d_lvl), spec);
    return BAZ_MUSIC_OK;
}

int ensure_h_al_big(baz_music_ctx* c, size_t floats)
{
    if (floats <= c->h_al_big_cap) return BAZ_MUSIC_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_al_big) (void)hipHostFree(c->h_al_big);
    c->h_al_big = nullptr; c->h_al_big_cap = 0;
    HIP_TRY(c, hipHostMalloc((void**)&c->h_al_big, floats * sizeof(float), hipHostMallocDefault));
    c->h_al_big_cap = floats;
    return BAZ_MUSIC_OK;
}

// baz_music_create's last step: both table sets, the builders' side stream and staging buffers, then the first table
int create_tables(baz_music_ctx* c, const float* table_ri)
{
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
    int prio = hi;
    if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_TAB_PRIORITY")) prio = atoi(v) > 0 ? hi : (atoi(v) < 0 ? lo : 0);   // lab: 1 highest, 0 default, -1 lowest
    if (hipStreamCreateWithPriority(&c->s_tab, hipStreamNonBlocking, prio) != hipSuccess) return BAZ_MUSIC_E_HIP;
    if (hipEventCreateWithFlags(&c->ev_swap, hipEventDisableTiming) != hipSuccess) return BAZ_MUSIC_E_HIP;
    const size_t raw_bytes = (size_t)c->res * c->m * 2 * sizeof(float);
    if (dev_malloc((void**)&c->dRaw, raw_bytes) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (hipHostMalloc((void**)&c->hRaw, raw_bytes, hipHostMallocDefault) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (dev_malloc(&c->dTabStats, sizeof(baztab::TableStats)) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    if (hipHostMalloc(&c->hTabStats, sizeof(baztab::TableStats), hipHostMallocDefault) != hipSuccess) return BAZ_MUSIC_E_NOMEM;
    TableSet first;
    int r = alloc_table_set(c, first);
    install_table_set(c, first);                          // (also after a failed allocation: destroy frees what exists)
    if (r == BAZ_MUSIC_OK) r = alloc_table_set(c, c->shadow);
    if (r != BAZ_MUSIC_OK) return r;
    std::lock_guard<std::mutex> tl(c->tab_mtx);
    r = retune(c, table_ri, true);
    if (r != BAZ_MUSIC_OK) return r;
    // first touch of the SHADOW set here, not in the first retune (fresh device allocations are mapped on first use: one retune in twenty
    // took 7 - 12 ms and held the submitting thread up by 0.9 ms, tests/test_retune.py): build the same table into it once
    r = build_tables_device(c, c->shadow);
    return r;
}

}  // namespace

extern "C" {

int baz_music_create(baz_music_ctx** out, uint32_t m, uint32_t n, uint32_t nsamples, uint32_t resolution,
                     const float* table_ri, int device_id)
{
    if (!out) return BAZ_MUSIC_E_INVALID;
    *out = nullptr;
    // lib/baz_music_doa.cc:45-50 (asserts, compiled out in Release) made real; n == m underflows .cc:93
    if (m == 0 || n == 0 || n >= m || nsamples == 0 || (nsamples % m) != 0 || resolution == 0 || !table_ri)
        return BAZ_MUSIC_E_INVALID;
    if (m > BAZ_MUSIC_MAX_M || n > BAZ_MUSIC_MAX_N) return BAZ_MUSIC_E_UNSUPPORTED;   // see baz_music_strerror()
    const bool wide = m > BAZ_MUSIC_FAST_M;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BAZ_MUSIC_E_NODEVICE;
    int dev = device_id;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return BAZ_MUSIC_E_NODEVICE;
    }
    if (dev >= ndev) return BAZ_MUSIC_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return BAZ_MUSIC_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BAZ_MUSIC_E_NODEVICE;   // kernels are gfx950-only

    baz_music_ctx* c = new (std::nothrow) baz_music_ctx;
    if (!c) return BAZ_MUSIC_E_NOMEM;
    c->m = m; c->n = n; c->nsamples = nsamples; c->res = resolution; c->K = nsamples / m;
    c->device = dev;
    if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SCAN_VARIANT")) c->lab_variant = atoi(v);
    if (const char* v = getenv("BAZ_MUSIC_CHUNK_MIB")) c->chunk_bytes = (size_t)std::max(1, std::min(1024, atoi(v))) << 20;
    if (const char* v = getenv("BAZ_MUSIC_PIN_LIMIT_MIB")) c->pin_limit = (uint64_t)std::max(0, atoi(v)) << 20;
    if (const char* v = getenv("BAZ_MUSIC_ZERO_COPY")) c->zero_copy = atoi(v) != 0;
    if (const char* v = getenv("BAZ_MUSIC_SINGLE_MIB")) c->single_limit_mib = std::max(1, std::min(1024, atoi(v)));
    DeviceGuard guard(dev);
    int r = BAZ_MUSIC_OK;
    do {
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { r = BAZ_MUSIC_E_HIP; break; }
        c->stream = c->own_stream;
        // (every fill of a device buffer below is a hipMemsetAsync ON THIS STREAM: hipMemset of device memory may return before the fill has run, and
        // the null stream orders nothing against a non-blocking stream -- round 6, found when the guard allocator's own prefill raced with a scan)
        c->wide = wide;
        if (wide) {   // run-time-m kernels: only the transposed table and the statistic counters (which stay 0)
            if (resolution > (1u << 20)) { r = BAZ_MUSIC_E_UNSUPPORTED; break; }
            {   // dynamic LDS beyond 64 KiB needs the attribute; it is per FUNCTION (shared by every context of the process),
                // so it is always set for the largest m, never for this context's
                constexpr uint32_t MX = BAZ_MUSIC_MAX_M;
                const int evd_lds = (int)wide_evd_lds(MX), scan_lds = (int)((size_t)MX * MX * sizeof(double2));
                const int sub_lds = (int)(((size_t)MX * (MX + 1) + 8u * 64u) * sizeof(double2));
                const void* fn[10] = {reinterpret_cast<const void*>(bazwide::evd_wide_kernel),
                                      reinterpret_cast<const void*>(bazwide::scan_wide_kernel),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<1, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<2, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<3, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<4, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<5, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<6, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<7, false>),
                                      reinterpret_cast<const void*>(bazwide::sub_wide_kernel<8, false>)};
                // (the triangular instantiations, m >= 50, stay below 64 KiB: 33 KiB + 8 KiB at 64 antennas and 8 emitters)
                const int sz[10] = {evd_lds, scan_lds, sub_lds, sub_lds, sub_lds, sub_lds, sub_lds, sub_lds, sub_lds, sub_lds};
                bool ok = true;
                for (int k = 0; k < 10; ++k)
                    ok = ok && hipFuncSetAttribute(fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, sz[k]) == hipSuccess;
                if (!ok) { r = BAZ_MUSIC_E_HIP; break; }
            }
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SUB_EVD")) c->sub_evd = atoi(v);                   // lab / tests
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_WIDE_LITERAL")) c->wide_literal_only = atoi(v);   // lab / tests
            if (dev_malloc((void**)&c->dRefined, 2 * sizeof(unsigned long long)) != hipSuccess) { r = BAZ_MUSIC_E_NOMEM; break; }
            if (hipMemsetAsync(c->dRefined, 0, 2 * sizeof(unsigned long long), c->stream) != hipSuccess) { r = BAZ_MUSIC_E_HIP; break; }
            c->wide_cov_mfma = 1;
            c->wide_cov_blocks = 2u * (uint32_t)std::max(1, prop.multiProcessorCount);
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_WIDE_COV_MFMA")) c->wide_cov_mfma = (c->wide_cov_mfma && atoi(v)) ? 1 : 0;   // lab / tests
            c->wide_mfma = (n <= 8) ? 1 : 0;
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_WIDE_MFMA")) c->wide_mfma = (c->wide_mfma && atoi(v)) ? 1 : 0;   // lab / tests
            if (c->wide_mfma) {
                c->fb_steps = (resolution + 63) / 64;
                c->keep_mask = (resolution <= (1u << 16)) ? 0xFFFF0000u : 0xFFF00000u;
                c->tb_step_elems = (size_t)2 * ((2 * m + 3) / 4) * 64;
            }
            r = create_tables(c, table_ri);
            break;
        }
        c->fb_steps = (resolution + 63) / 64;
        // bin field of the top-n key: 16 bits up to 65,536 bins (d truncated by <= 2^-36), else 20 bits
        if (resolution > (1u << 20)) { r = BAZ_MUSIC_E_UNSUPPORTED; break; }
        c->keep_mask = (resolution <= (1u << 16)) ? 0xFFFF0000u : 0xFFF00000u;
        c->fb_step_elems = (size_t)2 * ((m * m + 3) / 4) * 64;
        c->tb_step_elems = (size_t)2 * ((2 * m + 3) / 4) * 64;
        // row classes of the spectrum port (scan kernel, ROW CLASSES): rows whose byte offset 4*res*i agrees mod 256.
        // Only where the scan is bound by its stores (m <= 5: <= 28 fp64 FMAs per 4 stored bytes); from m = 6 on it is
        // bound by the matrix core and the shifted table windows only cost (config 5: scan 0.706 -> 0.773 ms).
        if (resolution % 4u == 0 && m <= 5) {
            uint32_t gcd = 64;
            while (resolution % gcd) gcd >>= 1;
            c->nclass = 64u / gcd;
        }
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_NO_ROWCLASS")) { if (atoi(v)) c->nclass = 1; }   // lab: round-1 row order
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_NO_REFINE")) c->refine_off = atoi(v);              // lab
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_NO_REFINE_COUNT")) c->refine_nocount = atoi(v);    // lab: no statistic atomics
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COV_OLD")) c->lab_cov_old = atoi(v);               // lab
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SUB_EVD")) c->sub_evd = atoi(v);                   // lab / tests
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_NSPLIT")) c->force_nsplit = std::max(0, atoi(v));  // tests / lab
        {   // covariance + EVD fused (cov4_evd_kernel) wherever the dwordx4 covariance applies
            int fuse = 1;
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_FUSE")) fuse = atoi(v);                        // lab: 0 = two kernels
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SCAN_LDS_PAD")) c->scan_lds_pad = (uint32_t)std::max(0, atoi(v));
            c->fused_covevd = (fuse > 0 && m == 4 && (c->K % 256u) == 0 && !c->lab_cov_old) ? 1 : 0;
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cov4_evd_kernel, 256, 0) == hipSuccess && per_cu > 0)
                c->covevd_blocks = (uint32_t)per_cu * (uint32_t)std::max(1, prop.multiProcessorCount);
            else (void)hipGetLastError();
        }
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SIG_SCAN")) c->sig_scan = atoi(v);                  // lab / tests
        if (const char* v = getenv("BAZ_MUSIC_COARSE")) c->coarse = atoi(v);                      // A/B, tests
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COARSE_RG")) c->coarse_rg = atoi(v);                // lab
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COARSE_LAB")) c->coarse_lab = atoi(v);              // lab
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COARSE_STATS")) c->coarse_stats = atoi(v);          // lab
        if (m <= 8) {
            c->cs_tiles = round_up((resolution + 15) / 16, 8);
            if (dev_malloc((void**)&c->dMargin, sizeof(unsigned long long)) != hipSuccess) { r = BAZ_MUSIC_E_NOMEM; break; }
            if (hipMemsetAsync(c->dMargin, 0, sizeof(unsigned long long), c->stream) != hipSuccess) { r = BAZ_MUSIC_E_HIP; break; }
        }
        if (const char* v = getenv("BAZ_MUSIC_EXACT")) c->i8_on = atoi(v) ? 0 : 1;                // A/B: 1 = the fp64 scan everywhere
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_I8_ABL")) c->i8_abl = atoi(v);                  // lab
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SEQ_WALK")) c->seq_walk = atoi(v) ? 1 : 0;        // lab
#ifdef BAZ_MUSIC_LAB
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_SORT")) c->sort_mode = atoi(v) < 0 ? -1 : (atoi(v) ? 1 : 0);   // lab / tests: 0 never, 1 always
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_I8P")) c->i8p_on = atoi(v) ? 1 : 0;             // lab: 1 = the level-packed int8 scan at m <= 4
#endif
        if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COVEVD_TASK_ITEMS")) { const int t = atoi(v); c->covevd_task_items = (t == 64 || t == 32 || t == 16) ? t : 0; }
        bool wants_i8_stat = wants_i8_image(c);
#ifdef BAZ_MUSIC_LAB
        wants_i8_stat = wants_i8_stat || (c->i8p_on && m <= 4 && n <= 4);
#endif
        if (wants_i8_stat) {
            if (dev_malloc((void**)&c->dI8Stat, 8 * sizeof(unsigned long long)) != hipSuccess) { r = BAZ_MUSIC_E_NOMEM; break; }
            if (hipMemsetAsync(c->dI8Stat, 0, 8 * sizeof(unsigned long long), c->stream) != hipSuccess) { r = BAZ_MUSIC_E_HIP; break; }
        }
        {
            // ONE workgroup per CU (4 persistent waves, 8 KiB in flight each = 8 MB chip-wide): measured against 2 / 3 / 4 /
            // 6 (the occupancy limit) / 8 per CU, the fewest concurrent input streams read fastest -- 0.370 vs 0.396 ms per
            // 262,144 items inside the pipeline (profiles/r02_cov_grid.txt); more waves only queue more requests.
            c->cov4_resident_blocks = (uint32_t)std::max(1, prop.multiProcessorCount);
            c->num_cus = (uint32_t)std::max(1, prop.multiProcessorCount);
            if (const char* v = BAZ_LAB_ENV("BAZ_MUSIC_COV_BLOCKS_PER_CU"))    // lab: grid of the covariance kernel
                if (atoi(v) > 0) c->cov4_resident_blocks = (uint32_t)atoi(v) * (uint32_t)std::max(1, prop.multiProcessorCount);
        }
        if (dev_malloc((void**)&c->dRefined, 2 * sizeof(unsigned long long)) != hipSuccess) { r = BAZ_MUSIC_E_NOMEM; break; }
        if (hipMemsetAsync(c->dRefined, 0, 2 * sizeof(unsigned long long), c->stream) != hipSuccess) { r = BAZ_MUSIC_E_HIP; break; }
        r = create_tables(c, table_ri);
    } while (0);
    if (r != BAZ_MUSIC_OK) {
        baz_music_destroy(c);
        return r;
    }
    if (wide) {
        c->stage_name[BAZ_MUSIC_STAGE_COV] = !c->wide_cov_mfma ? "bazwide::cov_wide_kernel" : (m > 32 ? "bazwide::cov_wide_pairs_kernel" : "bazwide::cov_wide_mfma_kernel");
        c->stage_name[BAZ_MUSIC_STAGE_EVD] = "bazwide::evd_wide_kernel";
        c->stage_name[BAZ_MUSIC_STAGE_SCAN] = c->wide_mfma ? "bazwide::scan_wide_mfma_kernel" : "bazwide::scan_wide_kernel";
        c->stage_name[BAZ_MUSIC_STAGE_MERGE] = c->wide_mfma ? (n <= 2 ? "bazmusic::topn_merge_kernel<2>" : (n <= 4 ? "bazmusic::topn_merge_kernel<4>" : "bazmusic::topn_merge_kernel<8>")) : "bazwide::topn_wide_kernel";
        *out = c;
        return BAZ_MUSIC_OK;
    }
    char buf[128];
    snprintf(buf, sizeof(buf), m <= 8 ? "bazmusic::cov_mfma_kernel<%u>" : "bazmusic::cov_mfma2_kernel<%u>", m);
    c->stage_name[BAZ_MUSIC_STAGE_COV] = buf;
    if (m == 4 && (c->K % 256u) == 0 && !c->lab_cov_old) c->stage_name[BAZ_MUSIC_STAGE_COV] = "bazmusic::cov4_x4_kernel";
    if (c->fused_covevd) c->stage_name[BAZ_MUSIC_STAGE_COV] = "bazmusic::cov4_evd_kernel";
    snprintf(buf, sizeof(buf), m <= 4 ? "bazmusic::evd_proj_kernel<%u>" : "bazmusic::evd_proj_lds_kernel<%u>", m);
    c->stage_name[BAZ_MUSIC_STAGE_EVD] = buf;
    snprintf(buf, sizeof(buf), i8_active(c) ? "bazmusic::scan_i8_kernel<%u," : "bazmusic::scan_mfma_kernel<%u,", m);
    c->stage_name[BAZ_MUSIC_STAGE_SCAN] = buf;
    snprintf(buf, sizeof(buf), "bazmusic::topn_merge_kernel<%u>", topn_list_len(n));
    c->stage_name[BAZ_MUSIC_STAGE_MERGE] = buf;
    snprintf(c->scan_names[0], sizeof(c->scan_names[0]), "bazmusic::scan_mfma_kernel<%u,", m);
    snprintf(c->scan_names[1], sizeof(c->scan_names[1]), "bazmusic::scan_i8_kernel<%u,", m);
    snprintf(c->scan_names[2], sizeof(c->scan_names[2]), "bazmusic::scan_coarse_kernel<%u,", m);
    snprintf(c->scan_names[3], sizeof(c->scan_names[3]), "bazmusic::scan_i8p_kernel<%u,", m);
    *out = c;
    return BAZ_MUSIC_OK;
}

void baz_music_destroy(baz_music_ctx* c)
{
    if (!c) return;
    {
        DeviceGuard guard(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (auto& p : c->prof)
            for (auto e : p.ev) (void)hipEventDestroy(e);
        if (c->h_al_big) (void)hipHostFree(c->h_al_big);
        if (c->s_tab) (void)hipStreamSynchronize(c->s_tab);
        {
            TableSet act = active_table_set(c);
            free_table_set(act);
            install_table_set(c, act);
            free_table_set(c->shadow);
        }
        if (c->dRaw) (void)dev_free(c->dRaw);
        if (c->hRaw) (void)hipHostFree(c->hRaw);
        if (c->dTabStats) (void)dev_free(c->dTabStats);
        if (c->hTabStats) (void)hipHostFree(c->hTabStats);
        if (c->ev_swap) (void)hipEventDestroy(c->ev_swap);
        if (c->s_tab) (void)hipStreamDestroy(c->s_tab);
#ifdef BAZ_MUSIC_LAB
        if (c->dKeys) (void)dev_free(c->dKeys);
        if (c->dPerm) (void)dev_free(c->dPerm);
        if (c->dHist) (void)dev_free(c->dHist);
        if (c->dCursor) (void)dev_free(c->dCursor);
        if (c->dFire) (void)dev_free(c->dFire);
        if (c->hFire) (void)hipHostFree(c->hFire);
#endif
        if (c->dCand) (void)dev_free(c->dCand);
        if (c->dR) (void)dev_free(c->dR);
        if (c->dQ) (void)dev_free(c->dQ);
        if (c->dG) (void)dev_free(c->dG);
        if (c->dRedo) (void)dev_free(c->dRedo);
        if (c->dSs) (void)dev_free(c->dSs);
        if (c->dI8Stat) (void)dev_free(c->dI8Stat);
        if (c->dMargin) (void)dev_free(c->dMargin);
        if (c->dRefined) (void)dev_free(c->dRefined);
        if (c->dPeakSpec) (void)dev_free(c->dPeakSpec);
        if (c->dGw) (void)dev_free(c->dGw);
        if (c->dWS) (void)dev_free(c->dWS);
        if (c->dSw) (void)dev_free(c->dSw);
        host_unregister_all_locked(c);
        free_slots(c);
        if (c->ev_in) (void)hipEventDestroy(c->ev_in);
        if (c->ev_out) (void)hipEventDestroy(c->ev_out);
        if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
        if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
}

int baz_music_set_table(baz_music_ctx* c, const float* table_ri)
{
    if (!c || !table_ri) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> tl(c->tab_mtx);   // one retune at a time; work() is held up for the pointer exchange only (retune())
    DeviceGuard guard(c->device);
    return retune(c, table_ri, false);
}

int baz_music_last_retune_ms(baz_music_ctx* c, double* total_ms, double* swap_ms)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> tl(c->tab_mtx);
    if (total_ms) *total_ms = c->last_retune_ms;
    if (swap_ms) *swap_ms = c->last_swap_wait_ms;
    return BAZ_MUSIC_OK;
}

int baz_music_set_stream(baz_music_ctx* c, void* hip_stream)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return BAZ_MUSIC_OK;
}

int baz_music_sync(baz_music_ctx* c)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BAZ_MUSIC_OK;
}

int baz_music_reserve(baz_music_ctx* c, uint32_t max_batch)
{
    if (!c || max_batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->wide) {
        const uint32_t pass = std::min(max_batch, wide_pass_items(c));
        const int rw = ensure_wide_workspace(c, pass);
        // ... and the candidate lists of the matrix-core scan's largest pass (they used to be allocated inside the first process call)
        return (rw || !c->wide_mfma || c->wide_literal_only) ? rw : ensure_candidates(c, wide_cand_entries(c, pass));
    }
    int r = ensure_workspace(c, max_batch);
#ifdef BAZ_MUSIC_LAB
    if (r == BAZ_MUSIC_OK && c->sort_mode != 0) r = ensure_sort_workspace(c, max_batch);
#endif
    return r ? r : reserve_candidates(c, max_batch);
}

uint32_t baz_music_q_stride(uint32_t batch) { return round_up(batch ? batch : 1, 64); }

int baz_music_process_device(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_ang, void* d_lvl,
                             void* d_spec)
{
    if (!c || !d_in || !d_ang) return BAZ_MUSIC_E_INVALID;
    if (batch == 0) return BAZ_MUSIC_OK;
    std::lock_guard<std::mutex> lk(c->mtx);   // .cc:101
    DeviceGuard guard(c->device);
    (void)hipGetLastError();                  // launches below are checked with hipGetLastError(): start from a clean slate
    int r = begin_statistic(c);
    return r ? r : process_device_locked(c, d_in, batch, d_ang, d_lvl, d_spec);
}

int baz_music_process_device_on(baz_music_ctx* c, void* caller_stream, const void* d_in, uint32_t batch, void* d_ang,
                                void* d_lvl, void* d_spec)
{
    if (!c || !d_in || !d_ang) return BAZ_MUSIC_E_INVALID;
    if (batch == 0) return BAZ_MUSIC_OK;
    std::lock_guard<std::mutex> lk(c->mtx);   // .cc:101
    DeviceGuard guard(c->device);
    (void)hipGetLastError();                  // launches below are checked with hipGetLastError(): start from a clean slate
    hipStream_t cs = static_cast<hipStream_t>(caller_stream);
    const bool foreign = (cs != c->stream);
    if (foreign) {
        if (!c->ev_in) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        if (!c->ev_out) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_in, cs));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_in, 0));
    }
    int r = begin_statistic(c);
    if (r == BAZ_MUSIC_OK) r = process_device_locked(c, d_in, batch, d_ang, d_lvl, d_spec);
    if (foreign) {   // also after a failed launch: whatever was enqueued is ordered before the caller's next work
        hipError_t e = hipEventRecord(c->ev_out, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs, c->ev_out, 0);
        if (e != hipSuccess && r == BAZ_MUSIC_OK) r = hip_fail(c, e, "hipStreamWaitEvent(caller_stream)");
    }
    return r;
}

int baz_music_process(baz_music_ctx* c, const float* in_ri, uint32_t batch, float* ang, float* lvl,
                      float* spectrum)
{
    if (!c || !in_ri || !ang) return BAZ_MUSIC_E_INVALID;
    if (batch == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    (void)hipGetLastError();

    if (c->auto_pin) {   // the caller's (scheduler's) buffers: lock what this call touches and is not locked yet
        // (the two DMA targets; ang / lvl travel through the slots' own page-locked image)
        (void)host_register_locked(c, in_ri, (size_t)batch * c->nsamples * 8);
        if (spectrum) (void)host_register_locked(c, spectrum, (size_t)batch * c->res * 4);
    }
    // Cutting a call into chunks (H2D of chunk i+1, the kernels of chunk i and D2H of chunk i-1 overlap on three
    // streams) only pays when a chunk carries enough bytes to hide what the extra copies, events and stream hops cost.
    // Measured under the scheduler model (profiles/r02_flowgraph_model_rates.txt): a 128-item config-2 call (2.9 MB) cut
    // in two ran at 0.41-0.53 ms, as one chunk on one stream at 0.20-0.22 ms; pageable 2,048-item calls (46 MB) in four
    // chunks were SLOWER per item than 512-item calls in one (the runtime stages pageable copies and blocks in them, so
    // little overlaps).  Hence:
    //   page-locked caller memory   below 64 MiB of traffic (BAZ_MUSIC_SINGLE_MIB; without the spectrum port: always) ONE chunk on ONE stream -- without copies
    //                               when zero-copy applies: a 1,024-item config-2 call (23 MB) ran at 2.06e6 items/s that way and
    //                               at 1.53e6 cut into three pipelined chunks (round 3, scripts/gpu/r03l.sh); above, >= 4 chunks of
    //                               >= 8 MiB and <= 32 MiB each;
    //   pageable caller memory      one chunk up to 64 MiB, 64-MiB chunks beyond (profiles/r01h_hostfed_chunk_sweep.txt);
    //   BAZ_MUSIC_CHUNK_MIB         forces the chunk size (tests, lab).
    const size_t per_item = (size_t)c->nsamples * 8 + (spectrum ? (size_t)c->res * 4 : 0) + (size_t)c->n * 8;   // what this call moves per item
    // "locked" is a statement about the WHOLE of both ranges (the first byte alone is not enough)
    const bool locked = range_is_pinned(in_ri, (size_t)batch * c->nsamples * 8) &&
                        (!spectrum || range_is_pinned(spectrum, (size_t)batch * c->res * 4));
    uint32_t chunk;
    if (c->chunk_bytes) {
        chunk = (uint32_t)std::min<size_t>(batch, std::max<size_t>(64, c->chunk_bytes / per_item));
    } else if (!locked) {
        chunk = (uint32_t)std::min<size_t>(batch, std::max<size_t>(64, (64u << 20) / per_item));
    } else if (!spectrum || (size_t)batch * per_item < (size_t)c->single_limit_mib << 20) {
        // (without the spectrum port a call moves its input only: the covariance kernel reads it over the link at the link's rate, and cutting the
        // call can only add to that -- 8,192 / 16,384-item calls ran at 6.5 / 6.6e6 items/s as one zero-copy sequence, at 5.5e6 cut into chunks:
        // profiles/r06_hostfed_chunk_sweep.txt)
        chunk = batch;
    } else {
        const size_t floor_items = std::max<size_t>(64, (8u << 20) / per_item);
        const size_t by_count = std::max<size_t>(floor_items, ((size_t)batch + 3) / 4);
        chunk = (uint32_t)std::min<size_t>(batch, std::min<size_t>(std::max<size_t>(64, (32u << 20) / per_item), by_count));
    }
    const bool single = chunk >= batch;
    const bool want_spec = spectrum != nullptr;
    int r = BAZ_MUSIC_OK;

    // Small call on page-locked caller memory: no copies at all.  The kernels address the caller's buffers over PCIe
    // (hipHostGetDevicePointer): the covariance kernel reads every input byte once, the scan stores every spectrum value
    // once, ang / lvl land in a page-locked image -- each byte crosses the link once, inside the launches, instead of
    // through two or three DMA transfers with their submission and completion latencies.  Larger calls keep the pipelined
    // copies below (there the DMA engines overlap both directions with the kernels).
    // A launch sequence uses the link in one direction at a time (a 1,024-item config-2 call: 0.21 ms in, then 0.26 ms out), and round 5
    // tried twice to overlap the two inside a call -- sub-chunks alternating between two contexts with event-staggered kernels, and the
    // input of sub-chunk i + 1 by DMA on a side stream beside the kernels of sub-chunk i.  Both LOSE at every call size: each sub-chunk
    // costs 0.1 - 0.17 ms of copy start-up and cross-stream event hops on this stack (profiles/r05_hostfed_calls.txt).  A third form without
    // events (finished groups of items counted by cov4_evd_kernel, their scans on a second stream behind a spinning one-wave gate) ran the
    // two side by side and gained nothing: kernel-issued reads and writes of host memory share most of one budget (75.8 GB/s in + out
    // against 56 / 53 alone; the covariance took twice as long beside a scan; profiles/r05_hostfed_overlap_negative.txt).  What pays is a
    // larger call (the host block's look-back) and more, shorter covariance tasks for small batches (launch_covevd).
    if (single && locked && c->zero_copy) {
        void *z_in = nullptr, *z_spec = nullptr, *z_al = nullptr;
        bool ok = ensure_h_al_big(c, (size_t)batch * c->n * 2) == BAZ_MUSIC_OK;
        ok = ok && hipHostGetDevicePointer(&z_in, (void*)in_ri, 0) == hipSuccess;
        if (ok && want_spec) ok = hipHostGetDevicePointer(&z_spec, (void*)spectrum, 0) == hipSuccess;
        if (ok) ok = hipHostGetDevicePointer(&z_al, (void*)c->h_al_big, 0) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();           // not addressable from the device after all: the copy path below
        } else {
            float* z_ang = static_cast<float*>(z_al);
            int zr = BAZ_MUSIC_OK;
            if (!c->wide) {
                zr = ensure_workspace(c, batch);
                if (zr == BAZ_MUSIC_OK) zr = reserve_candidates(c, batch);
            }
            if (zr == BAZ_MUSIC_OK) zr = begin_statistic(c);
            if (zr == BAZ_MUSIC_OK)
                zr = process_device_locked(c, z_in, batch, z_ang, z_ang + (size_t)batch * c->n, want_spec ? z_spec : nullptr);
            const hipError_t es = hipStreamSynchronize(c->stream);   // also after a failed launch
            if (zr == BAZ_MUSIC_OK && es != hipSuccess) zr = hip_fail(c, es, "hipStreamSynchronize");
            if (zr != BAZ_MUSIC_OK) return zr;
            const size_t cnt = (size_t)batch * c->n;
            memcpy(ang, c->h_al_big, cnt * 4);
            if (lvl) memcpy(lvl, c->h_al_big + cnt, cnt * 4);
            return (int)batch;
        }
    }

    r = ensure_slots(c, chunk, want_spec);
    if (r) return r;
    if (!c->wide) {
        r = ensure_workspace(c, chunk);
        if (r) return r;
        r = reserve_candidates(c, chunk);
        if (r) return r;
    }

    int rc = BAZ_MUSIC_OK;
    uint32_t idx = 0;
    rc = begin_statistic(c);                  // statistic of this call (all its chunks add to the same counter)
    // A failing HIP call must not return from inside the loop: copies already queued still target the caller's
    // buffers, so every exit goes through the drain below.
#define HIP_STEP(call)                                                     \
    if (rc == BAZ_MUSIC_OK) {                                              \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) rc = hip_fail(c, e__, #call);               \
    }
    // ang / lvl of a finished chunk: from the slot's page-locked image to the caller's two streams
    auto hand_over = [&](baz_music_ctx::Slot& sl) {
        if (!sl.pend_nb) return;
        const size_t cnt = (size_t)sl.pend_nb * c->n;
        memcpy(ang + (size_t)sl.pend_done * c->n, sl.h_al, cnt * 4);
        if (lvl) memcpy(lvl + (size_t)sl.pend_done * c->n, sl.h_al + cnt, cnt * 4);
        sl.pend_nb = 0;
    };
    hipStream_t s_in = single ? c->stream : c->s_h2d, s_out = single ? c->stream : c->s_d2h;
    // Page-locked caller memory, more than one chunk (round 6): the WHOLE call is enqueued without the host waiting for anything -- four
    // slots, a slot's reuse ordered on the device (the copy into slot.in behind the kernels that last read it, the kernels that write
    // slot.spec / slot.al behind the copy that last drained them), ang / lvl of every chunk copied straight into the call's page-locked
    // image and handed to the caller after the one synchronise at the end.  The copy engines then run both directions of the link side
    // by side (96 GB/s in + out against 57 one way, profiles/r05_hostfed_overlap_negative.txt) with the kernels in their shadow; the
    // two-slot form below blocked the host in hipEventSynchronize before every third chunk and so never had more than one copy queued
    // per direction.
    const bool deep = !single && locked && ensure_h_al_big(c, (size_t)batch * c->n * 2) == BAZ_MUSIC_OK;
    const uint32_t nslot = deep ? (uint32_t)BAZ_MUSIC_NSLOT : 2u;
    for (uint32_t done = 0; done < batch && rc == BAZ_MUSIC_OK; done += chunk, ++idx) {
        baz_music_ctx::Slot& sl = c->slot[idx % nslot];
        const uint32_t nb = std::min(chunk, batch - done);
        if (sl.busy && !deep) {   // chunk idx-2 used this slot: its outputs must be on the host before we reuse it
            HIP_STEP(hipEventSynchronize(sl.d2h));
            if (rc == BAZ_MUSIC_OK) hand_over(sl);
            sl.busy = false;
        }
        if (sl.busy && deep) {    // chunk idx-4 used this slot: order the reuse on the device, the host runs ahead
            HIP_STEP(hipStreamWaitEvent(s_in, sl.comp, 0));          // its kernels have read slot.in
            HIP_STEP(hipStreamWaitEvent(c->stream, sl.d2h, 0));      // its outputs have left slot.spec / slot.al
        }
        float* d_ang = sl.al;
        float* d_lvl = sl.al + (size_t)nb * c->n;
        HIP_STEP(copy_host_range(sl.in, in_ri + (size_t)done * c->nsamples * 2, (size_t)nb * c->nsamples * 8,
                                 hipMemcpyHostToDevice, s_in));
        if (!single) {
            HIP_STEP(hipEventRecord(sl.h2d, s_in));
            HIP_STEP(hipStreamWaitEvent(c->stream, sl.h2d, 0));
        }
        if (rc == BAZ_MUSIC_OK) rc = process_device_locked(c, sl.in, nb, d_ang, d_lvl, want_spec ? sl.spec : nullptr);
        if (!single) {
            HIP_STEP(hipEventRecord(sl.comp, c->stream));
            HIP_STEP(hipStreamWaitEvent(s_out, sl.comp, 0));
        }
        // (deep: chunk `done` keeps its [ang | lvl] image at 2 n done floats of the call's image: handed over after the final synchronise)
        HIP_STEP(hipMemcpyAsync(deep ? c->h_al_big + (size_t)done * c->n * 2 : sl.h_al, sl.al, (size_t)nb * c->n * 8, hipMemcpyDeviceToHost, s_out));
        if (want_spec)
            HIP_STEP(copy_host_range(spectrum + (size_t)done * c->res, sl.spec, (size_t)nb * c->res * 4,
                                     hipMemcpyDeviceToHost, s_out));
        if (!single) HIP_STEP(hipEventRecord(sl.d2h, s_out));
        sl.busy = (rc == BAZ_MUSIC_OK);
        if (sl.busy && !deep) {
            sl.pend_done = done;
            sl.pend_nb = nb;
        }
    }
#undef HIP_STEP
    // drain (also on the error path) so that no copy still targets the caller's buffers
    if (!single) (void)hipStreamSynchronize(c->s_h2d);
    (void)hipStreamSynchronize(c->stream);
    if (!single) (void)hipStreamSynchronize(c->s_d2h);
    for (auto& sl : c->slot) {
        if (rc == BAZ_MUSIC_OK && sl.busy) hand_over(sl);
        sl.busy = false;
        sl.pend_nb = 0;
    }
    if (deep && rc == BAZ_MUSIC_OK) {           // every chunk's [ang | lvl] image is on the host now
        for (uint32_t done = 0; done < batch; done += chunk) {
            const size_t cnt = (size_t)std::min(chunk, batch - done) * c->n;
            const float* img = c->h_al_big + (size_t)done * c->n * 2;
            memcpy(ang + (size_t)done * c->n, img, cnt * 4);
            if (lvl) memcpy(lvl + (size_t)done * c->n, img + cnt, cnt * 4);
        }
    }
    return rc ? rc : (int)batch;
}

int baz_music_host_register(baz_music_ctx* c, const void* p, size_t bytes)
{
    if (!c || (!p && bytes)) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return host_register_locked(c, p, bytes);
}

int baz_music_set_host_pinning(baz_music_ctx* c, int enable)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    c->auto_pin = enable ? 1 : 0;
    return BAZ_MUSIC_OK;
}

int baz_music_host_unregister_all(baz_music_ctx* c)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->s_h2d ? c->s_h2d : c->stream);
    host_unregister_all_locked(c);
    return BAZ_MUSIC_OK;
}

uint64_t baz_music_host_pinned_bytes(baz_music_ctx* c)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mtx);
    return c->pinned_bytes;
}

int baz_music_profile(baz_music_ctx* c, int enable)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (enable) {
        for (auto& p : c->prof) { p.used = 0; p.total_ms = 0.0; p.launches = 0; }
        c->profiling = (enable == 2) ? 2 : 1;
    } else {
        prof_collect(c);
        c->profiling = 0;
    }
    return BAZ_MUSIC_OK;
}

int baz_music_stage_ms(baz_music_ctx* c, int stage, double* total_ms, uint64_t* launches)
{
    if (!c || stage < 0 || stage >= BAZ_MUSIC_NUM_STAGES) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (total_ms) *total_ms = c->prof[stage].total_ms;
    if (launches) *launches = c->prof[stage].launches;
    return BAZ_MUSIC_OK;
}

const char* baz_music_stage_name(baz_music_ctx* c, int stage)
{
    if (!c || stage < 0 || stage >= BAZ_MUSIC_NUM_STAGES) return "";
    if (stage == BAZ_MUSIC_STAGE_SCAN && !c->wide) {
        // the kernel the LAST scan launch took (it depends on the wiring of the call and on the table in force: ADVICE r4);
        // before the first launch: what a call with the spectrum port would take
        // (the four names are formatted once, in baz_music_create: the pointer handed out stays valid for the context's life)
        std::lock_guard<std::mutex> lk(c->mtx);
        const int kind = c->scan_kind >= 0 ? c->scan_kind : (i8_active(c) ? 1 : (i8p_active(c) ? 3 : 0));
        return c->scan_names[kind >= 0 && kind < 4 ? kind : 0];
    }
    return c->stage_name[stage].c_str();
}

int baz_music_debug_cov(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_R)
{
    if (!c || !d_in || !d_R || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    if (c->wide) return launch_cov_wide(c, static_cast<const float*>(d_in), batch, static_cast<double2*>(d_R));
    if (c->fused_covevd) {   // the product's covariance lives inside the fused kernel: run THAT with its R tap
        int r = ensure_workspace(c, batch);
        if (r) return r;
        return launch_covevd(c, static_cast<const float*>(d_in), batch, c->dQ, baz_music_q_stride(batch), c->dG,
                             static_cast<double2*>(d_R));
    }
    return launch_cov(c, static_cast<const float*>(d_in), batch, static_cast<double2*>(d_R));
}

int baz_music_debug_q(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_Q)
{
    if (!c || !d_in || !d_Q || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    if (c->wide) return BAZ_MUSIC_E_UNSUPPORTED;   // the wide path forms no projector (literal form straight from G)
    int r = ensure_workspace(c, batch);
    if (r) return r;
    const uint32_t qstride = baz_music_q_stride(batch);
    if (c->fused_covevd) return launch_covevd(c, static_cast<const float*>(d_in), batch, static_cast<double*>(d_Q), qstride, c->dG);
    r = launch_cov(c, static_cast<const float*>(d_in), batch, c->dR);
    if (r) return r;
    return launch_evd(c, c->dR, batch, static_cast<double*>(d_Q), qstride, c->dG);
}

// Validation of the coarse-gated scan's error bound on this hardware (scan_coarse_kernels.hip.h, VAL): covariance + EVD of
// the batch, then EVERY (item, bin) in both forms; *worst = max |coarse / SC - exact| / (2^-16 (S + |exact|)).  The gate is
// sound while this stays below 1; the bound was derived with a factor > 2 to spare.
int baz_music_debug_coarse_margin(baz_music_ctx* c, const void* d_in, uint32_t batch, float* worst)
{
    if (!c || !d_in || !worst || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    if (c->wide || c->m > 8 || c->n > 4 || !c->cs_ok) return BAZ_MUSIC_E_UNSUPPORTED;
    int r = ensure_workspace(c, batch);
    if (r) return r;
    r = reserve_candidates(c, batch);
    if (r) return r;
    const uint32_t qstride = baz_music_q_stride(batch);
    if (c->fused_covevd) r = launch_covevd(c, static_cast<const float*>(d_in), batch, c->dQ, qstride, c->dG);
    else {
        r = launch_cov(c, static_cast<const float*>(d_in), batch, c->dR);
        if (!r) r = launch_evd(c, c->dR, batch, c->dQ, qstride, c->dG);
    }
    if (r) return r;
    HIP_TRY(c, hipMemsetAsync(c->dMargin, 0, sizeof(unsigned long long), c->stream));
    ScanRefine rf;
    rf.Gs = nullptr; rf.TB = c->dTB + c->tb_step_elems; rf.below = 0.0; rf.count = nullptr; rf.A2 = nullptr;
    const bool big = c->m > 4;                   // (m >= 5: the RG = 2, TPP = 4 instantiation, like the scan's)
    const uint32_t per_group = big ? 64u * (uint32_t)coarse_rg_wide((int)c->m, 2) : 256u;
    const uint32_t groups = (batch + per_group - 1) / per_group, nph = c->cs_tiles / (big ? 4 : 8);
    float* d_dump = nullptr;                     // lab (BAZ_MUSIC_DEBUG_DUMP=<file>): every ratio, [item][bin] float32
    const char* dump_path = BAZ_LAB_ENV("BAZ_MUSIC_DEBUG_DUMP");
    if (dump_path && dev_malloc((void**)&d_dump, (size_t)batch * c->res * sizeof(float)) != hipSuccess) d_dump = nullptr;
    if (d_dump) (void)hipMemsetAsync(d_dump, 0, (size_t)batch * c->res * sizeof(float), c->stream);
#define BAZ_VAL(MV, NV, RGV, TPV)                                                                                           \
    hipLaunchKernelGGL((scan_coarse_kernel<MV, NV, RGV, TPV, true>), dim3(groups), dim3(256), 0, c->stream, c->dQ, c->dCS, \
                       c->dCS + (size_t)(c->cs_tiles + 1) * cs_c_units(MV), c->dCand,                                                  \
                       batch, c->res, qstride, nph, 1u, c->keep_mask, c->n, rf, c->cs, c->dMargin, d_dump)
    const bool n2 = c->n <= 2;
    switch (c->m) {
        case 2: BAZ_VAL(2, 2, 4, 8); break;
        case 3: if (n2) BAZ_VAL(3, 2, 4, 8); else BAZ_VAL(3, 4, 4, 8); break;
        case 4: if (n2) BAZ_VAL(4, 2, 4, 8); else BAZ_VAL(4, 4, 4, 8); break;
        case 5: BAZ_VAL(5, 2, coarse_rg_wide(5, 2), 4); break;         // (the lists are not used by the validation build)
        case 6: BAZ_VAL(6, 2, coarse_rg_wide(6, 2), 4); break;
        case 7: BAZ_VAL(7, 2, coarse_rg_wide(7, 2), 4); break;
        default: BAZ_VAL(8, 2, coarse_rg_wide(8, 2), 4); break;
    }
#undef BAZ_VAL
    HIP_TRY(c, hipGetLastError());
    unsigned long long packed = 0;
    HIP_TRY(c, hipMemcpyAsync(&packed, c->dMargin, sizeof(packed), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d_dump) {
        std::vector<float> h((size_t)batch * c->res);
        if (hipMemcpy(h.data(), d_dump, h.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(dump_path, "wb")) { fwrite(h.data(), sizeof(float), h.size(), f); fclose(f); }
        (void)dev_free(d_dump);
    }
    const unsigned int bits = (unsigned int)(packed >> 32);
    std::memcpy(worst, &bits, sizeof(float));
    if (BAZ_LAB_ENV("BAZ_MUSIC_DEBUG_MARGIN"))   // lab: where the worst value sits
        fprintf(stderr, "[baz_music] coarse margin %.4g at bin %u, item %% 4096 = %u\n", *worst, (unsigned)(packed & 0xFFFFFu),
                (unsigned)((packed >> 20) & 0xFFFu));
    return BAZ_MUSIC_OK;
}

// lab statistic (BAZ_MUSIC_COARSE_STATS=1 at create): exact (16-item row group, 16-bin tile) evaluations of all coarse-gated
// scan launches since the last read; resets the counter.  -1 when the statistic is off.
int64_t baz_music_debug_coarse_fired(baz_music_ctx* c)
{
    if (!c || !c->coarse_stats || !c->dMargin) return -1;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    unsigned long long v = 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&v, c->dMargin, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemsetAsync(c->dMargin, 0, sizeof(v), c->stream) != hipSuccess) return -1;
    return (int64_t)v;
}

// Validation of the int8 scan's a-priori bounds on this hardware (scan_i8_kernels.hip.h, VAL): covariance + EVD of the batch,
// then EVERY (item, bin) in the four-, five-, seven-digit and the fp64 form; worst[0] = max |d5 - d| / E5, worst[1] = max |d7 - d| /
// allowance, worst[2] = max |d4 - d| / E4 over the items that take the integer forms.
int baz_music_debug_i8_margin(baz_music_ctx* c, const void* d_in, uint32_t batch, float* worst)
{
    if (!c || !d_in || !worst || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
#ifdef BAZ_MUSIC_LAB
    if (!c->wide && c->m <= 4 && c->m >= 2 && c->n <= 4 && c->dIP && c->i8_ok) {
        // 2 .. 4 antennas: the level-packed form (scan_i8p_kernels.hip.h), VAL instantiation, one bin range per row
        int r = ensure_workspace(c, batch);
        if (r) return r;
        r = reserve_candidates(c, batch);
        if (r) return r;
        const uint32_t qstride = baz_music_q_stride(batch);
        if (c->fused_covevd) r = launch_covevd(c, static_cast<const float*>(d_in), batch, c->dQ, qstride, c->dG);
        else {
            r = launch_cov(c, static_cast<const float*>(d_in), batch, c->dR);
            if (!r) r = launch_evd(c, c->dR, batch, c->dQ, qstride, c->dG);
        }
        if (r) return r;
        HIP_TRY(c, hipMemsetAsync(c->dI8Stat + 2, 0, 3 * sizeof(unsigned long long), c->stream));
        ScanRefine rf;
        rf.Gs = nullptr; rf.TB = c->dTB + c->tb_step_elems; rf.below = 0.0; rf.count = nullptr; rf.A2 = nullptr;
        const ScanGeom G = scan_geometry(batch, c->fb_steps, c->nclass, 1, c->m);
        const uint4* p1 = c->dIP + I8P_STEP_UNITS;
        const uint4* p2 = p1 + i8p_operand_units(c->fb_steps);
#define BAZ_VALP(MV)                                                                                                          \
    case MV: hipLaunchKernelGGL((scan_i8p_kernel<MV, 2, false, false, true>), dim3(G.blocks), dim3(256), 0, c->stream, c->dQ, p1, p2, \
                                c->dFB + c->fb_step_elems, nullptr, c->dCand, batch, c->res, qstride, 1u, c->nclass, G.rows_per_class, \
                                c->keep_mask, c->n, rf, c->i8, nullptr, c->dI8Stat + 2); break;
        switch (c->m) {
#ifndef BAZ_MUSIC_QUICK
            BAZ_VALP(2) BAZ_VALP(3)
#endif
            BAZ_VALP(4)
            default: return BAZ_MUSIC_E_UNSUPPORTED;
        }
#undef BAZ_VALP
        HIP_TRY(c, hipGetLastError());
        unsigned long long packed[3] = {0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(packed, c->dI8Stat + 2, sizeof(packed), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < 3; ++k) {
            const unsigned int bits = (unsigned int)packed[k];
            std::memcpy(worst + k, &bits, sizeof(float));
        }
        return BAZ_MUSIC_OK;
    }
#endif
    if (c->wide || !c->dIB || !c->i8_ok || c->m < 6 || c->m > 16 || c->n > 4) return BAZ_MUSIC_E_UNSUPPORTED;
    int r = ensure_workspace(c, batch);
    if (r) return r;
    r = reserve_candidates(c, batch);
    if (r) return r;
    const uint32_t qstride = baz_music_q_stride(batch);
    const int on = c->i8_on;
    c->i8_on = 1;                       // (the EVD must write the projector coefficients: short_form_in_use())
    r = launch_cov(c, static_cast<const float*>(d_in), batch, c->dR);
    if (!r) r = launch_evd(c, c->dR, batch, c->dQ, qstride, c->dG);
    c->i8_on = on;
    if (r) return r;
    HIP_TRY(c, hipMemsetAsync(c->dI8Stat + 2, 0, 3 * sizeof(unsigned long long), c->stream));
    ScanRefine rf;
    rf.Gs = nullptr; rf.TB = c->dTB + c->tb_step_elems; rf.below = 0.0; rf.count = nullptr; rf.A2 = nullptr;
    const uint32_t blocks = (batch + 63) / 64;
#define BAZ_VAL8(MV)                                                                                                      \
    case MV: hipLaunchKernelGGL((scan_i8_kernel<MV, 2, false, false, true>), dim3(blocks), dim3(256), 0, c->stream, c->dQ, \
                                c->dIB, c->dIB + i8_image_bytes5(c->m, c->fb_steps) / 16, c->dFB + c->fb_step_elems, nullptr, c->dCand, \
                                batch, c->res, qstride, 1u, c->keep_mask, c->n, rf, c->i8, nullptr, c->dI8Stat + 2); break;
    switch (c->m) {
        BAZ_VAL8(6) BAZ_VAL8(7) BAZ_VAL8(8) BAZ_VAL8(9) BAZ_VAL8(10) BAZ_VAL8(11) BAZ_VAL8(12) BAZ_VAL8(13) BAZ_VAL8(14)
        BAZ_VAL8(15) BAZ_VAL8(16)
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
#undef BAZ_VAL8
    HIP_TRY(c, hipGetLastError());
    unsigned long long packed[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(packed, c->dI8Stat + 2, sizeof(packed), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) {
        const unsigned int bits = (unsigned int)packed[k];
        std::memcpy(worst + k, &bits, sizeof(float));
    }
    return BAZ_MUSIC_OK;
}

// Statistic of the int8 scan since the last read (resets): wave tiles (16 items x 16 bins) that ran the refined (seven-digit)
// form, and wave tiles walked.  BAZ_MUSIC_E_UNSUPPORTED where that scan does not exist for the configuration.
int baz_music_debug_i8_stats(baz_music_ctx* c, uint64_t* refined_tiles, uint64_t* tiles)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    if (!c->dI8Stat) return BAZ_MUSIC_E_UNSUPPORTED;
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(v, c->dI8Stat, sizeof(v), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemsetAsync(c->dI8Stat, 0, sizeof(v), c->stream));
    if (refined_tiles) *refined_tiles = v[0];
    if (tiles) *tiles = v[1];
    return BAZ_MUSIC_OK;
}

#ifdef BAZ_MUSIC_LAB
// lab: the s_memtime sums of scan_i8_kernel's ABL 8192 instrumentation since the last read (wait, stores, barrier, whole loop; per wave, summed)
extern "C" __attribute__((visibility("default"))) int baz_music_debug_i8_times(baz_music_ctx* c, uint64_t out[4])
{
    if (!c || !c->dI8Stat) return BAZ_MUSIC_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    unsigned long long v[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(v, c->dI8Stat + 4, sizeof(v), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemsetAsync(c->dI8Stat + 4, 0, sizeof(v), c->stream));
    for (int i = 0; i < 4; ++i) out[i] = v[i];
    return BAZ_MUSIC_OK;
}
#endif

// 1 when this context's scan runs on the int8 matrix core (6 <= m <= 16, n <= 4, a table whose digit image exists, not
// BAZ_MUSIC_EXACT=1), else 0.
int baz_music_uses_i8_scan(const baz_music_ctx* c) { return (c && !c->wide && (i8_active(c) || i8p_active(c))) ? 1 : 0; }

// HOST-ONLY tap (no device needed): the bin ranges per item i8_nsplit() chooses for a launch of `batch` items over `nsteps`
// 64-bin steps on a device with `slots` resident workgroups (CUs x workgroups per CU).
uint32_t baz_music_debug_i8_nsplit(uint32_t batch, uint32_t nsteps, uint32_t slots)
{
    return i8_nsplit(batch, nsteps, std::max<uint32_t>(1u, slots), 0);
}

// HOST-ONLY tap (no device needed): the digit images and parameters build_i8_image() produces for a table.  Returns the
// images' size in bytes (also when `out` is NULL or too small: nothing is written then), 0 when the table has no image.
// params[0 .. 6] = level weights, [7] = 2^54, [8] = T, [9] = E5, [10] = allowance of the refined form, [11] = 5, [12] = 7,
// [13] = E4, [14] = T4 (as the float the kernel compares with).
size_t baz_music_debug_i8_image(uint32_t m, uint32_t resolution, const float* table_ri, uint8_t* out, size_t out_bytes,
                                double* params)
{
    if (!table_ri || m < 6 || m > BAZ_MUSIC_FAST_M || resolution == 0) return 0;
    std::vector<double> F;
    build_F(table_ri, m, resolution, F);
    std::vector<uint8_t> img;
    I8Params ip = {};
    if (!build_i8_image(F, m, resolution, (resolution + 63) / 64, img, ip)) return 0;
    if (params) {
        for (int l = 0; l < I8_ND; ++l) params[l] = ip.wt[l];
        params[7] = ip.sq; params[8] = ip.t_acc; params[9] = ip.e_bound; params[10] = ip.e_refined;
        params[11] = (double)I8_NS; params[12] = (double)I8_ND;
        params[13] = ip.e4_bound; params[14] = (double)ip.t4_f;
    }
    if (out && out_bytes >= img.size()) std::memcpy(out, img.data(), img.size());
    return img.size();
}

// ---- the table images: what the device builders produced (active set) against the former host builders -------------------
// which: 0 FB, 1 TB, 2 coarse (C then X), 3 int8 digits (0 .. 4 then 5, 6), 4 ||a||^2 padded, 5 transposed table (m > 16),
// 6 ||a||^2 plain (m > 16), 7 the parameters as BAZ_MUSIC_TABLE_NPARAMS doubles.  Returns the image's size in bytes (0: this
// configuration has no such image); written when out_bytes suffices.
namespace {
constexpr int TABLE_NPARAMS = 22;
void pack_table_params(const TableSet& T, bool has_cs, bool has_i8, double* p)
{
    for (int i = 0; i < TABLE_NPARAMS; ++i) p[i] = 0.0;
    p[0] = T.refine_below;
    if (has_cs && T.cs_ok) { p[1] = 1.0; p[2] = T.cs.sc_up; p[3] = T.cs.es_factor; p[4] = T.cs.sc; p[5] = T.cs.fmax; }
    if (has_i8 && T.i8_ok) {
        p[6] = 1.0;
        for (int l = 0; l < I8_ND; ++l) p[7 + l] = T.i8.wt[l];
        p[14] = T.i8.sq; p[15] = T.i8.t_acc; p[16] = T.i8.e_bound; p[17] = T.i8.e_refined; p[18] = T.i8.ws_f;
        p[19] = T.i8.t_acc_f; p[20] = T.i8.t4_f; p[21] = T.i8.e4_bound;
    }
}
}  // namespace

size_t baz_music_debug_table_image(baz_music_ctx* c, int which, void* out, size_t out_bytes)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    const size_t pad_steps = (size_t)c->fb_steps + 2;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = c->dFB; bytes = pad_steps * c->fb_step_elems * sizeof(double2); break;
        case 1: src = c->dTB; bytes = pad_steps * c->tb_step_elems * sizeof(double2); break;
        case 2: src = c->dCS; bytes = c->dCS ? coarse_image_bytes(c) : 0; break;
        case 3: src = c->dIB; bytes = c->dIB ? i8_image_bytes(c->m, c->fb_steps) : 0; break;
        case 4: src = c->dA2p; bytes = pad_steps * 64 * sizeof(double); break;
        case 5: src = c->dTA; bytes = (size_t)c->m * c->res * sizeof(float2); break;
        case 6: src = c->dA2; bytes = (size_t)c->res * sizeof(double); break;
#ifdef BAZ_MUSIC_LAB
        case 8: src = c->dIP; bytes = c->dIP ? i8p_image_bytes(c->fb_steps) : 0; break;
#endif
        case 7: {
            double p[TABLE_NPARAMS];
#ifdef BAZ_MUSIC_LAB
            const bool any_i8 = c->dIB != nullptr || c->dIP != nullptr;
#else
            const bool any_i8 = c->dIB != nullptr;
#endif
            pack_table_params(active_table_set(c), c->dCS != nullptr, any_i8, p);
            if (out && out_bytes >= sizeof(p)) std::memcpy(out, p, sizeof(p));
            return sizeof(p);
        }
        default: return 0;
    }
    if (!src) return 0;
    // an image whose parameters could not be formed is not built (its bytes are whatever the buffer held): report it as absent
    if ((which == 2 && !c->cs_ok) || ((which == 3 || which == 8) && !c->i8_ok)) return 0;
    if (out && out_bytes >= bytes) {
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
    }
    return bytes;
}

// HOST-ONLY (no device needed): the same image from the round-4 host builders -- the checker of the device builders.
size_t baz_music_debug_host_table_image(uint32_t m, uint32_t n, uint32_t resolution, const float* table_ri, int which, void* out,
                                        size_t out_bytes)
{
    if (!table_ri || m == 0 || n == 0 || n >= m || m > BAZ_MUSIC_MAX_M || resolution == 0) return 0;
    const bool wide = m > BAZ_MUSIC_FAST_M;
    const uint32_t res = resolution, steps = (res + 63) / 64;
    std::vector<uint8_t> bytes;
    auto from_doubles = [&](const std::vector<double>& v) {
        bytes.resize(v.size() * sizeof(double));
        std::memcpy(bytes.data(), v.data(), bytes.size());
    };
    auto a2_of = [&](uint32_t b) { return baztab::tab_a2(table_ri + 2 * (size_t)b * m, m); };   // the device builders' own routine (contraction off)
    const bool has_cs = !wide && m <= 8;
    const bool has_i8 = !wide && m >= 6 && n <= 4 && i8_image_bytes(m, steps) <= I8_IMAGE_LIMIT;
    [[maybe_unused]] const bool has_i8p = m <= 4 && n <= 4;
    const bool has_a2p = wide ? (n <= 8) : short_form_applies(m, n);
    const bool has_tb = wide ? (n <= 8) : true;
    std::vector<double> F;
    if (!wide && (which == 0 || which == 2 || which == 3 || which == 7 || which == 8)) build_F(table_ri, m, res, F);
    switch (which) {
        case 0: {
            if (wide) return 0;
            std::vector<double> FB;
            build_FB(F, m, res, steps, FB);
            from_doubles(FB);
            break;
        }
        case 1: {
            if (!has_tb) return 0;
            std::vector<double> TB;
            build_TB(table_ri, m, res, steps, TB);
            from_doubles(TB);
            break;
        }
        case 2: {
            if (!has_cs) return 0;
            CoarseParams cp = {0.0f, 0.0f, 0.0, 0.0, 1};
            if (!build_coarse_image(F, m, res, round_up((res + 15) / 16, 8), bytes, cp)) return 0;
            break;
        }
        case 3: {
            if (!has_i8) return 0;
            I8Params ip = {};
            if (!build_i8_image(F, m, res, steps, bytes, ip)) return 0;
            break;
        }
#ifdef BAZ_MUSIC_LAB
        case 8: {
            if (!has_i8p) return 0;
            I8Params ip = {};
            if (!build_i8p_image(F, m, res, steps, bytes, ip)) return 0;
            break;
        }
#endif
        case 4: {
            if (!has_a2p) return 0;
            std::vector<double> a2((size_t)(steps + 2) * 64, 1e300);
            for (uint32_t b = 0; b < res; ++b) a2[64 + b] = a2_of(b);
            from_doubles(a2);
            break;
        }
        case 5: {
            if (!wide) return 0;
            std::vector<float> ta((size_t)m * res * 2);
            for (uint32_t b = 0; b < res; ++b)
                for (uint32_t i = 0; i < m; ++i) {
                    ta[2 * ((size_t)i * res + b)] = table_ri[2 * ((size_t)b * m + i)];
                    ta[2 * ((size_t)i * res + b) + 1] = table_ri[2 * ((size_t)b * m + i) + 1];
                }
            bytes.resize(ta.size() * sizeof(float));
            std::memcpy(bytes.data(), ta.data(), bytes.size());
            break;
        }
        case 6: {
            if (!wide) return 0;
            std::vector<double> a2(res);
            for (uint32_t b = 0; b < res; ++b) a2[b] = a2_of(b);
            from_doubles(a2);
            break;
        }
        case 7: {
            TableSet T;
            double amax2 = 0.0;
            for (uint32_t b = 0; b < res; ++b) {
                const double a2 = a2_of(b);
                if (a2 > amax2 && a2 < 1e300) amax2 = a2;
            }
            T.refine_below = amax2 * (double)m * 1e-8;
            std::vector<uint8_t> scratch;
            if (has_cs) T.cs_ok = build_coarse_image(F, m, res, round_up((res + 15) / 16, 8), scratch, T.cs);
            if (has_i8) T.i8_ok = build_i8_image(F, m, res, steps, scratch, T.i8);
            // (the packed operands of m <= 4 exist in lab contexts only -- BAZ_MUSIC_I8P=1 --: their parameters are not part of this image)
            std::vector<double> p(TABLE_NPARAMS);
            pack_table_params(T, has_cs, has_i8, p.data());
            from_doubles(p);
            break;
        }
        default: return 0;
    }
    if (out && out_bytes >= bytes.size()) std::memcpy(out, bytes.data(), bytes.size());
    return bytes.size();
}

// The gated scan's sorting policy: out[0] / out[1] = gated-scan launches that sorted / did not sort their items first, out[2] / out[3] = exact
// evaluations and (row group, tile) pairs walked of the most recent FINISHED gated launch, out[4] = 1 if that launch was sorted.  Synchronises.
int baz_music_debug_sort_state(baz_music_ctx* c, uint64_t out[5])
{
    if (!c || !out) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
#ifdef BAZ_MUSIC_LAB
    out[0] = c->sorted_calls; out[1] = c->unsorted_calls;
    out[2] = c->hFire ? c->hFire[0] : 0; out[3] = c->hFire ? c->hFire[1] : 0; out[4] = c->hFire ? (c->hFire[2] & 1ull) : 0;
#else
    for (int k = 0; k < 5; ++k) out[k] = 0;               // the release library never sorts and keeps no fire statistic
#endif
    return BAZ_MUSIC_OK;
}

int baz_music_debug_evd(baz_music_ctx* c, const void* d_R, uint32_t batch, void* d_Q)
{
    if (!c || !d_R || !d_Q || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    if (c->wide) return BAZ_MUSIC_E_UNSUPPORTED;
    return launch_evd(c, static_cast<const double2*>(d_R), batch, static_cast<double*>(d_Q), baz_music_q_stride(batch), nullptr);
}

uint64_t baz_music_bytes_per_item(const baz_music_ctx* c, int with_spectrum)
{
    if (!c) return 0;
    return 8ull * c->nsamples + 8ull * c->n + (with_spectrum ? 4ull * c->res : 0ull);
}

// Lab build with BAZ_MUSIC_GUARD=1: checks the guard zones of every live device buffer of this library (synchronises the device) and returns
// the number of zones found damaged since the process started (at this check or at an earlier free); 0 when the guard is off / in the release build.
int baz_music_debug_guard_check(void)
{
#ifdef BAZ_MUSIC_LAB
    if (!guard_on()) return 0;
    std::lock_guard<std::mutex> lk(g_guard_mtx);
    unsigned long long now = 0;
    for (const auto& g : g_guards) now += (unsigned long long)guard_check_one(g.first, g.second);
    return (int)std::min<unsigned long long>(g_guard_damaged + now, 0x7FFFFFFFull);
#else
    return 0;
#endif
}

// 1 when this library allocates with guard zones (lab build, BAZ_MUSIC_GUARD=1), else 0
int baz_music_debug_guard_active(void)
{
#ifdef BAZ_MUSIC_LAB
    return guard_on() ? 1 : 0;
#else
    return 0;
#endif
}

const char* baz_music_strerror(int code)
{
    switch (code) {
        case BAZ_MUSIC_OK: return "ok";
        case BAZ_MUSIC_E_INVALID: return "invalid argument";
        case BAZ_MUSIC_E_NOMEM: return "out of memory";
        case BAZ_MUSIC_E_HIP: return "HIP runtime error";
        case BAZ_MUSIC_E_UNSUPPORTED:
            return "configuration not supported by the gfx950 kernels (limits: m <= 64 antennas -- specialised kernels up "
                   "to 16, run-time-m kernels from 17 to 64 --, n < m emitters, resolution <= 1048576 bins, the local-maximum "
                   "picker only up to 16 antennas; the reference itself has none, lib/baz_music_doa.cc:45-50)";
        case BAZ_MUSIC_E_NODEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

const char* baz_music_last_hip_error(const baz_music_ctx* c) { return c ? c->hip_err : ""; }

const char* baz_music_version(void) { return "gr_baz_amd/baz_music_hip 0.1 (gfx950)"; }

int baz_music_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int usable = 0;
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
        else break;   // device ids must stay dense: stop at the first foreign device
    }
    return usable;
}

int baz_music_set_peak_mode(baz_music_ctx* c, int mode)
{
    if (!c || (mode != 0 && mode != 1)) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    if (mode && c->wide) return BAZ_MUSIC_E_UNSUPPORTED;   // the opt-in picker exists for the specialised kernels only
    c->peak_mode = mode;
    return BAZ_MUSIC_OK;
}

int64_t baz_music_refined_items(baz_music_ctx* c) { return baz_music_refined_values(c); }

int64_t baz_music_refined_values(baz_music_ctx* c)
{
    if (!c) return -1;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    if (!c->dRefined) return 0;
    unsigned long long n = 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(&n, c->dRefined + c->stat_parity, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)n;
}

int baz_music_device(const baz_music_ctx* c) { return c ? c->device : BAZ_MUSIC_E_INVALID; }

}  // extern "C"

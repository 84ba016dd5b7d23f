// Lab probe (not part of the product): operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950,
// including the CBSZ/ABID block broadcast of the A operand, found by brute force with one-hot operands.
// For every pair (la, lb): A = 1.0 in lane la, B = 1.0 in lane lb, C = 0 -> which result lanes are non-zero.
// Output: one line per (cbsz, abid) with the inferred bit-field mapping, plus the raw table as JSON.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/probe_mfma4 scripts/probe_mfma4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CBSZ, int ABID>
__global__ void probe(double* out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = (lane == la) ? 1.0 : 0.0;
            const double b = (lane == lb) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
            out[((size_t)la * 64 + lb) * 64 + lane] = d;
        }
}

static int f(int lane, int field) { return (lane >> (2 * field)) & 3; }

template <int CBSZ, int ABID>
int run(FILE* js, bool first)
{
    double* d_out = nullptr;
    const size_t n = 64 * 64 * 64;
    if (hipMalloc((void**)&d_out, n * sizeof(double)) != hipSuccess) return 1;
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d_out);
    std::vector<double> h(n);
    if (hipMemcpy(h.data(), d_out, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    (void)hipFree(d_out);
    // raw: for each (la, lb) the list of non-zero result lanes
    fprintf(js, "%s\"cbsz%d_abid%d\": [", first ? "" : ",\n", CBSZ, ABID);
    int total = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            int cnt = 0, lane0 = -1;
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) { if (!cnt) lane0 = l; ++cnt; }
            fprintf(js, "%s[%d,%d]", (la || lb) ? "," : "", lane0, cnt);
            total += cnt;
        }
    fprintf(js, "]");
    // hypothesis search: A lane fields (i,k,blk) = perm pa of (f0,f1,f2); B (j,k,blk) = perm pb; D (i->?, j->?, blk->?) = perm pd
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    int found = 0;
    for (int pa = 0; pa < 6; ++pa)
        for (int pb = 0; pb < 6; ++pb)
            for (int pd = 0; pd < 6; ++pd) {
                bool ok = true;
                for (int la = 0; la < 64 && ok; ++la)
                    for (int lb = 0; lb < 64 && ok; ++lb) {
                        const int ai = f(la, perms[pa][0]), ak = f(la, perms[pa][1]), ab = f(la, perms[pa][2]);
                        const int bj = f(lb, perms[pb][0]), bk = f(lb, perms[pb][1]), bb = f(lb, perms[pb][2]);
                        // with CBSZ: result block r uses A block ((r >> CBSZ) << CBSZ) + ABID and B block r
                        for (int l = 0; l < 64 && ok; ++l) {
                            const int di = f(l, perms[pd][0]), dj = f(l, perms[pd][1]), dbk = f(l, perms[pd][2]);
                            const int a_src_blk = CBSZ ? (((dbk >> CBSZ) << CBSZ) + ABID) : dbk;
                            const bool expect = (ak == bk) && (bb == dbk) && (ab == a_src_blk) && (ai == di) && (bj == dj);
                            const bool got = h[((size_t)la * 64 + lb) * 64 + l] != 0.0;
                            if (expect != got) ok = false;
                        }
                    }
                if (ok) {
                    printf("cbsz=%d abid=%d: A lane fields (i,k,blk)=f%d,f%d,f%d  B (j,k,blk)=f%d,f%d,f%d  D (i,j,blk)=f%d,f%d,f%d   [f0=lane&3 f1=(lane>>2)&3 f2=lane>>4]\n",
                           CBSZ, ABID, perms[pa][0], perms[pa][1], perms[pa][2], perms[pb][0], perms[pb][1], perms[pb][2],
                           perms[pd][0], perms[pd][1], perms[pd][2]);
                    ++found;
                }
            }
    if (!found) printf("cbsz=%d abid=%d: no bit-field hypothesis matches (total non-zeros %d) -- see raw JSON\n", CBSZ, ABID, total);
    return 0;
}

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "probe_mfma4.json";
    FILE* js = fopen(path, "w");
    if (!js) return 1;
    fprintf(js, "{");
    int rc = 0;
    rc |= run<0, 0>(js, true);
    rc |= run<1, 0>(js, false);
    rc |= run<1, 1>(js, false);
    rc |= run<2, 0>(js, false);
    rc |= run<2, 1>(js, false);
    rc |= run<2, 3>(js, false);
    fprintf(js, "}\n");
    fclose(js);
    return rc;
}

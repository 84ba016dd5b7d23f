// ORACLE-ONLY API SHIM: the handful of pmt calls in /root/reference/lib/baz_fractional_resampler_cc.cc:99-139
// (a double, or a pair (long . double)).
#ifndef BAZ_ORACLE_PMT_SHIM
#define BAZ_ORACLE_PMT_SHIM
#include <memory>
#include <stdexcept>
#include <string>
namespace pmt {
struct pmt_base {
    int kind;            // 0 symbol, 1 double, 2 long, 3 pair
    double d; long l; std::string s;
    std::shared_ptr<pmt_base> a, b;
};
typedef std::shared_ptr<pmt_base> pmt_t;
inline pmt_t mp(const char* s) { pmt_t p(new pmt_base()); p->kind = 0; p->s = s; return p; }
inline pmt_t from_double(double d) { pmt_t p(new pmt_base()); p->kind = 1; p->d = d; return p; }
inline pmt_t from_long(long l) { pmt_t p(new pmt_base()); p->kind = 2; p->l = l; return p; }
inline pmt_t cons(pmt_t a, pmt_t b) { pmt_t p(new pmt_base()); p->kind = 3; p->a = a; p->b = b; return p; }
inline bool is_pair(pmt_t p) { return p && p->kind == 3; }
inline pmt_t car(pmt_t p) { return p->a; }
inline pmt_t cdr(pmt_t p) { return p->b; }
inline long to_long(pmt_t p) { if (!p || p->kind != 2) throw std::runtime_error("pmt: not a long"); return p->l; }
inline double to_double(pmt_t p)
{
    if (p && p->kind == 1) return p->d;
    if (p && p->kind == 2) return (double)p->l;
    throw std::runtime_error("pmt: not a number");
}
}  // namespace pmt
#endif

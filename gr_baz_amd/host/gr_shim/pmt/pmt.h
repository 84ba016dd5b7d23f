/* GNU Radio 3.7 API stand-in: the handful of PMT calls the fractional resampler's "msg" port touches
 * (/root/reference/lib/baz_fractional_resampler_cc.cc:104-139: mp, is_pair, car, cdr, to_long, to_double; cons /
 * from_long / from_double to build test messages).  Used ONLY where GNU Radio is not installed. */
#ifndef GR_BAZ_AMD_SHIM_PMT_H
#define GR_BAZ_AMD_SHIM_PMT_H

#include <memory>
#include <stdexcept>
#include <string>

namespace pmt {

class wrong_type : public std::invalid_argument {
public:
    explicit wrong_type(const std::string& what) : std::invalid_argument(what) {}
};

struct pmt_base {
    enum kind_t { SYMBOL, LONG, DOUBLE, PAIR } kind;
    std::string sym;
    long l;
    double d;
    std::shared_ptr<pmt_base> a, b;
    explicit pmt_base(kind_t k) : kind(k), l(0), d(0.0) {}
};
typedef std::shared_ptr<pmt_base> pmt_t;

inline pmt_t mp(const std::string& s) { pmt_t p(new pmt_base(pmt_base::SYMBOL)); p->sym = s; return p; }
inline pmt_t intern(const std::string& s) { return mp(s); }
inline pmt_t from_long(long v) { pmt_t p(new pmt_base(pmt_base::LONG)); p->l = v; return p; }
inline pmt_t from_double(double v) { pmt_t p(new pmt_base(pmt_base::DOUBLE)); p->d = v; return p; }
inline pmt_t cons(const pmt_t& x, const pmt_t& y) { pmt_t p(new pmt_base(pmt_base::PAIR)); p->a = x; p->b = y; return p; }
inline bool is_symbol(const pmt_t& p) { return p && p->kind == pmt_base::SYMBOL; }
inline bool is_pair(const pmt_t& p) { return p && p->kind == pmt_base::PAIR; }
inline pmt_t car(const pmt_t& p) { if (!is_pair(p)) throw wrong_type("pmt::car"); return p->a; }
inline pmt_t cdr(const pmt_t& p) { if (!is_pair(p)) throw wrong_type("pmt::cdr"); return p->b; }
inline std::string symbol_to_string(const pmt_t& p) { if (!is_symbol(p)) throw wrong_type("pmt::symbol_to_string"); return p->sym; }
inline long to_long(const pmt_t& p) { if (!p || p->kind != pmt_base::LONG) throw wrong_type("pmt::to_long"); return p->l; }
inline double to_double(const pmt_t& p)       /* like GNU Radio: reals and integers convert */
{
    if (p && p->kind == pmt_base::DOUBLE) return p->d;
    if (p && p->kind == pmt_base::LONG) return (double)p->l;
    throw wrong_type("pmt::to_double");
}

}  // namespace pmt
#endif

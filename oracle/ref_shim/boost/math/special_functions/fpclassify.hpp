// ORACLE-ONLY API SHIM: boost::math::isfinite / isnan / isinf as used by the (dead) tail of
// /root/reference/lib/baz_agc_cc.cc:104-106.
#ifndef BAZ_ORACLE_BOOST_FPCLASSIFY_SHIM
#define BAZ_ORACLE_BOOST_FPCLASSIFY_SHIM
#include <cmath>
namespace boost { namespace math {
template <class T> inline bool isfinite(T v) { return std::isfinite(v); }
template <class T> inline bool isnan(T v) { return std::isnan(v); }
template <class T> inline bool isinf(T v) { return std::isinf(v); }
} }
#endif

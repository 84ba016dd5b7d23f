/* GNU Radio 3.7 API stand-in: the item/pointer-vector typedefs and boost::shared_ptr. */
#ifndef GR_BAZ_AMD_SHIM_TYPES_H
#define GR_BAZ_AMD_SHIM_TYPES_H

#include <complex>
#include <memory>
#include <vector>

/* GNU Radio 3.7 hands out boost::shared_ptr; the shim maps it onto std::shared_ptr. */
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
}

typedef std::complex<float> gr_complex;
typedef std::complex<double> gr_complexd;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
typedef std::vector<int> gr_vector_int;

#endif

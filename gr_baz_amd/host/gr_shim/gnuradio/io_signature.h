/* GNU Radio 3.7 API stand-in (see sync_block.h in this directory). */
#ifndef GR_BAZ_AMD_SHIM_IO_SIGNATURE_H
#define GR_BAZ_AMD_SHIM_IO_SIGNATURE_H

#include <gnuradio/types.h>

#include <stdexcept>
#include <vector>

namespace gr {

class io_signature {
public:
    typedef boost::shared_ptr<io_signature> sptr;
    static const int IO_INFINITE = -1;

    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return sptr(new io_signature(min_streams, max_streams, std::vector<int>(1, sizeof_stream_item)));
    }
    static sptr make2(int min_streams, int max_streams, int s1, int s2)
    {
        std::vector<int> v; v.push_back(s1); v.push_back(s2);
        return sptr(new io_signature(min_streams, max_streams, v));
    }
    static sptr make3(int min_streams, int max_streams, int s1, int s2, int s3)
    {
        std::vector<int> v; v.push_back(s1); v.push_back(s2); v.push_back(s3);
        return sptr(new io_signature(min_streams, max_streams, v));
    }

    int min_streams() const { return d_min_streams; }
    int max_streams() const { return d_max_streams; }
    /* like GNU Radio: streams past the listed sizes reuse the last size */
    int sizeof_stream_item(int index) const
    {
        if (index < 0) throw std::invalid_argument("gr::io_signature::sizeof_stream_item");
        size_t i = (size_t)index < d_sizes.size() ? (size_t)index : d_sizes.size() - 1;
        return d_sizes[i];
    }
    std::vector<int> sizeof_stream_items() const { return d_sizes; }

private:
    io_signature(int mn, int mx, const std::vector<int>& sizes) : d_min_streams(mn), d_max_streams(mx), d_sizes(sizes)
    {
        if (mn < 0 || (mx != IO_INFINITE && mx < mn) || sizes.empty()) throw std::invalid_argument("gr::io_signature");
    }
    int d_min_streams, d_max_streams;
    std::vector<int> d_sizes;
};

}  // namespace gr
#endif

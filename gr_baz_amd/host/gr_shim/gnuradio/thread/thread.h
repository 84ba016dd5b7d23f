/* GNU Radio 3.7 API stand-in: gr::thread::mutex / scoped_lock (boost::mutex there). */
#ifndef GR_BAZ_AMD_SHIM_THREAD_H
#define GR_BAZ_AMD_SHIM_THREAD_H
#include <mutex>
namespace gr {
namespace thread {
typedef std::mutex mutex;
typedef std::unique_lock<std::mutex> scoped_lock;
}  // namespace thread
}  // namespace gr
#endif

// ORACLE-ONLY API SHIM: gr::thread::mutex / scoped_lock (boost::mutex in GNU Radio 3.7).
#ifndef BAZ_ORACLE_GR_THREAD_SHIM
#define BAZ_ORACLE_GR_THREAD_SHIM
#include <mutex>
namespace gr { namespace thread {
typedef std::mutex mutex;
typedef std::unique_lock<std::mutex> scoped_lock;
} }
#endif

// ORACLE-ONLY API SHIM: gr::io_signature lives in the sync_block shim.
#include <gnuradio/sync_block.h>

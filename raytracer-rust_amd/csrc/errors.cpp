#include "errors.h"
namespace rte {
namespace { thread_local std::string g_err; }
void set_error(const std::string& m) { g_err = m; }
const char* last_error() { return g_err.c_str(); }
}

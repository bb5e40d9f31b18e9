// errors.h -- the per-thread Err(String) behind rt_last_error().
#ifndef RT_ERRORS_H
#define RT_ERRORS_H
#include <string>
namespace rte {
void set_error(const std::string& m);
const char* last_error();
}
#endif

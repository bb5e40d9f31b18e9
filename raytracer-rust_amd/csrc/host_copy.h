// host_copy.h -- memcpy on all host threads (one thread moves about a third of what the memory system can).
#ifndef RT_HOST_COPY_H
#define RT_HOST_COPY_H
#include <stddef.h>
namespace rtu {
void parallel_copy(void* dst, const void* src, size_t bytes);
}
#endif

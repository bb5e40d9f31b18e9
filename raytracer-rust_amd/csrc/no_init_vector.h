// no_init_vector.h -- std::vector whose resize leaves plain-data elements unwritten.
#ifndef RT_NO_INIT_VECTOR_H
#define RT_NO_INIT_VECTOR_H

#include <memory>
#include <utility>
#include <vector>

namespace rtu {

// The big arrays of the scene packer and the tree builder are filled right after they are sized, by all
// threads; writing zeros into hundreds of megabytes first, from one thread, costs more than the fill.
template <class T> struct NoInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAllocator<U>; };
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using NoInitVector = std::vector<T, NoInitAllocator<T>>;

}  // namespace rtu
#endif

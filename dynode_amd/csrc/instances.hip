// instances.hip -- explicit kernel instantiations for translation unit DYN_TU
// (compile with -DDYN_TU=<n>, n in [0, DYN_NUM_TU)).  See instances.def.
#include "solve_kernel.hpp"

#ifndef DYN_TU
#error "compile with -DDYN_TU=<n>"
#endif

namespace dyn {
#define X(T, METHOD, G, S, E, WN, C, W, ND, SPL) \
    template hipError_t launch<T, METHOD, G, S, E, WN, C, W, ND, SPL>(const KArgs<T> &, hipStream_t);
#define XI(T, METHOD, G, S, E, WN, C, W, ND, SPL) \
    template hipError_t launch<T, METHOD, G, S, E, WN, C, W, ND, SPL, 1>(const KArgs<T> &, hipStream_t);
#define XF(T, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT) \
    template hipError_t launch<T, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT>(const KArgs<T> &, hipStream_t);
#include "instances.def"
#undef XF
#undef XI
#undef X
} // namespace dyn

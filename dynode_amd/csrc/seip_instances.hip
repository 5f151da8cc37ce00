// seip_instances.hip -- explicit SEIP kernel instantiations for translation unit SEIP_TU (see seip_instances.def).
#include "seip_kernel.hpp"

#ifndef SEIP_TU
#error "compile with -DSEIP_TU=<n>"
#endif

namespace dyn {
#define Y(T, METHOD, GA, L, K1, M1) template hipError_t launch_seip<T, METHOD, GA, L, K1, M1>(const KArgs<T> &, hipStream_t);
#define YT(T, METHOD, GA, L, K1, M1) template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, 2>(const KArgs<T> &, hipStream_t);
#define YW(T, METHOD, GA, L, K1, M1, KT, NW) template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, KT, NW>(const KArgs<T> &, hipStream_t);
#define YP(T, METHOD, GA, L, K1, M1, KT, NW) template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, KT, NW, 1>(const KArgs<T> &, hipStream_t);
#define YS(T, METHOD, GA, L, K1, M1, KT, NW) template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, KT, NW, 3>(const KArgs<T> &, hipStream_t);
#include "seip_instances.def"
#undef YS
#undef YP
#undef YW
#undef YT
#undef Y
} // namespace dyn

// agz_small_inst.hip — explicit instantiations of k_search_small (agz_search_small.hpp), compiled as seven translation
// units (-DAGZ_PART=0..7; parts 4-6: the variants with rows by legal rank, part 7: lane-groups of 4 and 2) so that the ~60 kernels build in parallel; agz_engine.hip declares them `extern template`.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/agz.h"
#include "agz_games.hpp"
#include "agz_device.hpp"
#include "agz_search_small.hpp"
#include "agz_search_big.hpp"
#include "agz_selfplay_small.hpp"
#include "agz_selfplay_big.hpp"

namespace agz {
#define X(F, C, K) AGZ_SMALL_VARIANTS(F, C, K, ) AGZ_BIG_VARIANTS(F, C, K, ) AGZ_PERSIST_VARIANTS(F, C, K, ) AGZ_PERSIST_BIG_VARIANTS(F, C, K, )
#if AGZ_PART == 0
AGZ_SMALL_SHAPES_0(X)
#elif AGZ_PART == 1
AGZ_SMALL_SHAPES_1(X)
#elif AGZ_PART == 2
AGZ_SMALL_SHAPES_2(X)
#elif AGZ_PART == 3
AGZ_SMALL_SHAPES_3(X)
#endif
#undef X
#if AGZ_PART == 7
#define X(F, C, K, R, GG) AGZ_SMALL_NARROW_VARIANTS(F, C, K, R, GG, )
AGZ_SMALL_NARROW_SHAPES(X)
#undef X
#define X(F, C, K, GG) AGZ_PERSIST_NARROW_VARIANTS(F, C, K, GG, )
AGZ_PERSIST_NARROW_SHAPES(X)
#undef X
#define X(F, C, K, R, GG) AGZ_SMALL_NARROW_SPARSE_VARIANTS(F, C, K, R, GG, )
AGZ_SMALL_NARROW_SPARSE_SHAPES(X)
#undef X
#elif AGZ_PART == 8
#define X(F, C, K4) AGZ_PERSIST_BIG4_VARIANTS(F, C, K4, )
X(F_LINE, 2, 24) X(F_HEX, 2, 24) X(F_REV, 1, 24)
#undef X
#elif AGZ_PART == 9
#define X(F, C, K4, R4) AGZ_BIG4_VARIANTS(F, C, K4, R4, )
X(F_LINE, 2, 24, 0) X(F_LINE, 2, 24, 16) X(F_LINE, 2, 24, 8) X(F_HEX, 2, 24, 0) X(F_HEX, 2, 24, 16) X(F_HEX, 2, 24, 8) X(F_REV, 1, 24, 0)
#undef X
#elif AGZ_PART == 10
#define X(F, C, K4) AGZ_PERSIST_BIG4_VARIANTS(F, C, K4, )
AGZ_PERSIST_BIG4_SHAPES_MORE(X)
#undef X
#elif AGZ_PART == 11
#define X(F, C, K4, R4) AGZ_BIG4_VARIANTS(F, C, K4, R4, )
AGZ_BIG4_SHAPES_MORE(X)
#undef X
#elif AGZ_PART >= 4
#define X(F, C, K, R) AGZ_SMALL_CMP_VARIANTS(F, C, K, R, ) AGZ_BIG_CMP_VARIANTS(F, C, K, R, )
#if AGZ_PART == 4
AGZ_SMALL_CMP_SHAPES_4(X)
#undef X
#define X(F, C, K, R) AGZ_PERSIST_AGE_VARIANTS(F, C, K, R, )
AGZ_PERSIST_AGE_SHAPES(X)
#elif AGZ_PART == 5
AGZ_SMALL_CMP_SHAPES_5(X)
#else
AGZ_SMALL_CMP_SHAPES_6(X)
#endif
#undef X
#endif
}  // namespace agz

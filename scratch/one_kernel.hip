// compile-time probe: only the 32768-game whole-search kernel (register usage / ISA experiments)
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../include/agz.h"
#include "../alphagpu_amd/csrc/agz_games.hpp"
#include "../alphagpu_amd/csrc/agz_device.hpp"
#include "../alphagpu_amd/csrc/agz_search_small.hpp"
namespace agz {
#ifndef WVV
#define WVV 4
#endif
template __global__ void k_search_small<F_LINE, 2, 12, 128, 4, WVV>(const SmallPar);
}
namespace agz { template __global__ void k_rollout_eager<F_LINE, 2, 12, WVV>(const TreePar); }
#include "../alphagpu_amd/csrc/agz_search_big.hpp"
namespace agz { template __global__ void k_search_big<F_LINE, 2, 12, 512, 1>(const BigSearchPar); template __global__ void k_search_big<F_LINE, 2, 12, 512, 2>(const BigSearchPar); }

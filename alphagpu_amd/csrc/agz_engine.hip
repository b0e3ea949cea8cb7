// agz_engine.hip — libagz: host side of the engine and the C ABI of include/agz.h.
//
// One engine = one device + one HIP stream.  The engine owns the trees (HBM), the network weights and the
// sample store; the per-move loop of mcts_single (mcts_gpu.jl:376-462) is V x { k_rollout, network } with no
// host synchronisation, and the per-generation loop of mcts() (:477-579) crosses PCIe with 4 bytes per ply.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "../../include/agz.h"
#include "agz_games.hpp"
#include "agz_device.hpp"
#include "agz_wave.hpp"
#include "agz_small_kernels.hpp"
#include "agz_tree_eager.hpp"
#include "agz_nn.hpp"
#include "agz_nn_wave.hpp"
#include "agz_nn_big.hpp"
#include "agz_search_small.hpp"
#include "agz_search_big.hpp"
#include "agz_selfplay.hpp"
#include "agz_selfplay_small.hpp"
#include "agz_selfplay_big.hpp"

using namespace agz;

static thread_local std::string g_create_error;

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) { (h)->fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return AGZ_ERR_HIP; } \
    } while (0)

struct DevNet {
    int H = 0, T = 0, in = 0, A = 0, INP = 0, AOP = 0;
    bool loaded = false;
    // exact mode: Flux-layout fp32
    float *W0 = nullptr, *Wres = nullptr, *Wp = nullptr, *bp = nullptr, *Wv = nullptr, *bv = nullptr;
    // bf16 mode: pre-tiled B fragments
    uint16_t *t0 = nullptr, *tres = nullptr, *thead = nullptr;
    uint16_t* w16w = nullptr;      // uniform k-rows for agz_nn_wave.hpp
    uint16_t* wbig = nullptr; int k0r = 0;   // hidden k-rows for agz_nn_big.hpp (layer 0 padded to k0r rows)
    uint16_t* w16 = nullptr;       // the same three sections tiled for v_mfma_f32_16x16x32_bf16 (the head section feeds agz_nn_big.hpp)
    float* bias_head = nullptr;
    int NT_h = 0, NT_head = 0;
};

// bytes of a row of the LDS hand-over window of the 128-wide one-launch kernels (one row per game: the leaf's planes on the way to the
// network — INP / 32 k-rows of 64 bytes + 16 bytes that spread the rows over the banks; the network skips the zero k-rows that pad layer 0 to
// whole groups, agz_nn_wave.hpp KR0 — and its AOP logits on the way back)
static int small_io_row_bytes(const DevNet& n) { return (std::max((n.INP / 32) * 64 + 16, 4 * n.AOP) + 15) & ~15; }

// The next-word tables of a whole-search workgroup (agz_tree_eager.hpp nxw: 16 bits per node of each of its games) go behind `shared` when the
// workgroup's share of the CU's LDS leaves them room next to a work list of at least 128 bytes per tree wave; returns their offset (0: none —
// the descent then reads the records) and moves `shared` past them.  AGZ_NXL=0 turns them off (A/B).
static int place_nxw(size_t& shared, size_t cu_lds, int tree_waves, int games_per_wave, int V) {
    static const bool on = AGZ_PERSIST_NXL && !(getenv("AGZ_NXL") && atoi(getenv("AGZ_NXL")) == 0);
    const size_t bytes = ((size_t)tree_waves * (size_t)games_per_wave * (size_t)V * 2 + 15) & ~(size_t)15;
    if (!on || V > 128 || shared + bytes + (size_t)tree_waves * 128 > cu_lds) return 0;   // (a table word holds the child id in 7 bits)
    const int off = (int)shared;
    shared += bytes;
    return off;
}

typedef void (*rollout_fn)(const TreePar);
typedef void (*small_fn)(const SmallPar);
typedef void (*big_fn)(const BigSearchPar);
typedef void (*persist_fn)(const PersistPar);
typedef void (*persist_big_fn)(const PersistBigPar);
namespace agz {
#define X(F, C, K) AGZ_SMALL_VARIANTS(F, C, K, extern) AGZ_BIG_VARIANTS(F, C, K, extern) AGZ_PERSIST_VARIANTS(F, C, K, extern) AGZ_PERSIST_BIG_VARIANTS(F, C, K, extern)
AGZ_SMALL_SHAPES(X)          // defined in agz_small_inst.hip
#undef X
#define X(F, C, K, GG) AGZ_PERSIST_NARROW_VARIANTS(F, C, K, GG, extern)
AGZ_PERSIST_NARROW_SHAPES(X)
#undef X
#define X(F, C, K, R) AGZ_PERSIST_AGE_VARIANTS(F, C, K, R, extern)
AGZ_PERSIST_AGE_SHAPES(X)
#undef X
#define X(F, C, K4) AGZ_PERSIST_BIG4_VARIANTS(F, C, K4, extern)
AGZ_PERSIST_BIG4_SHAPES(X)
#undef X
#define X(F, C, K4, R4) AGZ_BIG4_VARIANTS(F, C, K4, R4, extern)
AGZ_BIG4_SHAPES(X)
#undef X
#define X(F, C, K, R) AGZ_SMALL_CMP_VARIANTS(F, C, K, R, extern) AGZ_BIG_CMP_VARIANTS(F, C, K, R, extern)
AGZ_SMALL_CMP_SHAPES(X)
#undef X
#define X(F, C, K, R, GG) AGZ_SMALL_NARROW_VARIANTS(F, C, K, R, GG, extern)
AGZ_SMALL_NARROW_SHAPES(X)
#undef X
#define X(F, C, K, R, GG) AGZ_SMALL_NARROW_SPARSE_VARIANTS(F, C, K, R, GG, extern)
AGZ_SMALL_NARROW_SPARSE_SHAPES(X)
#undef X
}
typedef void (*advance_fn)(const PlyPar);
typedef void (*softmax_fn)(const float*, int, float*, int, int, int);

struct agz_engine {
    agz_config cfg;
    GamePar G;
    agz_game_info info;
    std::string err;
    hipStream_t stream = nullptr;
    static constexpr int KCH = 4;       // sub-batches of a search run as independent chains on parallel streams
    hipStream_t aux[KCH - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[KCH - 1] = {nullptr, nullptr, nullptr};
    int chains = 0;                     // 0 = automatic (AGZ_CHAINS overrides)
    int nn_wave_depth = 0;              // layers of weights in flight: 0 = by batch size (AGZ_NN_WAVE_DEPTH = 2, 4)
    int nn_wave_lt = 0;                 // 16-leaf tiles per workgroup of that kernel: 0 = by batch size (AGZ_NN_WAVE_LT = 1, 2, 4, 8)
    bool no_fastdiv = false;            // AGZ_NO_FASTDIV: the tree keeps the compiler's division everywhere (A/B, tests)
    bool no_fused_nn = false;           // AGZ_NO_FUSED_NN: one network launch per layer (k_layer_bf16) and the two-kernel search form (tests)
    int L = 0;                 // active slots
    int Lmax = 0, V = 0;
    TreePar tp;                // template of kernel arguments
    // tree memory
    uint8_t* recs = nullptr; Pos* states = nullptr; uint32_t* meta = nullptr;
    uint32_t *ncount = nullptr, *leaf = nullptr, *game_id = nullptr, *game_id2 = nullptr, *cnt_p = nullptr, *cnt_new = nullptr;
    uint32_t *slot_ply = nullptr, *slot_ply2 = nullptr;   // the ply of every slot's game (TreePar::slot_ply): filled with `step` by a plain search, kept per game by the ply loop
    bool in_ply_loop = false;
    // network i/o
    void* planes = nullptr; float* logits = nullptr; float *prior_eval = nullptr, *v_eval = nullptr, *policy_final = nullptr;
    uint16_t *act0 = nullptr, *act1 = nullptr; float *actf0 = nullptr, *actf1 = nullptr;
    int INP = 0, LGS = 0, Hcap = 0;
    DevNet net[2];
    // selfplay
    Pos* newpos = nullptr; uint32_t *alive = nullptr, *newslot = nullptr, *d_count = nullptr;
    int sample_games = 0;
    uint64_t* s_boards = nullptr; float* s_policy = nullptr; int16_t* s_move = nullptr; uint8_t* s_net = nullptr;
    uint32_t net_tag = 0;      // agz_set_network_tag: stored with every sample (which network searched the ply)
    int32_t* g_nplies = nullptr; int8_t* g_result = nullptr; Pos* g_final = nullptr;
    unsigned long long* d_stats = nullptr;
    unsigned long long* d_acc = nullptr;   // device-side sums of cnt_p / cnt_new over the instrumented plies of a generation
    int sp_games = 0;          // games of the last selfplay
    bool cnt_live = false;     // per-slot counters of the last search not yet folded into acc_*
    float* scratch_f = nullptr; // [Lmax][max(A,2VS)] getter staging
    // search state
    float cpuct = 1.5f; int training = 1; uint32_t step = 0; bool need_reset = true; bool injected = false;
    uint64_t total_rollouts = 0, acc_p = 0, acc_new = 0;
    // profiling
    bool step_last = false;    // stepwise API: the last select of the search has been launched
    bool prof_this = true; uint32_t search_seq = 0;   // whether the current search is instrumented (profiling bit 2 = sample every 4th)
    int profiling = 0;         // bit 0: HIP events around every tree-kernel launch, bit 1: around every network launch
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_tree, ev_nn;
    size_t ev_tree_used = 0, ev_nn_used = 0;
    double tree_ms = 0, nn_ms = 0, tree_busy_ms = 0; int64_t tree_launches = 0;
    hipEvent_t ev_ref = nullptr; bool ev_ref_live = false;
    hipEvent_t ev_ply0 = nullptr, ev_ply1 = nullptr; uint32_t* hcount = nullptr;   // ply loop: search timing, pinned alive count
    hipStream_t fold_stream = nullptr; hipEvent_t ev_fold = nullptr;   // ply loop with profiling: the descent counters of a search are folded beside the ply kernels, not in front of them
    hipEvent_t ev_adv = nullptr; bool ply_sleep = true;   // ply loop: the host thread SLEEPS (blocking event) until the ply's k_advance has run, then polls the scan's word for a few microseconds (AGZ_PLY_SPIN=1: spin all the way)
    unsigned long long* hflag = nullptr; unsigned long long* hflag_dev = nullptr; uint32_t ply_seq = 0;   // ... host-visible (seq, count) word the scan kernel publishes
    uint32_t* d_order = nullptr; int64_t sp_nsamples = 0; int sp_maxplies = 0;   // PoolSample order of the last generation (device), its length, its longest game
    uint8_t *stage_dev = nullptr, *stage_host = nullptr; size_t stage_cap = 0;     // agz_get_samples: packed records on the device / in pinned host memory (kept)
    uint64_t nn_leaves = 0;    // leaves sent through stand-alone network launches of the instrumented searches
    advance_fn k_adv = nullptr; softmax_fn k_soft = nullptr;
    big_fn k_big[2] = {nullptr, nullptr};   // whole-search kernel for 512-wide trunks (agz_search_big.hpp), 1 / 2 workgroups per CU
    int wl_lds_max = 1 << 30;    // cap (bytes per tree wave) of the LDS part of the work lists (AGZ_WL_LDS_BYTES; tests: the global overflow path)
    int big_mt = 0;              // != 0: force the leaf tiles per workgroup of the stand-alone wide-trunk network (AGZ_BIG_MT = 2, 4, 8)
    int big_maxl = 16384;        // ... used for batches up to this many games (AGZ_BIG_MAXL): 6.2 vs 8.0 ms per ply at 8192 games, 4.5 vs 7.3 at 1024, 9.9 vs 10.2 at 16384
    // the same kernels with node rows indexed by the root's legal rank (agz_tree_eager.hpp KPR_; Gobang / Hex 9x9: 8 instead of 12 entries
    // per lane), used by the ply loop once the roots cannot have more legal actions than the rows hold (legal_bound, set per ply)
    // (levels by entries per lane R: usable while no root has more than 8 R legal actions; 9x9 boards 8 / 4, 11x11 12 / 8 / 4, 13x13 16 / 8 / 4)
    struct CmpLevel { int kpr = 0; small_fn s2 = nullptr, s4[3] = {nullptr, nullptr, nullptr}, s8 = nullptr; big_fn b[2] = {nullptr, nullptr}, b8 = nullptr, b8x = nullptr; };
    CmpLevel cmp[4]; int ncmp = 0;
    // the one-launch search with NARROW lane-groups (agz_tree_eager.hpp G_: 4 or 2 lanes per tree, 16 / 32 trees per wave; workgroups of
    // four tree waves): g lanes per tree, kpl actions per lane, kpr rows per lane by legal rank (0: rows by action); k[0] / k[1]: register
    // budgets for two / one wave per SIMD.  narrow_mode (AGZ_NARROW): -1 never, 0 by batch size (from narrow_minl games on), 4 / 2: only
    // that group width (A/B).
    struct Narrow { int g = 0, kpl = 0, kpr = 0; small_fn k[2] = {nullptr, nullptr}; small_fn ksp = nullptr; };   // ksp: sparse waves (half the lane-groups hold a game), four waves per SIMD
    Narrow nar[8]; int nnar = 0, narrow_mode = 0, narrow_minl = -1, narrow_occ = -1;   // narrow_occ (AGZ_NARROW_OCC, tests): force the 2 (0) / 1 (1) waves-per-SIMD build
    // record geometry of the LAST search (the narrow builds of games with few actions lay their records out for their own row width):
    // what the root read-back kernels use
    uint32_t rd_rec_bytes = 0, rd_off_rk = 0, rd_off_el = 0, rd_off_vis = 0;
    // 64-game workgroups of eight waves (every wave a tree wave; the network body gives each one tile of neurons: half the weight
    // stream per game) for batches beyond 96 games per CU; tw8: -1 never, 1 wherever the batch allows, 0 (default) the shapes it was measured on
    small_fn k_small8 = nullptr; int tw8 = 0;
    big_fn k_big8 = nullptr, k_big8x = nullptr; int big8 = 0;   // k_big8x: two such workgroups per CU (128 registers) above 64 games per CU   // k_search_big with 64-game workgroups (eight tree waves, one workgroup per CU) above 32 games per CU: AGZ_BIG8
    int legal_bound = 1 << 30, tree_kpr = 0;
    // chained self-play calls (agz_selfplay_chain): the slots keep the games a call leaves in flight; chain_k0 = games handed to the earlier
    // calls of the chain (= the number of the next call's first game), chain_started = games started so far, chain_L = slots in flight
    // when the last call returned (which of a call's games are already over when it begins is read from their own entries)
    bool chain_live = false; unsigned long long chain_k0 = 0, chain_started = 0; int chain_L = 0;
    // the persistent self-play kernels (agz_selfplay_small.hpp): one launch per agz_selfplay / agz_selfplay_chain call, a workgroup keeps its
    // slots and loops over the plies of its games by itself.  persist: AGZ_PERSIST = 1 wherever a kernel exists (tests), 0 never, default
    // (-1): calls with refilled slots on an engine of more than 96 slots per CU.  chain_persist: the running chain's slots are not compacted.
    persist_fn k_persist = nullptr, k_persist_nar = nullptr, k_persist_nar_sp = nullptr; int persist_nar_g = 0, persist_nar_kpl = 0;
    int nar_sparse = 1;                // few-action games: 8 games per 16-group wave, four waves per SIMD (AGZ_NARROW_SPARSE=0: 16 games per wave, two per SIMD; 2: the searches at every batch size, tests)
    persist_big_fn k_persist_big[2] = {nullptr, nullptr};    // 512-wide trunks (agz_selfplay_big.hpp): one / two 64-game workgroups per CU
    int persist = -1; bool chain_persist = false; unsigned long long* d_pacc = nullptr;
    int reserve_slots = 0;             // AGZ_RESERVE_CUS x 128: slots the persistent launches leave without a workgroup (room for the exchange's RCCL kernels)
    // ... with age classes (workgroups that prefer old games run rows by legal rank; games migrate through a queue in device memory):
    // age_kpr rows per lane of the old body, age_on (AGZ_AGE=0 turns it off), age_old16 of 16 CU pairs prefer old games (AGZ_AGE_OLD16),
    // age_by_block (AGZ_AGE_CLASS=block, tests: odd workgroups prefer old games), age_backlog: the queue's length at which nothing is pushed
    // wide trunks: ONE 128-game workgroup per CU, 4 lanes per tree (k_selfplay_big4: a layer's weights stream once per 128 leaves); big4: AGZ_BIG4=1 wherever it
    // fits, 0 never; default: engines of more than 64 slots per CU (where the 8-lane form needs two 64-game workgroups per CU at 128 registers)
    persist_big_fn k_persist_big4 = nullptr; int big4_kpl = 0, big4 = -1;
    big_fn k_big4[3] = {nullptr, nullptr, nullptr}; int big4_kpr[3] = {0, 0, 0};   // k_search_big4: rows by action, then by legal rank (4-lane rows per lane = twice the 8-lane KPR)
    persist_fn k_persist_tw4 = nullptr, k_persist_tw4_age = nullptr; bool persist_tw4 = true;   // 32-game workgroups of four waves (default; AGZ_PERSIST_TW=8: 64-game workgroups of eight)
    persist_fn k_persist_age = nullptr; int age_kpr = 0; bool age_on = true, age_by_block = false, age_by_wave = false; int age_old16 = 8, age_backlog = 0;
    MigEntry* mq_buf = nullptr; unsigned long long* mq_ctr = nullptr; uint32_t mq_cap = 0; bool mq_dirty = false;
    int run_ahead = 8;                    // plies the ply loop may queue before it waits for a ply's counters (AGZ_RUN_AHEAD; while the pool cannot run dry)
    uint64_t age_ranked_searches = 0, age_searches = 0, age_pushed = 0;   // persistent form since the last agz_get_kernel_times(reset): game-searches with rows by rank / all / games migrated
    uint32_t sp_ring0 = 0, sp_k0 = 0;      // where the games of the last call sit in the per-game sample arrays / in the chain
    bool no_compact = false;            // AGZ_NO_COMPACT (A/B, tests)
    advance_fn k_spread = nullptr;      // policy_final rows from rank order back to action order after such a search
    small_fn k_small4[3] = {nullptr, nullptr, nullptr};   // the same with 32 games per workgroup, register budgets for 2 / 3 / 4 workgroups per SIMD set
    std::string form_tree, form_nn;   // kernels of the last search (agz_get_search_form)
    int small4_occ = -1;         // >= 0: force the register budget k_small4[occ] (AGZ_SMALL4_OCC = 0, 1, 2)
    int small4_maxl = 1 << 30;   // ... used for batches in (small_maxl, small4_maxl] that fit the chip at once (AGZ_SMALL4_MAXL)
    int cus = 256;
    int small_gpw = 0;           // games per tree wave of the 16-game variant: 0 = by batch size (AGZ_SMALL_GPW = 1, 2, 4, 8)
    small_fn k_small = nullptr; int small_maxl = 8192;   // 16-game workgroups of the whole-search kernel (helper waves take the backup items) up to small_maxl games (AGZ_SMALL_MAXL): 1.88 vs 2.04 ms per ply at 6144 games, 2.01 vs 2.18 at 7168, 2.14 vs 2.17 at 8192 against 32-game workgroups with sparse waves
    int reg3_max_waves = 0;      // largest grid the 3-waves-per-SIMD build of the stand-alone tree kernel is used for
    rollout_fn k_eager = nullptr, k_eager3 = nullptr;   // the tree kernel (agz_tree_eager.hpp), register budgets for 4 / 3 waves per SIMD
    uint32_t *wl = nullptr, *wl_n = nullptr, *sp = nullptr; uint32_t wl_cap = 0;
    size_t reg_lds = 0; int reg_kpl = 0;   // LDS of one tree wave; actions per lane (8 lanes per tree)
    uint32_t step_rollout = 0;   // stepwise API: rollout index of the last select

    int fail(const char* fmt, ...) {
        char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf; return 0;
    }
};

// ---- kernel dispatch tables -------------------------------------------------------------------------
// game shapes: (family, 64-bit words of a row of the board, 64-bit chunks of a bitboard) for the ply kernel; (family, chunks,
// actions per lane) for the tree kernels (agz_search_small.hpp AGZ_SMALL_SHAPES lists the same shapes for the whole-search form)
#define AGZ_COMBOS(X) \
    X(F_LINE, 1, 1) X(F_LINE, 2, 2) X(F_LINE, 3, 3) X(F_C4, 1, 1) \
    X(F_HEX, 1, 1) X(F_HEX, 1, 2) X(F_HEX, 2, 2) X(F_HEX, 2, 3) X(F_HEX, 3, 3) \
    X(F_REV, 1, 1) X(F_REV, 2, 1) AGZ_EXTRA_COMBOS(X)

static bool bind_kernels(agz_engine* h) {
    const GamePar& P = h->G;
#define X(F, R, C) if (P.fam == F && P.NR == R && P.NC == C) { h->k_adv = k_advance<F, R, C>; h->k_spread = k_spread_policy<F, R, C>; }
    AGZ_COMBOS(X)
#undef X
    // smallest block length KPL with 8*KPL >= A among the instantiated shapes
    const int kpl = P.A <= 32 ? 4 : (P.A <= 64 ? 8 : (P.A <= 96 ? 12 : (P.A <= 128 ? 16 : (P.A <= 192 ? 24 : 0))));
#define Z(F, C, K) if (P.fam == F && P.NC == C && kpl == K) { h->k_eager = k_rollout_eager<F, C, K, 4>; h->k_eager3 = k_rollout_eager<F, C, K, 3>; \
        h->k_small = k_search_small<F, C, K, 128, 2, 2>; h->k_small4[0] = k_search_small<F, C, K, 128, 4, 2>; h->k_small4[1] = k_search_small<F, C, K, 128, 4, 3>; \
        h->k_small4[2] = k_search_small<F, C, K, 128, 4, 4>; h->k_small8 = k_search_small<F, C, K, 128, 8, 4>; h->k_big[0] = k_search_big<F, C, K, 512, 1>; h->k_big[1] = k_search_big<F, C, K, 512, 2>; h->k_big8 = k_search_big<F, C, K, 512, 1, 0, 8>; h->k_big8x = k_search_big<F, C, K, 512, 2, 0, 8>; h->reg_kpl = K; \
        h->k_persist = k_selfplay_small<F, C, K, 128, 8, 4>; h->k_persist_tw4 = k_selfplay_small<F, C, K, 128, 4, 4>; h->k_persist_big[0] = k_selfplay_big<F, C, K, 512, 1>; h->k_persist_big[1] = k_selfplay_big<F, C, K, 512, 2>; }
    AGZ_SMALL_SHAPES(Z)
#undef Z
#define Z(F, C, K, GG) if (P.fam == F && P.NC == C && GG * K >= P.A && GG * K <= 8 * kpl) { h->k_persist_nar = k_selfplay_small<F, C, K, 128, 4, 2, GG>; h->k_persist_nar_sp = k_selfplay_small<F, C, K, 128, 4, 4, GG, 0, 32 / GG>; h->persist_nar_g = GG; h->persist_nar_kpl = K; }
    AGZ_PERSIST_NARROW_SHAPES(Z)
#undef Z
#define Z(F, C, K4) if (P.fam == F && P.NC == C && 2 * kpl == K4) { h->k_persist_big4 = k_selfplay_big4<F, C, K4, 512>; h->big4_kpl = K4; }
    AGZ_PERSIST_BIG4_SHAPES(Z)
#undef Z
#define Z(F, C, K4, R4) if (P.fam == F && P.NC == C && 2 * kpl == K4) { for (int i = 0; i < 3; ++i) if (!h->k_big4[i] && (i > 0) == (R4 > 0)) { h->k_big4[i] = k_search_big4<F, C, K4, 512, R4>; h->big4_kpr[i] = R4; break; } }
    AGZ_BIG4_SHAPES(Z)
#undef Z
#define Z(F, C, K, R) if (P.fam == F && P.NC == C && kpl == K) { h->k_persist_age = k_selfplay_small<F, C, K, 128, 8, 4, 8, R>; h->k_persist_tw4_age = k_selfplay_small<F, C, K, 128, 4, 4, 8, R>; h->age_kpr = R; }
    AGZ_PERSIST_AGE_SHAPES(Z)
#undef Z
#define Z(F, C, K, R) if (P.fam == F && P.NC == C && kpl == K && h->ncmp < 4) { agz_engine::CmpLevel& c = h->cmp[h->ncmp++]; c.kpr = R; \
        c.s2 = k_search_small<F, C, K, 128, 2, 2, R>; c.s4[0] = k_search_small<F, C, K, 128, 4, 2, R>; c.s4[1] = k_search_small<F, C, K, 128, 4, 3, R>; \
        c.s4[2] = k_search_small<F, C, K, 128, 4, 4, R>; c.s8 = k_search_small<F, C, K, 128, 8, 4, R>; c.b[0] = k_search_big<F, C, K, 512, 1, R>; c.b[1] = k_search_big<F, C, K, 512, 2, R>; c.b8 = k_search_big<F, C, K, 512, 1, R, 8>; c.b8x = k_search_big<F, C, K, 512, 2, R, 8>; }
    AGZ_SMALL_CMP_SHAPES(Z)
#undef Z
    // narrow lane-groups: the shape must hold the game's actions and its records must fit the allocation (rows no wider than the 8-lane rows)
#define Z(F, C, K, R, GG) if (P.fam == F && P.NC == C && GG * K >= P.A && GG * K <= 8 * kpl && h->nnar < 8) { agz_engine::Narrow& c = h->nar[h->nnar++]; \
        c.g = GG; c.kpl = K; c.kpr = R; c.k[0] = k_search_small<F, C, K, 128, 4, 2, R, GG>; c.k[1] = k_search_small<F, C, K, 128, 4, 1, R, GG>; }
    AGZ_SMALL_NARROW_SHAPES(Z)
#undef Z
#define Z(F, C, K, R, GG) for (int i = 0; i < h->nnar; ++i) if (P.fam == F && P.NC == C && h->nar[i].g == GG && h->nar[i].kpl == K && h->nar[i].kpr == R) h->nar[i].ksp = k_search_small<F, C, K, 128, 4, 4, R, GG, 32 / GG>;
    AGZ_SMALL_NARROW_SPARSE_SHAPES(Z)
#undef Z
    if (P.NR == 1) h->k_soft = k_softmax<1>; else if (P.NR == 2) h->k_soft = k_softmax<2>; else h->k_soft = k_softmax<3>;
    return h->k_adv != nullptr && h->k_eager != nullptr;
}

// ---- helpers ----------------------------------------------------------------------------------------
static inline uint16_t host_f2bf(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static int round_up(int x, int m) { return (x + m - 1) / m * m; }

// tile W (Flux (out,in) column-major, N x K) into MFMA B-operand fragments [KT][NT][64][8]
static void tile_weights(const float* W, int N, int K, int NT, int KT, std::vector<uint16_t>& out, int row_off = 0, int ldn = -1) {
    if (ldn < 0) ldn = N;
    for (int kt = 0; kt < KT; ++kt)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    int n = 32 * nt + (l & 31) - row_off, k = 16 * kt + 8 * (l >> 5) + j;
                    size_t o = (((size_t)kt * NT + nt) * 64 + l) * 8 + j;
                    if (n >= 0 && n < N && k < K) out[o] = host_f2bf(W[(size_t)n + (size_t)ldn * k]);
                }
}

// tile W (N x K, Flux layout) for the 16x16x32 MFMA: tile (kt, nt), lane l, element j = W[16 nt + (l & 15)][32 kt + 8 (l >> 4) + j]
static void tile_weights16(const float* W, int N, int K, int NT, int KT, uint16_t* out, int row_off = 0, int ldn = -1) {
    if (ldn < 0) ldn = N;
    for (int kt = 0; kt < KT; ++kt)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    int n = 16 * nt + (l & 15) - row_off, k = 32 * kt + 8 * (l >> 4) + j;
                    size_t o = (((size_t)kt * NT + nt) * 64 + l) * 8 + j;
                    if (n >= 0 && n < N && k < K) out[o] = host_f2bf(W[(size_t)n + (size_t)ldn * k]);
                }
}

template <typename T> static hipError_t dmalloc(T** p, size_t n) { return hipMalloc((void**)p, n * sizeof(T)); }

static void free_net(DevNet& n) {
    hipFree(n.W0); hipFree(n.Wres); hipFree(n.Wp); hipFree(n.bp); hipFree(n.Wv); hipFree(n.bv);
    hipFree(n.t0); hipFree(n.w16); hipFree(n.w16w); hipFree(n.wbig); n.wbig = nullptr; hipFree(n.bias_head);          // tres / thead point into t0's allocation
    n = DevNet();
}

// ======================================================================================================
extern "C" {

int agz_query_game(const agz_config* cfg, agz_game_info* out) {
    if (!cfg || !out) return AGZ_ERR_ARG;
    GamePar P;
    if (make_game_par(cfg->game, cfg->n, cfg->nvict, P) != 0) return AGZ_ERR_ARG;
    out->A = P.A; out->VS = P.VS; out->FS = P.FS; out->ML = P.ML; out->max_plies = P.max_plies;
    out->pos_image_bytes = (P.fam == F_REV) ? 152 : 104;
    out->rec_bytes = round_up(20 + 4 * P.A + 2 * P.VS + P.FS, 16);
    out->reserved = 0;
    return AGZ_OK;
}

const char* agz_last_error(const agz_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void agz_destroy(agz_engine* h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    hipFree(h->recs); hipFree(h->states); hipFree(h->meta); hipFree(h->ncount); hipFree(h->leaf); hipFree(h->game_id);
    hipFree(h->game_id2); hipFree(h->slot_ply); hipFree(h->slot_ply2); hipFree(h->cnt_p); hipFree(h->cnt_new); hipFree(h->planes); hipFree(h->logits);
    hipFree(h->prior_eval); hipFree(h->v_eval); hipFree(h->policy_final); hipFree(h->act0); hipFree(h->act1);
    hipFree(h->actf0); hipFree(h->actf1); hipFree(h->newpos); hipFree(h->alive); hipFree(h->newslot); hipFree(h->d_count);
    hipFree(h->s_boards); hipFree(h->s_policy); hipFree(h->s_move); hipFree(h->s_net); hipFree(h->g_nplies); hipFree(h->g_result);
    hipFree(h->g_final); hipFree(h->d_stats); hipFree(h->d_acc); hipFree(h->d_pacc); hipFree(h->mq_buf); hipFree(h->mq_ctr); hipFree(h->scratch_f);
    hipFree(h->wl); hipFree(h->wl_n); hipFree(h->sp);
    hipFree(h->d_order);
    hipFree(h->stage_dev); if (h->stage_host) hipHostFree(h->stage_host);
    free_net(h->net[0]); free_net(h->net[1]);
    for (auto& e : h->ev_tree) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : h->ev_nn) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (h->ev_ref) hipEventDestroy(h->ev_ref);
    if (h->ev_ply0) hipEventDestroy(h->ev_ply0);
    if (h->ev_ply1) hipEventDestroy(h->ev_ply1);
    if (h->ev_adv) hipEventDestroy(h->ev_adv);
    if (h->ev_fold) hipEventDestroy(h->ev_fold);
    if (h->fold_stream) hipStreamDestroy(h->fold_stream);
    if (h->hcount) hipHostFree(h->hcount);
    if (h->hflag) hipHostFree(h->hflag);
    for (int c = 0; c < agz_engine::KCH - 1; ++c) { if (h->aux[c]) hipStreamDestroy(h->aux[c]); if (h->ev_join[c]) hipEventDestroy(h->ev_join[c]); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int agz_create(const agz_config* cfg, agz_engine** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return AGZ_ERR_ARG; }
    *out = nullptr;
    agz_engine* h = new agz_engine();
    h->cfg = *cfg;
    auto bail = [&](int rc) { g_create_error = h->err; agz_destroy(h); return rc; };
    if (make_game_par(cfg->game, cfg->n, cfg->nvict, h->G) != 0) { h->fail("unsupported game parameters"); return bail(AGZ_ERR_ARG); }
    if (cfg->max_games < 1 || cfg->max_visits < 1 || cfg->max_visits > 256) { h->fail("max_games >= 1 and 1 <= max_visits <= 256 required"); return bail(AGZ_ERR_ARG); }
    if (cfg->nn_mode != AGZ_NN_BF16 && cfg->nn_mode != AGZ_NN_EXACT) { h->fail("bad nn_mode"); return bail(AGZ_ERR_ARG); }
    agz_query_game(cfg, &h->info);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { h->fail("no HIP device available (libagz has no CPU fallback)"); return bail(AGZ_ERR_HIP); }
    if (cfg->device < 0 || cfg->device >= ndev) { h->fail("device %d out of range (%d devices)", cfg->device, ndev); return bail(AGZ_ERR_ARG); }
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        h->fail("cannot initialise device %d", cfg->device); return bail(AGZ_ERR_HIP);
    }
    {
        const char* e = getenv("AGZ_CHAINS");
        h->chains = e ? atoi(e) : 0;
        h->no_fastdiv = getenv("AGZ_NO_FASTDIV") != nullptr;        // every switch is read here, once: no getenv on a launch path
        h->no_fused_nn = getenv("AGZ_NO_FUSED_NN") != nullptr;
        h->no_compact = getenv("AGZ_NO_COMPACT") != nullptr;
        { const char* e8 = getenv("AGZ_BIG8"); h->big8 = e8 ? (atoi(e8) > 0 ? 1 : -1) : 0; }
        { const char* e8 = getenv("AGZ_TW8"); h->tw8 = e8 ? (atoi(e8) > 0 ? 1 : -1) : 0; }
        e = getenv("AGZ_NN_WAVE_DEPTH");
        if (e && (atoi(e) == 2 || atoi(e) == 4)) h->nn_wave_depth = atoi(e);
        e = getenv("AGZ_NN_WAVE_LT");
        if (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4 || atoi(e) == 8)) h->nn_wave_lt = atoi(e);
        bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int c = 0; ok && c < agz_engine::KCH - 1; ++c)
            ok = hipStreamCreateWithFlags(&h->aux[c], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&h->ev_join[c], hipEventDisableTiming) == hipSuccess;
        if (!ok) { h->fail("cannot create the sub-batch streams"); return bail(AGZ_ERR_HIP); }
    }
    const GamePar& P = h->G;
    h->Lmax = cfg->max_games; h->V = cfg->max_visits;
    if (!bind_kernels(h)) { h->fail("no kernel instantiation for this game shape"); return bail(AGZ_ERR_UNSUPPORTED); }
    h->ply_sleep = getenv("AGZ_PLY_SPIN") == nullptr;
    { const char* ra = getenv("AGZ_RUN_AHEAD"); if (ra && atoi(ra) >= 0 && atoi(ra) <= 64) h->run_ahead = atoi(ra); }
    if (hipStreamCreateWithFlags(&h->fold_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fold, hipEventDisableTiming) != hipSuccess) {
        h->fold_stream = nullptr; h->ev_fold = nullptr;
    }
    if (hipEventCreateWithFlags(&h->ev_adv, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) h->ev_adv = nullptr;
    if (hipEventCreate(&h->ev_ply0) != hipSuccess || hipEventCreate(&h->ev_ply1) != hipSuccess ||
        hipHostMalloc((void**)&h->hcount, 4, 0) != hipSuccess) { h->fail("cannot create the ply-loop events / pinned counter"); return bail(AGZ_ERR_HIP); }
    if (!getenv("AGZ_NO_HOST_FLAG") && hipHostMalloc((void**)&h->hflag, 32, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        h->hflag[0] = 0; h->hflag[1] = 0; h->hflag[2] = 0; h->hflag[3] = 0;
        if (hipHostGetDevicePointer((void**)&h->hflag_dev, h->hflag, 0) != hipSuccess) { hipHostFree(h->hflag); h->hflag = nullptr; h->hflag_dev = nullptr; }
    }
    hipError_t fa = hipSuccess;                                     // first failure of the attribute / memset calls below
#define FA_(call) do { hipError_t r_ = (call); if (fa == hipSuccess) fa = r_; } while (0)
    FA_(hipFuncSetAttribute((const void*)k_mlp_wave<128, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_wave<128, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_wave<128, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_wave<128, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_wave<64, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_wave<64, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_big<512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_big<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_big<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_big<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_big<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_mlp_big<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_layer_exact<EX_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_layer_exact<EX_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_layer_exact<EX_POLICY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FA_(hipFuncSetAttribute((const void*)k_layer_exact<EX_VALUE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // node record: [prior f32 x A2][q f32 x A2][rank u8 x A2][cid u8 x A2][vis u8 x A2], A2 = 8 lanes x KPL actions
    // (agz_tree_eager.hpp)
    h->reg_lds = (size_t)eager_lds_layout(h->V, 8, 8 * h->reg_kpl).total;
#ifdef AGZ_STAMPS
    h->reg_lds += 256;
#endif
    const uint32_t A2 = (uint32_t)(8 * h->reg_kpl);
    const uint32_t rec_bytes = (uint32_t)eager_rec_bytes((int)A2, h->V);
    {
        FA_(hipFuncSetAttribute((const void*)h->k_eager, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->reg_lds));
        FA_(hipFuncSetAttribute((const void*)h->k_eager3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->reg_lds));
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) { h->reg3_max_waves = 12 * prop.multiProcessorCount; h->cus = prop.multiProcessorCount; }   // 3 waves x 4 SIMDs per CU
        const char* e3 = getenv("AGZ_REG3_MAX_WAVES");
        if (e3) h->reg3_max_waves = atoi(e3);
        e3 = getenv("AGZ_SMALL_MAXL");
        if (e3) h->small_maxl = atoi(e3);
        e3 = getenv("AGZ_SMALL_GPW");
        if (e3 && atoi(e3) >= 1 && atoi(e3) <= 8) h->small_gpw = atoi(e3);
        e3 = getenv("AGZ_SMALL4_MAXL");
        if (e3) h->small4_maxl = atoi(e3);
        e3 = getenv("AGZ_BIG_MAXL");
        if (e3) h->big_maxl = atoi(e3);
        e3 = getenv("AGZ_BIG_MT");
        if (e3 && (atoi(e3) == 2 || atoi(e3) == 4 || atoi(e3) == 8)) h->big_mt = atoi(e3);
        e3 = getenv("AGZ_WL_LDS_BYTES");
        if (e3 && atoi(e3) >= 0) h->wl_lds_max = atoi(e3) & ~15;
        for (int i = 0; i < 2; ++i) if (h->k_big[i]) FA_(hipFuncSetAttribute((const void*)h->k_big[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_big8) FA_(hipFuncSetAttribute((const void*)h->k_big8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_big8x) FA_(hipFuncSetAttribute((const void*)h->k_big8x, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e3 = getenv("AGZ_NARROW");
        if (e3) h->narrow_mode = atoi(e3);
        e3 = getenv("AGZ_NARROW_MINL");
        if (e3) h->narrow_minl = atoi(e3);
        e3 = getenv("AGZ_NARROW_OCC");
        if (e3 && (atoi(e3) == 0 || atoi(e3) == 1)) h->narrow_occ = atoi(e3);
        for (int i = 0; i < h->nnar; ++i) for (int j = 0; j < 2; ++j) FA_(hipFuncSetAttribute((const void*)h->nar[i].k[j], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < h->nnar; ++i) if (h->nar[i].ksp) FA_(hipFuncSetAttribute((const void*)h->nar[i].ksp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e3 = getenv("AGZ_PERSIST");
        if (e3) h->persist = atoi(e3) > 0 ? 1 : 0;
        e3 = getenv("AGZ_RESERVE_CUS");
        if (e3 && atoi(e3) > 0) h->reserve_slots = std::min(atoi(e3) * 128, std::max(0, (h->Lmax - 128) / 128 * 128));
        if (h->k_persist) FA_(hipFuncSetAttribute((const void*)h->k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_persist_nar) FA_(hipFuncSetAttribute((const void*)h->k_persist_nar, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_persist_nar_sp) FA_(hipFuncSetAttribute((const void*)h->k_persist_nar_sp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e3 = getenv("AGZ_NARROW_SPARSE");
        if (e3) h->nar_sparse = atoi(e3);
        if (h->k_persist_age) FA_(hipFuncSetAttribute((const void*)h->k_persist_age, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 2; ++i) if (h->k_persist_big[i]) FA_(hipFuncSetAttribute((const void*)h->k_persist_big[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_persist_tw4) FA_(hipFuncSetAttribute((const void*)h->k_persist_tw4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_persist_tw4_age) FA_(hipFuncSetAttribute((const void*)h->k_persist_tw4_age, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_persist_big4) FA_(hipFuncSetAttribute((const void*)h->k_persist_big4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 3; ++i) if (h->k_big4[i]) FA_(hipFuncSetAttribute((const void*)h->k_big4[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e3 = getenv("AGZ_BIG4");
        if (e3) h->big4 = atoi(e3) > 0 ? 1 : 0;
        e3 = getenv("AGZ_PERSIST_TW");
        h->persist_tw4 = !(e3 && atoi(e3) == 8) && h->k_persist_tw4;
        e3 = getenv("AGZ_AGE");
        if (e3) h->age_on = atoi(e3) > 0;
        e3 = getenv("AGZ_AGE_OLD16");
        if (e3 && atoi(e3) >= 0 && atoi(e3) <= 16) h->age_old16 = atoi(e3);
        e3 = getenv("AGZ_AGE_CLASS");
        if (e3) h->age_by_block = strcmp(e3, "block") == 0;
        e3 = getenv("AGZ_AGE_WAVE");
        if (e3) h->age_by_wave = atoi(e3) > 0;
        e3 = getenv("AGZ_AGE_BACKLOG");
        if (e3 && atoi(e3) > 0) h->age_backlog = atoi(e3);
        e3 = getenv("AGZ_SMALL4_OCC");
        if (e3 && atoi(e3) >= 0 && atoi(e3) <= 2) h->small4_occ = atoi(e3);
        if (h->k_small) FA_(hipFuncSetAttribute((const void*)h->k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (h->k_small8) FA_(hipFuncSetAttribute((const void*)h->k_small8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 3; ++i) if (h->k_small4[i]) FA_(hipFuncSetAttribute((const void*)h->k_small4[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int lv = 0; lv < h->ncmp; ++lv) {
            FA_(hipFuncSetAttribute((const void*)h->cmp[lv].s2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            FA_(hipFuncSetAttribute((const void*)h->cmp[lv].s8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (int i = 0; i < 3; ++i) FA_(hipFuncSetAttribute((const void*)h->cmp[lv].s4[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (int i = 0; i < 2; ++i) FA_(hipFuncSetAttribute((const void*)h->cmp[lv].b[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            FA_(hipFuncSetAttribute((const void*)h->cmp[lv].b8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            FA_(hipFuncSetAttribute((const void*)h->cmp[lv].b8x, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    }
    const size_t Lm = (size_t)h->Lmax, V = (size_t)h->V;
    h->INP = round_up(2 * P.VS, 32);
    h->LGS = round_up(P.A + 1, 32);
    hipError_t e = hipSuccess;
    auto A_ = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A_(dmalloc(&h->recs, Lm * V * rec_bytes));
    if (h->recs) FA_(hipMemsetAsync(h->recs, 0, Lm * V * rec_bytes, h->stream));   // a lane-group without a work item reads its game's root record: finite numbers from the start
    A_(dmalloc(&h->states, Lm * V));
    A_(dmalloc(&h->meta, Lm * V));
    A_(dmalloc(&h->ncount, Lm)); A_(dmalloc(&h->leaf, Lm)); A_(dmalloc(&h->game_id, Lm)); A_(dmalloc(&h->game_id2, Lm));
    A_(dmalloc(&h->slot_ply, Lm)); A_(dmalloc(&h->slot_ply2, Lm));
    if (h->slot_ply) FA_(hipMemsetAsync(h->slot_ply, 0, Lm * 4, h->stream));
    A_(dmalloc(&h->cnt_p, Lm)); A_(dmalloc(&h->cnt_new, Lm));
    size_t wl_blocks = 0;
    {
        h->wl_cap = (uint32_t)(8 * h->V);
        wl_blocks = (size_t)h->Lmax + 64;                  // one block per tree wave; a forced AGZ_SMALL_GPW = 1 makes every game a wave
        A_(dmalloc(&h->wl, wl_blocks * h->wl_cap)); A_(dmalloc(&h->wl_n, wl_blocks)); A_(dmalloc(&h->sp, Lm));
    }
    if (cfg->nn_mode == AGZ_NN_BF16) { uint16_t* p = nullptr; A_(dmalloc(&p, Lm * h->INP)); h->planes = p; }
    else { float* p = nullptr; A_(dmalloc(&p, Lm * h->INP)); h->planes = p; }
    A_(dmalloc(&h->logits, Lm * h->LGS));
    A_(dmalloc(&h->prior_eval, Lm * P.A + 64)); A_(dmalloc(&h->v_eval, Lm)); A_(dmalloc(&h->policy_final, Lm * P.A));
    A_(dmalloc(&h->newpos, Lm)); A_(dmalloc(&h->alive, Lm)); A_(dmalloc(&h->newslot, Lm)); A_(dmalloc(&h->d_count, 4));
    A_(dmalloc(&h->d_stats, 16)); A_(dmalloc(&h->d_acc, 2)); A_(dmalloc(&h->d_pacc, 16));
    if (h->k_persist_age) {                                                     // the migration queue: a ring of at least twice the slots
        h->mq_cap = 64; while (h->mq_cap < 2u * (uint32_t)h->Lmax) h->mq_cap <<= 1;
        A_(dmalloc(&h->mq_buf, h->mq_cap)); A_(dmalloc(&h->mq_ctr, 2));
        if (e == hipSuccess) { FA_(hipMemset(h->mq_buf, 0, (size_t)h->mq_cap * sizeof(MigEntry))); FA_(hipMemset(h->mq_ctr, 0, 16)); }
    }
    FA_(hipMemset(h->d_acc, 0, 16));
    A_(dmalloc(&h->scratch_f, Lm * (size_t)((P.A > 2 * P.VS) ? P.A : 2 * P.VS)));
    h->sample_games = cfg->sample_capacity_games > 0 ? cfg->sample_capacity_games : h->Lmax;
    const size_t SG = (size_t)h->sample_games, MP = (size_t)P.max_plies;
    A_(dmalloc(&h->s_boards, SG * MP * 6)); A_(dmalloc(&h->s_policy, SG * MP * P.A)); A_(dmalloc(&h->s_move, SG * MP)); A_(dmalloc(&h->s_net, SG * MP));
    A_(dmalloc(&h->g_nplies, SG)); A_(dmalloc(&h->g_result, SG)); A_(dmalloc(&h->g_final, SG));
    A_(dmalloc(&h->d_order, SG * MP));
    if (e != hipSuccess) { h->fail("device allocation failed: %s", hipGetErrorString(e)); return bail(AGZ_ERR_NOMEM); }
    FA_(hipMemsetAsync(h->meta, 0, Lm * V * 4, h->stream));
    FA_(hipMemsetAsync(h->policy_final, 0, Lm * P.A * 4, h->stream));
    FA_(hipMemsetAsync(h->planes, 0, Lm * h->INP * (cfg->nn_mode == AGZ_NN_BF16 ? 2 : 4), h->stream));
    FA_(hipMemsetAsync(h->logits, 0, Lm * h->LGS * 4, h->stream));
    FA_(hipMemsetAsync(h->prior_eval, 0, Lm * P.A * 4, h->stream));
    FA_(hipMemsetAsync(h->v_eval, 0, Lm * 4, h->stream));
    FA_(hipMemsetAsync(h->cnt_p, 0, Lm * 4, h->stream)); hipMemsetAsync(h->cnt_new, 0, Lm * 4, h->stream);
    FA_(hipMemsetAsync(h->g_nplies, 0, SG * 4, h->stream));

    TreePar& T = h->tp;
    memset(&T, 0, sizeof T);
    T.G = P; T.V = h->V; T.rec_bytes = rec_bytes; T.off_q = 16 + A2 * 6; T.off_vis = 16 + A2 * 6 + 8 * (uint32_t)eager_vl(h->V, (int)A2); T.A2 = A2;   // (off_q: the edge list {q, prior} by rank)
    T.recs = h->recs; T.states = h->states; T.meta = h->meta; T.ncount = h->ncount; T.leaf = h->leaf; T.game_id = h->game_id;
    T.cnt_p = h->cnt_p; T.cnt_new = h->cnt_new; T.planes = h->planes; T.INP = h->INP; T.planes_f32 = cfg->nn_mode == AGZ_NN_EXACT;
    T.logits = h->logits; T.LGS = h->LGS; T.prior_eval = h->prior_eval; T.v_eval = h->v_eval; T.policy_final = h->policy_final;
    T.seed = cfg->seed; T.exact = cfg->nn_mode == AGZ_NN_EXACT;
    T.wl = h->wl; T.wl_n = h->wl_n; T.sp = h->sp; T.wl_cap = h->wl_cap;
    T.rank_fault = h->d_stats + 5;
    T.slot_ply = h->slot_ply;
    h->rd_rec_bytes = rec_bytes; h->rd_off_rk = 16 + A2 * 4; h->rd_off_el = T.off_q; h->rd_off_vis = T.off_vis;
    FA_(hipMemsetAsync(h->wl_n, 0, wl_blocks * 4, h->stream)); hipMemsetAsync(h->sp, 0, Lm * 4, h->stream);
#if defined(AGZ_STAMPS) || defined(AGZ_BIGSTAMPS) || defined(AGZ_WGTIME)
    { unsigned long long* d = nullptr; hipMalloc((void**)&d, (size_t)65536 * 16 * 8); hipMemset(d, 0, (size_t)65536 * 16 * 8); T.dbg = d; }
#endif
#undef FA_
    if (fa != hipSuccess) { h->fail("device setup failed: %s", hipGetErrorString(fa)); return bail(AGZ_ERR_HIP); }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->fail("device init failed"); return bail(AGZ_ERR_HIP); }
    *out = h;
    return AGZ_OK;
}

int agz_get_search_form(agz_engine* h, char* tree_kernel, char* nn_kernel, int cap) {
    if (!h || cap < 1) return AGZ_ERR_ARG;
    if (tree_kernel) { strncpy(tree_kernel, h->form_tree.c_str(), (size_t)cap - 1); tree_kernel[cap - 1] = 0; }
    if (nn_kernel) { strncpy(nn_kernel, h->form_nn.c_str(), (size_t)cap - 1); nn_kernel[cap - 1] = 0; }
    return AGZ_OK;
}
int agz_set_seed(agz_engine* h, uint64_t seed) {
    if (!h) return AGZ_ERR_ARG;
    if (h->chain_live && h->chain_L > 0 && seed != h->cfg.seed) {   // (the games in flight draw from the key they started with)
        h->fail("agz_set_seed: %d games of a chain of self-play calls are in flight (end the chain with next_ngames = 0 first)", h->chain_L); return AGZ_ERR_STATE; }
    h->cfg.seed = seed; h->tp.seed = seed;
    return AGZ_OK;
}
int agz_set_network_tag(agz_engine* h, uint32_t tag) {
    if (!h) return AGZ_ERR_ARG;
    if (tag > 255u) { h->fail("agz_set_network_tag: tag=%u outside [0,255]", tag); return AGZ_ERR_ARG; }
    h->net_tag = tag;
    return AGZ_OK;
}
int agz_get_info(const agz_engine* h, agz_game_info* out) {
    if (!h || !out) return AGZ_ERR_ARG;
    *out = h->info; return AGZ_OK;
}
void* agz_stream(agz_engine* h) { return h ? (void*)h->stream : nullptr; }
#if defined(AGZ_WGTIME)
// start / end times (100 MHz) of the workgroups of the last 256 searches: out[(step & 255)][workgroup < 512][2]
extern "C" int agz_debug_wgtimes(agz_engine* h, unsigned long long* out) {
    hipStreamSynchronize(h->stream);
    return hipMemcpy(out, h->tp.dbg, (size_t)256 * 512 * 2 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
#if defined(AGZ_STAMPS) || defined(AGZ_BIGSTAMPS)
// out[0..15]: the tree step's phases (blocks below 32768), out[16..31]: the network body's (blocks from 32768 on)
extern "C" int agz_debug_stamps(agz_engine* h, unsigned long long* out, int reset) {
    hipStreamSynchronize(h->stream);
    std::vector<unsigned long long> all((size_t)65536 * 16);
    hipMemcpy(all.data(), h->tp.dbg, all.size() * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) out[i] = 0;
    for (size_t b = 0; b < 65536; ++b) for (int i = 0; i < 16; ++i) out[(b >= 32768 ? 16 : 0) + i] += all[b * 16 + i];
    if (reset) hipMemset(h->tp.dbg, 0, all.size() * 8);
    return 0;
}
#endif
// diagnostic: per-slot count of expanded nodes traversed since the search began (not part of include/agz.h)
extern "C" int agz_debug_slot_depths(agz_engine* h, uint32_t* out) {
    hipStreamSynchronize(h->stream);
    return hipMemcpy(out, h->cnt_p, (size_t)h->L * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
int agz_synchronize(agz_engine* h) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AGZ_OK;
}

int agz_init_weights(uint64_t seed, int in, int H, int T, int A, float* W0, float* Wres, float* Wp, float* bp, float* Wv, float* bv) {
    // Flux 0.12 Dense default: glorot_uniform weights U(+-sqrt(6/(fan_in+fan_out))), zero bias (DenseNet.jl:195-197)
    auto glorot = [&](uint32_t tensor, int out, int inn, float* W) {
        float limit = sqrtf(6.0f / (float)(inn + out));
        for (long e = 0; e < (long)out * inn; ++e) {
            uint32_t o[4];
            philox4x32_10((uint32_t)e, tensor, 0u, 0x57454947u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
            float u = (float)(o[0] >> 8) * 5.9604644775390625e-8f;
            W[e] = (2.0f * u - 1.0f) * limit;
        }
    };
    if (!W0 || !Wp || !bp || !Wv || !bv || (T > 0 && !Wres)) return AGZ_ERR_ARG;
    glorot(0, H, in, W0);
    for (int t = 0; t < T; ++t) glorot((uint32_t)(1 + t), H, H, Wres + (size_t)t * H * H);
    glorot((uint32_t)(T + 1), A, H, Wp);
    glorot((uint32_t)(T + 2), 1, H, Wv);
    for (int a = 0; a < A; ++a) bp[a] = 0.0f;
    bv[0] = 0.0f;
    return AGZ_OK;
}

int agz_set_network_slot(agz_engine* h, int which, int H, int T, const float* W0, const float* Wres, const float* Wp,
                         const float* bp, const float* Wv, const float* bv) {
    if (!h) return AGZ_ERR_ARG;
    if (which < 0 || which > 1 || H < 1 || H > 1024 || T < 0 || !W0 || !Wp || !bp || !Wv || !bv || (T > 0 && !Wres)) {
        h->fail("agz_set_network: bad arguments"); return AGZ_ERR_ARG;
    }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const GamePar& P = h->G;
    DevNet& n = h->net[which];
    free_net(n);
    n.H = H; n.T = T; n.in = 2 * P.VS; n.A = P.A; n.INP = h->INP; n.AOP = h->LGS;
    const size_t Lm = (size_t)h->Lmax;
    if (h->cfg.nn_mode == AGZ_NN_EXACT) {
        HIPCHK(h, dmalloc(&n.W0, (size_t)H * n.in)); HIPCHK(h, dmalloc(&n.Wres, (size_t)(T > 0 ? T : 1) * H * H));
        HIPCHK(h, dmalloc(&n.Wp, (size_t)P.A * H)); HIPCHK(h, dmalloc(&n.bp, (size_t)P.A)); HIPCHK(h, dmalloc(&n.Wv, (size_t)H)); HIPCHK(h, dmalloc(&n.bv, 1));
        HIPCHK(h, hipMemcpy(n.W0, W0, (size_t)H * n.in * 4, hipMemcpyHostToDevice));
        if (T > 0) HIPCHK(h, hipMemcpy(n.Wres, Wres, (size_t)T * H * H * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(n.Wp, Wp, (size_t)P.A * H * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(n.bp, bp, (size_t)P.A * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(n.Wv, Wv, (size_t)H * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(n.bv, bv, 4, hipMemcpyHostToDevice));
        if (H > h->Hcap || !h->actf0) {
            hipFree(h->actf0); hipFree(h->actf1); h->actf0 = h->actf1 = nullptr;
            HIPCHK(h, dmalloc(&h->actf0, Lm * H)); HIPCHK(h, dmalloc(&h->actf1, Lm * H));
            h->Hcap = H;
        }
    } else {
        if (H % 32 != 0) { h->fail("bf16 mode needs H to be a multiple of 32 (got %d)", H); return AGZ_ERR_ARG; }
        const int KT0 = n.INP / 16, NTh = H / 32, KTh = H / 16, NThead = n.AOP / 32;
        n.NT_h = NTh; n.NT_head = NThead;
        // all fragments live in ONE allocation [layer 0 | T residual layers | head] (t0 owns it)
        std::vector<uint16_t> buf;
        const size_t sz0 = (size_t)KT0 * NTh * 512, per = (size_t)KTh * NTh * 512, szh = (size_t)KTh * NThead * 512;
        HIPCHK(h, dmalloc(&n.t0, sz0 + per * T + szh));
        n.tres = n.t0 + sz0; n.thead = n.tres + per * T;
        buf.assign(sz0, 0);
        tile_weights(W0, H, n.in, NTh, KT0, buf);
        HIPCHK(h, hipMemcpy(n.t0, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
        for (int t = 0; t < T; ++t) {
            buf.assign(per, 0);
            tile_weights(Wres + (size_t)t * H * H, H, H, NTh, KTh, buf);
            HIPCHK(h, hipMemcpy(n.tres + per * t, buf.data(), per * 2, hipMemcpyHostToDevice));
        }
        buf.assign(szh, 0);
        tile_weights(Wp, P.A, H, NThead, KTh, buf);                 // rows 0..A-1: policy head
        tile_weights(Wv, 1, H, NThead, KTh, buf, P.A, 1);           // row A: value head
        HIPCHK(h, hipMemcpy(n.thead, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
        if (H % 16 == 0 && n.INP % 32 == 0 && H % 32 == 0) {       // 16x16x32 tiling of the same sections
            const int KT0b = n.INP / 32, KThb = H / 32, NTb = H / 16, NTheadb = n.AOP / 16;
            const size_t s0 = (size_t)KT0b * NTb * 512, sr = (size_t)KThb * NTb * 512, sh = (size_t)KThb * NTheadb * 512;
            std::vector<uint16_t> b16(s0 + sr * T + sh, 0);
            tile_weights16(W0, H, n.in, NTb, KT0b, b16.data());
            for (int t = 0; t < T; ++t) tile_weights16(Wres + (size_t)t * H * H, H, H, NTb, KThb, b16.data() + s0 + sr * t);
            tile_weights16(Wp, P.A, H, NTheadb, KThb, b16.data() + s0 + sr * T);
            tile_weights16(Wv, 1, H, NTheadb, KThb, b16.data() + s0 + sr * T, P.A, 1);
            HIPCHK(h, dmalloc(&n.w16, b16.size()));
            HIPCHK(h, hipMemcpy(n.w16, b16.data(), b16.size() * 2, hipMemcpyHostToDevice));
            if (H == 256 || H == 512) {   // agz_nn_big.hpp: layer 0 padded to an even row count, 2 rows of slack
                const int k0r = (KT0b + 3) / 4 * 4;                 // (zero rows pad layer 0: the network body walks four k-rows per turn)
                const size_t row = (size_t)NTb * 512;
                std::vector<uint16_t> bb(((size_t)k0r + (size_t)T * KThb + 4) * row, 0);   // (+ 4 rows of slack: the prefetch runs four rows ahead)
                std::copy(b16.begin(), b16.begin() + s0, bb.begin());
                std::copy(b16.begin() + s0, b16.begin() + (s0 + sr * T), bb.begin() + (size_t)k0r * row);
                HIPCHK(h, dmalloc(&n.wbig, bb.size()));
                HIPCHK(h, hipMemcpy(n.wbig, bb.data(), bb.size() * 2, hipMemcpyHostToDevice));
                n.k0r = k0r;
            }
            // agz_nn_wave.hpp: groups of KThb k-rows x NTb tiles: layer 0 padded with zero rows to whole groups, zero-weight
            // residual groups (identity) up to a multiple of NW_DEPTH, the head padded with zero tiles, NW_DEPTH groups of slack
            // (a head of up to 2 H outputs takes two groups: Gobang 13x13 / Hex 12x12 on a 128-wide trunk)
            if (NTheadb <= 2 * NTb && NTb % NW_WAVES == 0) {
                const int G0 = (KT0b + KThb - 1) / KThb, NGH = nw_hidden_groups(n.INP, H, T), HG = (NTheadb + NTb - 1) / NTb;
                const size_t s0p = (size_t)G0 * sr;
                std::vector<uint16_t> bw((size_t)(NGH + 2 + NW_DEPTH) * sr, 0);
                std::copy(b16.begin(), b16.begin() + s0, bw.begin());
                std::copy(b16.begin() + s0, b16.begin() + (s0 + sr * T), bw.begin() + s0p);
                for (int hg = 0; hg < HG; ++hg) {                  // head group hg: output rows 16 NTb hg ...
                    tile_weights16(Wp, P.A, H, NTb, KThb, bw.data() + (size_t)(NGH + hg) * sr, -16 * NTb * hg);
                    tile_weights16(Wv, 1, H, NTb, KThb, bw.data() + (size_t)(NGH + hg) * sr, P.A - 16 * NTb * hg, 1);
                }
                HIPCHK(h, dmalloc(&n.w16w, bw.size()));
                HIPCHK(h, hipMemcpy(n.w16w, bw.data(), bw.size() * 2, hipMemcpyHostToDevice));
            }
        }
        std::vector<float> bh((size_t)n.AOP, 0.0f);
        for (int a = 0; a < P.A; ++a) bh[a] = bp[a];
        bh[P.A] = bv[0];
        HIPCHK(h, dmalloc(&n.bias_head, bh.size()));
        HIPCHK(h, hipMemcpy(n.bias_head, bh.data(), bh.size() * 4, hipMemcpyHostToDevice));
        if (H > h->Hcap || !h->act0) {
            hipFree(h->act0); hipFree(h->act1); h->act0 = h->act1 = nullptr;
            HIPCHK(h, dmalloc(&h->act0, Lm * H)); HIPCHK(h, dmalloc(&h->act1, Lm * H));
            h->Hcap = H;
        }
    }
    n.loaded = true;
    return AGZ_OK;
}
int agz_set_network(agz_engine* h, int H, int T, const float* W0, const float* Wres, const float* Wp, const float* bp,
                    const float* Wv, const float* bv) {
    return agz_set_network_slot(h, 0, H, T, W0, Wres, Wp, bp, Wv, bv);
}

// ---- roots ------------------------------------------------------------------------------------------
static void image_to_pos(const GamePar& P, const uint8_t* s, Pos& p) {     // SURVEY Appendix B memory image
    p = Pos();
    memcpy(p.p, s, 24); memcpy(p.o, s + 48, 24);
    if (P.fam == F_REV) { memcpy(p.lg, s + 96, 24); p.player = (int8_t)s[144]; }
    else { p.player = (int8_t)s[96]; p.aux = (int8_t)s[97]; }
}

int agz_set_roots(agz_engine* h, const void* positions, int format, const uint32_t* game_ids, int L) {
    if (!h) return AGZ_ERR_ARG;
    if (L < 0 || L > h->Lmax) { h->fail("agz_set_roots: L=%d outside [0,%d]", L, h->Lmax); return AGZ_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const GamePar& P = h->G;
    std::vector<Pos> roots((size_t)L);
    std::vector<uint32_t> ids((size_t)L);
    const int ib = h->info.pos_image_bytes;
    for (int i = 0; i < L; ++i) {
        if (!positions) roots[i] = start_pos(P);
        else if (format == AGZ_POS_JULIA) image_to_pos(P, (const uint8_t*)positions + (size_t)i * ib, roots[i]);
        else if (format == AGZ_POS_COMPACT) memcpy(&roots[i], (const uint8_t*)positions + (size_t)i * sizeof(Pos), sizeof(Pos));
        else { h->fail("agz_set_roots: unknown position format %d", format); return AGZ_ERR_ARG; }
        ids[i] = game_ids ? game_ids[i] : h->cfg.game_id_base + (uint32_t)i;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (L > 0) {
        HIPCHK(h, hipMemcpy2D(h->states, (size_t)h->V * sizeof(Pos), roots.data(), sizeof(Pos), sizeof(Pos), (size_t)L, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->game_id, ids.data(), (size_t)L * 4, hipMemcpyHostToDevice));
    }
    h->L = L; h->need_reset = true; h->chain_live = false;     // (new roots end a chain of self-play calls)
    return AGZ_OK;
}

// ---- launches ---------------------------------------------------------------------------------------
static std::pair<hipEvent_t, hipEvent_t>* next_events(agz_engine* h, std::vector<std::pair<hipEvent_t, hipEvent_t>>& pool, size_t& used) {
    if (!h->ev_ref_live) {                    // time origin of this batch of events (recorded before any of them)
        if (!h->ev_ref) hipEventCreate(&h->ev_ref);
        hipEventRecord(h->ev_ref, h->stream);
        h->ev_ref_live = true;
    }
    if (used == pool.size()) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        pool.push_back({a, b});
    }
    return &pool[used++];
}
static void drain_events(agz_engine* h) {     // every recorded event has completed
    // tree kernels: sum of the launch durations, and the BUSY time = length of the union of the launch intervals (with
    // sub-batch chains several launches run side by side; the union is the time during which the kernel type was active)
    std::vector<std::pair<float, float>> iv;
    iv.reserve(h->ev_tree_used);
    for (size_t i = 0; i < h->ev_tree_used; ++i) {
        float ms = 0, t0 = 0;
        if (hipEventElapsedTime(&ms, h->ev_tree[i].first, h->ev_tree[i].second) == hipSuccess) h->tree_ms += ms;
        if (h->ev_ref_live && hipEventElapsedTime(&t0, h->ev_ref, h->ev_tree[i].first) == hipSuccess) iv.push_back({t0, t0 + ms});
    }
    std::sort(iv.begin(), iv.end());
    float cur_s = 0, cur_e = -1;
    for (auto& x : iv) {
        if (cur_e < cur_s || x.first > cur_e) { if (cur_e > cur_s) h->tree_busy_ms += cur_e - cur_s; cur_s = x.first; cur_e = x.second; }
        else if (x.second > cur_e) cur_e = x.second;
    }
    if (cur_e > cur_s) h->tree_busy_ms += cur_e - cur_s;
    h->tree_launches += (int64_t)h->ev_tree_used;
    for (size_t i = 0; i < h->ev_nn_used; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, h->ev_nn[i].first, h->ev_nn[i].second) == hipSuccess) h->nn_ms += ms; }
    h->ev_tree_used = h->ev_nn_used = 0;
    h->ev_ref_live = false;
}

// agz_fastdiv.hpp's operand range: lambda p >= 2^-100 needs lambda = cpuct sqrt(n) / (A + n) >= 2^-13 (p >= 2^-87), i.e. cpuct >= 2^-4
// for every A + n <= 512
static inline bool fastdiv_range(const agz_engine* h) { return !h->no_fastdiv && h->cpuct >= 0.0625f && h->cpuct <= 1024.0f; }

static int launch_rollout(agz_engine* h, uint32_t rollout, int do_reset, int do_expand, int do_select, int last, int inject, int capture,
                          int s0 = 0, int s1 = -1, hipStream_t stream = nullptr, int final_ = 0) {
    if (h->L == 0) return AGZ_OK;
    if (s1 < 0) s1 = h->L;
    if (!stream) stream = h->stream;
    const int n = s1 - s0;
    TreePar T = h->tp;
    T.L = s1; T.slot0 = s0; T.gpw = 8; T.step = h->step; T.rollout = rollout; T.cpuct = h->cpuct; T.training = h->training;
    T.do_reset = do_reset; T.do_expand = do_expand; T.do_select = do_select; T.last = last; T.inject = inject; T.capture = capture;
    T.final_ = final_;
    T.fastdiv = h->cfg.nn_mode == AGZ_NN_BF16 && fastdiv_range(h);
    dim3 grid((unsigned)((n + 7) / 8)), block(64);                // 8 games per wavefront
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if ((h->profiling & 1) && h->prof_this) { ev = next_events(h, h->ev_tree, h->ev_tree_used); hipEventRecord(ev->first, stream); }
    hipLaunchKernelGGL((int)grid.x <= h->reg3_max_waves ? h->k_eager3 : h->k_eager, grid, block, h->reg_lds, stream, T);
    if (ev) hipEventRecord(ev->second, stream);
    h->cnt_live = true;
    HIPCHK(h, hipGetLastError());
    return AGZ_OK;
}

static int launch_network(agz_engine* h, int which, int s0 = 0, int s1 = -1, hipStream_t stream = nullptr) {
    if (h->L == 0) return AGZ_OK;
    if (s1 < 0) s1 = h->L;
    if (!stream) stream = h->stream;
    DevNet& n = h->net[which];
    if (!n.loaded) { h->fail("no network loaded in slot %d (call agz_set_network)", which); return AGZ_ERR_STATE; }
    const int L = s1 - s0;                                  // rows [s0, s1) of every per-slot buffer
    const size_t pe = h->cfg.nn_mode == AGZ_NN_EXACT ? 4 : 2;
    const uint8_t* const planes = (const uint8_t*)h->planes + (size_t)s0 * h->INP * pe;
    float* const logits = h->logits + (size_t)s0 * h->LGS; float* const v_eval = h->v_eval + s0;
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if ((h->profiling & 2) && h->prof_this) { ev = next_events(h, h->ev_nn, h->ev_nn_used); hipEventRecord(ev->first, stream); h->nn_leaves += (uint64_t)L; }
    const int big_rowb = (2 * std::max(n.H, 32 * n.k0r) + 255) & ~255;   // (a multiple of the LDS bank row: agz_nn_big.hpp's XOR swizzle)
    if (h->cfg.nn_mode == AGZ_NN_BF16 && n.wbig && (size_t)NB_M * big_rowb <= 160 * 1024 && !h->no_fused_nn) {   // wide trunk: activations resident in LDS, weights streamed from L2
        BigPar B;
        B.planes = (const uint16_t*)planes; B.INP = n.INP; B.wh = n.wbig;
        B.whead = n.w16 + (size_t)(n.INP / 32) * (n.H / 16) * 512 + (size_t)n.T * (n.H / 32) * (n.H / 16) * 512;
        B.bias_head = n.bias_head; B.logits = logits; B.LGS = h->LGS; B.vout = v_eval;
        B.L = L; B.T = n.T; B.A = h->G.A; B.AOP = n.AOP; B.K0R = n.k0r; B.ROWB = big_rowb;
        // 32, 64 or 128 leaves per workgroup.  Measured per ply (512x8, V = 64): sub-batch chains of 10923 leaves (32768 games) 14.4 ms
        // with 64, 17.1 with 128, 17.4 with 32; one chain of 16384 leaves 10.0 with 64, 11.0 with 32; of 8192 leaves 8.0 with 32, 8.8 with 64
        // (sub-batch chains: what counts is the whole batch in flight — chains of 8192 leaves, 24576 games: 12.2 ms with 64, 13.5 with 32)
        int mt = h->L <= 8192 ? 2 : (L <= 20000 ? 4 : 8);
        if (h->big_mt) mt = h->big_mt;
        { char b[96]; snprintf(b, sizeof b, "k_mlp_big<H=%d,MT=%d> (%d leaves per workgroup, one launch)", n.H, mt, 16 * mt); h->form_nn = b; }
        const size_t lds = (size_t)16 * mt * big_rowb;
        dim3 grid((unsigned)((L + 16 * mt - 1) / (16 * mt))), block(NB_THREADS);
        if (n.H == 512) { if (mt == 8) hipLaunchKernelGGL((k_mlp_big<512, 8>), grid, block, lds, stream, B); else if (mt == 4) hipLaunchKernelGGL((k_mlp_big<512, 4>), grid, block, lds, stream, B); else hipLaunchKernelGGL((k_mlp_big<512, 2>), grid, block, lds, stream, B); }
        else { if (mt == 8) hipLaunchKernelGGL((k_mlp_big<256, 8>), grid, block, lds, stream, B); else if (mt == 4) hipLaunchKernelGGL((k_mlp_big<256, 4>), grid, block, lds, stream, B); else hipLaunchKernelGGL((k_mlp_big<256, 2>), grid, block, lds, stream, B); }
    } else if (h->cfg.nn_mode == AGZ_NN_BF16 && (n.H == 64 || n.H == 128) && n.w16w && !h->no_fused_nn) {   // one wave per 16 leaves, weights streamed from L2: lowest latency
        Fused3Par F;
        F.planes = (const uint16_t*)planes; F.INP = n.INP; F.w16 = n.w16w; F.bias_head = n.bias_head;
        F.logits = logits; F.LGS = h->LGS; F.vout = v_eval; F.L = L; F.T = n.T; F.A = h->G.A; F.AOP = n.AOP; F.gpw = 0; F.tw = 0; F.rb = 8;
        const int kth = n.H / 32, g0 = (n.INP / 32 + kth - 1) / kth;
        int lt = h->nn_wave_lt > 0 ? h->nn_wave_lt : (L <= 16384 ? 1 : 2);   // measured (128x6): 11 / 14 / 20 us at 2048 / 8192 / 16384 leaves with 1 tile, 27 us at 32768 with 2
        const size_t lds = (size_t)16 * lt * (2 * (n.H * 2 + 16) + (g0 * kth * 64 + 16));
        dim3 grid((unsigned)((L + 16 * lt - 1) / (16 * lt))), block(64 * NW_WAVES);
        const int depth = h->nn_wave_depth > 0 ? h->nn_wave_depth : 2;   // two layers ahead already cover the L2 latency; 4 only costs occupancy
        { char b[96]; snprintf(b, sizeof b, "k_mlp_wave<H=%d,LT=%d,DEPTH=%d> (one launch)", n.H, lt, depth); h->form_nn = b; }
#define NW_LAUNCH(HH, LL) do { if (depth == 4) hipLaunchKernelGGL((k_mlp_wave<HH, LL, 4>), grid, block, lds, stream, F); \
                               else hipLaunchKernelGGL((k_mlp_wave<HH, LL, 2>), grid, block, lds, stream, F); } while (0)
        if (n.H == 128) { if (lt == 1) NW_LAUNCH(128, 1); else if (lt == 2) NW_LAUNCH(128, 2); else if (lt == 4) NW_LAUNCH(128, 4); else NW_LAUNCH(128, 8); }
        else { if (lt == 1) NW_LAUNCH(64, 1); else if (lt == 2) NW_LAUNCH(64, 2); else if (lt == 4) NW_LAUNCH(64, 4); else NW_LAUNCH(64, 8); }
#undef NW_LAUNCH
    } else if (h->cfg.nn_mode == AGZ_NN_BF16) {
        h->form_nn = "k_layer_bf16 (one launch per layer)";   // any trunk width (H = 32, 96, 1024 ...): the generic path
        dim3 block(256);
        dim3 gh((unsigned)((L + GB_M - 1) / GB_M), (unsigned)((n.H + GB_N - 1) / GB_N));
        const uint16_t* x = (const uint16_t*)planes;
        hipLaunchKernelGGL(k_layer_bf16<EPI_RELU>, gh, block, 0, stream, x, n.INP, n.INP, n.t0, n.NT_h, h->act0 + (size_t)s0 * n.H, n.H, n.H, L,
                           (const float*)nullptr, (float*)nullptr, 0, (float*)nullptr, 0);
        uint16_t *a = h->act0 + (size_t)s0 * n.H, *b = h->act1 + (size_t)s0 * n.H;
        const size_t per = (size_t)(n.H / 16) * n.NT_h * 512;
        for (int t = 0; t < n.T; ++t) {
            hipLaunchKernelGGL(k_layer_bf16<EPI_RES>, gh, block, 0, stream, (const uint16_t*)a, n.H, n.H, n.tres + per * t, n.NT_h, b, n.H,
                               n.H, L, (const float*)nullptr, (float*)nullptr, 0, (float*)nullptr, 0);
            uint16_t* s = a; a = b; b = s;
        }
        dim3 gp((unsigned)((L + GB_M - 1) / GB_M), (unsigned)((n.AOP + GB_N - 1) / GB_N));
        hipLaunchKernelGGL(k_layer_bf16<EPI_HEAD>, gp, block, 0, stream, (const uint16_t*)a, n.H, n.H, n.thead, n.NT_head, (uint16_t*)nullptr, 0,
                           n.AOP, L, (const float*)n.bias_head, logits, h->LGS, v_eval, h->G.A);
    } else {
        h->form_nn = "k_layer_exact (fp32, one launch per layer)";
        dim3 block(64, 4);
        auto grid = [&](int O) { return dim3((unsigned)((L + EX_TL - 1) / EX_TL), (unsigned)((O + 63) / 64)); };
        auto shm = [&](int K) { return (size_t)EX_TL * (size_t)((K + 3) & ~3) * 4; };
        const float* x = (const float*)planes;
        hipLaunchKernelGGL(k_layer_exact<EX_RELU>, grid(n.H), block, shm(n.in), stream, x, h->INP, n.in, (const float*)n.W0, n.H, h->actf0 + (size_t)s0 * n.H, n.H, L, (const float*)nullptr);
        float *a = h->actf0 + (size_t)s0 * n.H, *b = h->actf1 + (size_t)s0 * n.H;
        for (int t = 0; t < n.T; ++t) {
            hipLaunchKernelGGL(k_layer_exact<EX_RES>, grid(n.H), block, shm(n.H), stream, (const float*)a, n.H, n.H,
                               (const float*)(n.Wres + (size_t)t * n.H * n.H), n.H, b, n.H, L, (const float*)nullptr);
            float* s = a; a = b; b = s;
        }
        hipLaunchKernelGGL(k_layer_exact<EX_POLICY>, grid(n.A), block, shm(n.H), stream, (const float*)a, n.H, n.H, (const float*)n.Wp, n.A,
                           logits, h->LGS, L, (const float*)n.bp);
        hipLaunchKernelGGL(k_layer_exact<EX_VALUE>, grid(1), block, shm(n.H), stream, (const float*)a, n.H, n.H, (const float*)n.Wv, 1,
                           v_eval, 1, L, (const float*)n.bv);
    }
    if (ev) hipEventRecord(ev->second, stream);
    HIPCHK(h, hipGetLastError());
    return AGZ_OK;
}

static int check_search_args(agz_engine* h, int V) {
    if (V < 1 || V > h->V) { h->fail("V=%d outside [1,%d]", V, h->V); return AGZ_ERR_ARG; }
    return AGZ_OK;
}

// the level of rows by legal rank for the coming search: the fewest entries per lane R with 8 R >= the bound on a root's legal actions
// (set by the ply loop) whose compaction buffer fits the lane-group's edge table (2 V floats >= 8 R); -1: rows by action
static int cmp_level(const agz_engine* h) {
    int best = -1;
    if (h->no_compact) return best;
    for (int i = 0; i < h->ncmp; ++i) {
        const int r = h->cmp[i].kpr;
        if (h->legal_bound <= 8 * r && 2 * h->V >= 8 * r && (best < 0 || r < h->cmp[best].kpr)) best = i;
    }
    return best;
}

int agz_search_actor(agz_engine* h, int which, int V, float cpuct, int training, uint32_t step) {
    if (h) h->tree_kpr = 0;
    if (!h) return AGZ_ERR_ARG;
    int rc = check_search_args(h, V); if (rc) return rc;
    if (which < 0 || which > 1 || !h->net[which].loaded) { h->fail("no network loaded in slot %d", which); return AGZ_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->chain_live && h->chain_L > 0 && !h->in_ply_loop) {   // (a search on those slots would overwrite the plies of the games in flight)
        h->fail("agz_search: %d games of a chain of self-play calls are in flight (end the chain with next_ngames = 0, or set new roots)", h->chain_L); return AGZ_ERR_STATE; }
    h->cpuct = cpuct; h->training = training; h->step = step;
    if (!h->in_ply_loop && h->L > 0) HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->slot_ply, (int)step, (size_t)h->L, h->stream));   // every slot's game is at ply `step`
    h->rd_rec_bytes = h->tp.rec_bytes; h->rd_off_rk = 16 + h->tp.A2 * 4; h->rd_off_el = h->tp.off_q; h->rd_off_vis = h->tp.off_vis;
    // profiling bit 2: instrument (events, per-slot counters) only every 4th search: HIP events around ~10^4 launches per
    // generation cost ~10 % of the time they are meant to measure
    h->prof_this = !(h->profiling & 4) || (h->search_seq++ & 3u) == 0;
    // The slots are independent, so the batch is cut into up to KCH sub-batches (multiples of 128 slots = one network
    // tile) and each runs its own select -> network -> expand/backup chain on its own stream: while one chain's tree
    // kernel waits on memory latency the other chains' network and tree kernels fill the machine.  Results do not depend
    // on the cut (every per-game quantity is keyed by game id).
    {   // narrow lane-groups (4 or 2 lanes per tree): big batches of the 128-wide trunk, every game resident at once
        DevNet& n = h->net[which];
        const agz_engine::Narrow* nk = nullptr;
        if (h->narrow_mode >= 0 && h->cfg.nn_mode == AGZ_NN_BF16 && n.H == 128 && n.w16w && h->L >= (h->narrow_minl >= 0 ? h->narrow_minl : 112 * h->cus) && h->V <= 128 && (h->V & 3) == 0 && !h->no_fused_nn)
            for (int i = 0; i < h->nnar; ++i) {
                const agz_engine::Narrow& c = h->nar[i];
                if (h->narrow_mode > 0 && c.g != h->narrow_mode) continue;
                // by default only where it was measured faster: games with few actions (Connect4: 4 lanes x 4 actions per tree, 2.54 vs 2.84 ms
                // per search at 32768 games; 2.41 vs 2.24 at 24576) — on a 9x9 board 4 lanes x 24 actions halve the vector instructions and
                // the waves, and the launch takes as long as before (4.35 vs 4.21 ms): it is bound by the waves' dependent chains, not by issue
                if (h->narrow_mode == 0 && (c.kpl > 4 || c.g != 4)) continue;
                if (c.g * c.kpl > h->LGS) continue;                                   // (the lean tree step reads whole blocks of logits)
                if (c.kpr && (h->no_compact || h->legal_bound > c.g * c.kpr || 2 * h->V < c.g * c.kpr)) continue;
                const int wgs = (h->L + 4 * (64 / c.g) - 1) / (4 * (64 / c.g));
                if (wgs > 2 * h->cus) continue;                                       // every workgroup resident (two per CU at most)
                // the narrowest group first (h->nar lists wider groups first only by accident: pick by g, then the narrowest rows)
                if (!nk || c.g < nk->g || (c.g == nk->g && (c.kpr ? c.kpr : c.kpl) < (nk->kpr ? nk->kpr : nk->kpl))) nk = &c;
            }
        if (nk) {
            // sparse waves (round 6): above 64 games per CU — where the dense form has its two waves per SIMD — eight games per wave of sixteen groups,
            // 32-game workgroups, four per CU, four waves per SIMD; the groups without a game take work items
            const bool sp = nk->ksp && h->nar_sparse > 0 && h->narrow_occ < 0 && (h->L > 64 * h->cus || h->nar_sparse > 1) && (h->L + 4 * (32 / nk->g) - 1) / (4 * (32 / nk->g)) <= 4 * h->cus;
            const int G = nk->g, NG = 64 / G, GPW = sp ? NG / 2 : NG, tw = 4, gpwg = tw * GPW;
            const int wgs = (h->L + gpwg - 1) / gpwg;
            int one = sp ? 0 : (h->narrow_occ >= 0 && wgs <= h->cus ? h->narrow_occ : (wgs <= h->cus ? 1 : 0));   // one workgroup per CU: the 512-register build
            const uint32_t A2 = (uint32_t)(G * (nk->kpr ? nk->kpr : nk->kpl));
            h->tree_kpr = nk->kpr;
            SmallPar S; S.nxw_off = 0;
            S.T = h->tp;
            S.T.L = h->L; S.T.slot0 = 0; S.T.step = h->step; S.T.cpuct = h->cpuct; S.T.training = h->training;
            S.T.fastdiv = fastdiv_range(h);
            S.T.inject = 0; S.T.capture = 0; S.T.rollout = 0; S.T.do_reset = 1; S.T.do_expand = 0; S.T.do_select = 1; S.T.last = 0;
            S.T.gpw = GPW;
            S.F.planes = (const uint16_t*)h->planes; S.F.INP = n.INP; S.F.w16 = n.w16w; S.F.bias_head = n.bias_head;
            S.F.logits = h->logits; S.F.LGS = h->LGS; S.F.vout = h->v_eval; S.F.L = h->L; S.F.T = n.T; S.F.A = h->G.A; S.F.AOP = n.AOP;
            S.F.gpw = 0; S.F.tw = tw; S.F.rb = GPW;
            S.V = V; S.tree_lds = eager_lds_layout(h->V, NG, G * nk->kpl).total;
            const int rs = small_io_row_bytes(n);
            S.io_prowb = rs; S.io_lgs = rs / 4; S.io_bw = GPW * rs;
            S.io_off = (int)((std::max((size_t)tw * (size_t)S.tree_lds, (size_t)gpwg * 2 * (n.H * 2 + 16)) + 15) & ~(size_t)15);
            S.xch_off = S.io_off + tw * S.io_bw;
            size_t shared = (size_t)S.xch_off + (size_t)tw * (16 * NG + 16);
            if (!one && wgs <= h->cus && shared > (size_t)(80 * 1024)) one = 1;     // (32 trees per wave: the tables of a workgroup take more than half a CU's LDS)
            const size_t cu_lds = (size_t)(160 * 1024) / (size_t)(sp ? 4 : (one ? 1 : 2));
            S.nxw_off = place_nxw(shared, cu_lds, tw, NG, h->V);
            const size_t room = cu_lds > shared ? cu_lds - shared : 0;
            S.wl_off = (int)shared; S.wl_bytes = (int)std::min({(size_t)(NG * h->V * 4), (room / (size_t)tw) & ~(size_t)15, (size_t)h->wl_lds_max});
            const size_t lds = shared + (size_t)tw * S.wl_bytes;
            if (lds <= cu_lds) {
                std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
                if ((h->profiling & 1) && h->prof_this) { ev = next_events(h, h->ev_tree, h->ev_tree_used); hipEventRecord(ev->first, h->stream); }
                hipLaunchKernelGGL(sp ? nk->ksp : nk->k[one], dim3((unsigned)wgs), dim3(64 * NW_WAVES), lds, h->stream, S);
                if (nk->kpr) {   // policy_final back to action order (one wave per game, in place)
                    PlyPar Q; memset(&Q, 0, sizeof Q);
                    Q.G = h->G; Q.L = h->L; Q.V = h->V; Q.states = h->states; Q.policy_final = h->policy_final;
                    hipLaunchKernelGGL(h->k_spread, dim3((unsigned)((h->L + 3) / 4)), dim3(256), 0, h->stream, Q);
                }
                if (ev) hipEventRecord(ev->second, h->stream);
                HIPCHK(h, hipGetLastError());
                h->rd_rec_bytes = (uint32_t)eager_rec_bytes((int)A2, h->V); h->rd_off_rk = 16 + A2 * 4; h->rd_off_el = 16 + A2 * 6;
                h->rd_off_vis = 16 + A2 * 6 + 8 * (uint32_t)eager_vl(h->V, (int)A2);
                { char kb[48] = ""; if (nk->kpr) snprintf(kb, sizeof kb, ",rows by legal rank KPR=%d", nk->kpr);
                  char b[220]; snprintf(b, sizeof b, "k_search_small<KPL=%d,H=128,TW=4,WV=%d%s,G=%d> (whole mcts_single per launch, %d lanes per tree, %d games per tree wave, %d per workgroup)",
                                        nk->kpl, sp ? 4 : (one ? 1 : 2), kb, G, G, GPW, gpwg); h->form_tree = b; h->form_nn = "inside k_search_small (mlp_wave_body<128>)"; }
                h->cnt_live = true;
                h->need_reset = true; h->injected = false;
                if (h->prof_this) h->total_rollouts += (uint64_t)h->L * (uint64_t)V;
                return AGZ_OK;
            }
        }
    }
    {   // every game resident at once (<= 128 per CU): the whole search in one launch (agz_search_small.hpp); profiling bit 0
        // then times that launch (it counts as one "tree launch" of agz_get_kernel_times)
        DevNet& n = h->net[which];
        if (h->k_small && h->cfg.nn_mode == AGZ_NN_BF16 && n.H == 128 && n.w16w && h->L > 0 &&
            h->V <= 128 && (h->V & 3) == 0 && 8 * h->reg_kpl <= h->LGS &&   // the lean build of the tree step (rollout_eager_body<..., LEAN>)
            // (rows of 24 actions per lane — 13x13 boards — still spill ~60 vector registers in the 3- and 4-workgroups-per-CU builds, and
            //  the one-launch form wins all the same since the scalar spills went: 11.4 vs 24.0 ms per ply at 32768 games of Gobang
            //  13x13, 9.2 vs 20.5 at 24576; 16 actions per lane (11x11 boards) do not spill: 5.6 vs 16.5 ms at 32768 games of Gobang 11x11)
            h->L <= std::min(std::max(h->small_maxl, h->small4_maxl), 128 * h->cus) && !h->no_fused_nn) {
            // 16 games per workgroup up to small_maxl; beyond, 32 games per workgroup with the loosest register budget that still
            // keeps every workgroup resident (2 / 3 / 4 workgroups per CU = 64 / 96 / 128 games per CU)
            int tw = h->L <= h->small_maxl ? 2 : 4;
            int occ = h->L <= 64 * h->cus ? 0 : (h->L <= 96 * h->cus ? 1 : 2);
            if (h->small4_occ >= 0) occ = h->small4_occ;      // AGZ_SMALL4_OCC (tests: every register budget at small sizes)
            // beyond 96 games per CU: 64-game workgroups of eight waves (measured per ply at 32768 / 28672 games of Gobang 9x9: 4.10 / 3.91
            // vs 4.18 / 4.00 ms with four 32-game workgroups per CU; at 24576 games 3.80 vs 3.39 ms: not below)
            const bool t8 = h->k_small8 && tw == 4 && occ == 2 && (h->tw8 > 0 || (h->tw8 == 0 && h->reg_kpl == 12 && h->G.fam != F_REV));   // (Connect4 +1 %, Reversi 8x8 0 .. +2 %, 11x11 / 13x13 -0.5 .. -1.6 %: not by default)
            if (t8) tw = 8;
            // rows by the root's legal rank once no root can have more legal actions than they hold (the ply loop knows: A - ply); the
            // expansion compacts through the group's edge table, 2 V >= 8 KPR floats
            const int lv = cmp_level(h);                          // the narrowest rows that hold every root's legal actions, or -1
            const bool cmp = lv >= 0;
            const small_fn kfn = t8 ? (cmp ? h->cmp[lv].s8 : h->k_small8)
                                    : (cmp ? (tw == 2 ? h->cmp[lv].s2 : h->cmp[lv].s4[occ]) : (tw == 2 ? h->k_small : h->k_small4[occ]));
            h->tree_kpr = cmp ? h->cmp[lv].kpr : 0;
            SmallPar S; S.nxw_off = 0;
            S.T = h->tp;
            S.T.L = h->L; S.T.slot0 = 0; S.T.step = h->step; S.T.cpuct = h->cpuct; S.T.training = h->training;
            S.T.fastdiv = fastdiv_range(h);
            S.T.inject = 0; S.T.capture = 0; S.T.rollout = 0; S.T.do_reset = 1; S.T.do_expand = 0; S.T.do_select = 1; S.T.last = 0;
            S.F.planes = (const uint16_t*)h->planes; S.F.INP = n.INP; S.F.w16 = n.w16w; S.F.bias_head = n.bias_head;
            S.F.logits = h->logits; S.F.LGS = h->LGS; S.F.vout = h->v_eval; S.F.L = h->L; S.F.T = n.T; S.F.A = h->G.A; S.F.AOP = n.AOP;
            // few games: sparse waves — a rollout lasts as long as the deepest descent among the games of a workgroup, and with
            // <= 2 workgroups per CU idle lanes cost nothing: 1 / 2 games per tree wave up to 4 / 8 games per CU (measured per ply:
            // 2.9 vs 4.0 ms at 256 games, 3.1 vs 3.9 at 1024, 3.5 vs 3.9 at 2048; no gain from 4 games per wave at 4096)
            // the fewest games per tree wave that keep every workgroup resident (4 tree waves per CU): 1 .. 8
            // (16-game workgroups: 2 tree waves x 2 workgroups per CU; 32-game workgroups at 2 per CU: 8 tree waves per CU)
            const int res_waves = (tw >= 4 ? 4 * (2 + occ) : 4) * h->cus;      // tree waves resident at once
            // sparse waves up to 64 games per CU (measured per ply: 2.40 vs 2.82 ms at 5120 games, 2.90 vs 3.43 at 10240); with 3 and 4
            // workgroups per CU (more than 16384 games) 7 games per wave are not reliably faster than 8 (3.8 vs 4.2 ms at 17408, 6.8 vs 6.2 at 26624)
            const int gpw = (tw >= 4 && occ > 0) ? 8 : std::min(8, std::max(1, (h->L + res_waves - 1) / res_waves));
            S.T.gpw = h->small_gpw > 0 && (tw == 2 || occ == 0) ? h->small_gpw : gpw;
            S.F.gpw = S.T.gpw < 8 ? S.T.gpw : 0; S.F.tw = tw;
            S.V = V; S.tree_lds = (int)h->reg_lds;
            // the tree waves' tables and the network's two activation strips share one window (the phases never overlap); the hand-over
            // window behind it carries planes (tree -> network) and logits (network -> tree), one block of 8 rows per tree wave
            const int rs = small_io_row_bytes(n);   // one row per game: the leaf's planes on the way to the network, its logits on the way back
            S.io_prowb = rs; S.io_lgs = rs / 4; S.io_bw = 8 * rs; S.F.rb = 8;
            h->rd_rec_bytes = h->tp.rec_bytes; h->rd_off_rk = 16 + h->tp.A2 * 4; h->rd_off_el = h->tp.off_q; h->rd_off_vis = h->tp.off_vis;
            S.io_off = (int)((std::max((size_t)(tw == 8 ? 8 : 4) * h->reg_lds, (size_t)8 * tw * 2 * (n.H * 2 + 16)) + 15) & ~(size_t)15);   // (16-game workgroups: the two helper waves have tables of their own)
            size_t shared = (size_t)S.io_off + (size_t)tw * S.io_bw + (size_t)tw * 144;   // ... + the carry a tree wave publishes for its helper
            S.xch_off = S.io_off + tw * S.io_bw;
            // + the tree waves' next-word tables and work lists (kept across the network phase): what the CU's LDS leaves when every workgroup of
            // the launch must be resident (4 per CU at 32768 games); list entries beyond the region, rare, go to the global list
            const int wgs_per_cu = tw == 2 || tw == 8 ? 2 : 2 + occ;
            S.nxw_off = place_nxw(shared, (size_t)(160 * 1024) / (size_t)wgs_per_cu, tw, 8, h->V);
            const size_t room = (size_t)(160 * 1024) / (size_t)wgs_per_cu > shared ? (size_t)(160 * 1024) / (size_t)wgs_per_cu - shared : 0;
            S.wl_off = (int)shared; S.wl_bytes = (int)std::min({(size_t)(8 * h->V * 4), (room / (size_t)tw) & ~(size_t)15, (size_t)h->wl_lds_max});
            const size_t lds = shared + (size_t)tw * S.wl_bytes;
            std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
            if ((h->profiling & 1) && h->prof_this) { ev = next_events(h, h->ev_tree, h->ev_tree_used); hipEventRecord(ev->first, h->stream); }
            hipLaunchKernelGGL(kfn, dim3((unsigned)((h->L + S.T.gpw * tw - 1) / (S.T.gpw * tw))), dim3(64 * (tw == 8 ? 8 : NW_WAVES)), lds, h->stream, S);
            if (cmp) {   // policy_final back to action order (one wave per game, in place)
                PlyPar Q; memset(&Q, 0, sizeof Q);
                Q.G = h->G; Q.L = h->L; Q.V = h->V; Q.states = h->states; Q.policy_final = h->policy_final;
                hipLaunchKernelGGL(h->k_spread, dim3((unsigned)((h->L + 3) / 4)), dim3(256), 0, h->stream, Q);
            }
            { char kb[48] = ""; if (cmp) snprintf(kb, sizeof kb, ",rows by legal rank KPR=%d", h->tree_kpr);
              char b[200]; snprintf(b, sizeof b, "k_search_small<KPL=%d,H=128,TW=%d,WV=%d%s> (whole mcts_single per launch, %d games per workgroup, %d per tree wave)",
                                    h->reg_kpl, tw, tw == 2 ? 2 : 2 + occ, kb, S.T.gpw * tw, S.T.gpw); h->form_tree = b; h->form_nn = "inside k_search_small (mlp_wave_body<128>)"; }
            if (ev) hipEventRecord(ev->second, h->stream);
            HIPCHK(h, hipGetLastError());
            h->cnt_live = true;
            h->need_reset = true; h->injected = false;
            if (h->prof_this) h->total_rollouts += (uint64_t)h->L * (uint64_t)V;
            return AGZ_OK;
        }
    }
    {   // wide trunk, batch small enough to be latency-bound in the two-kernel form: the whole search in one launch (agz_search_big.hpp)
        DevNet& n = h->net[which];
        const int big_rowb = (2 * std::max(n.H, 32 * n.k0r) + 255) & ~255;   // (a multiple of the LDS bank row: agz_nn_big.hpp's XOR swizzle)
        if (h->k_big[0] && h->cfg.nn_mode == AGZ_NN_BF16 && n.H == 512 && n.wbig && h->L > 0 && 8 * h->reg_kpl <= h->LGS &&
            h->V <= 128 && (h->V & 3) == 0 &&                    // (the shapes the lean build of the tree step is tested at, as for k_search_small)
            h->L <= (h->big8 >= 0 && h->k_big8x && h->big_maxl >= 64 * h->cus ? 128 * h->cus : std::min(h->big_maxl, 64 * h->cus)) && !h->no_fused_nn) {
            BigSearchPar S;
            S.T = h->tp;
            S.T.L = h->L; S.T.slot0 = 0; S.T.step = h->step; S.T.cpuct = h->cpuct; S.T.training = h->training;
            S.T.fastdiv = fastdiv_range(h);
            S.T.inject = 0; S.T.capture = 0; S.T.rollout = 0; S.T.do_reset = 1; S.T.do_expand = 0; S.T.do_select = 1; S.T.last = 0;
            BigPar& B = S.B;
            B.planes = (const uint16_t*)h->planes; B.INP = n.INP; B.wh = n.wbig;
            B.whead = n.w16 + (size_t)(n.INP / 32) * (n.H / 16) * 512 + (size_t)n.T * (n.H / 32) * (n.H / 16) * 512;
            B.bias_head = n.bias_head; B.logits = h->logits; B.LGS = h->LGS; B.vout = h->v_eval;
            B.L = h->L; B.T = n.T; B.A = h->G.A; B.AOP = n.AOP; B.K0R = n.k0r; B.ROWB = big_rowb;
            const int gpw = std::min(8, std::max(1, (h->L + 4 * h->cus - 1) / (4 * h->cus)));   // sparse waves (see k_search_small)
            S.T.gpw = h->small_gpw > 0 ? h->small_gpw : gpw;
            int wgs = (h->L + 4 * S.T.gpw - 1) / (4 * S.T.gpw);
            const int occ = wgs <= h->cus ? 0 : 1;                 // 1 or 2 workgroups per CU
            // above 32 games per CU: one 64-game workgroup per CU (eight tree waves, four leaf tiles per network pass: a layer's weights
            // stream once for 64 games, and 185 instead of 128 registers) rather than two 32-game workgroups
            const bool b8 = occ == 1 && h->k_big8 && h->big8 >= 0 && S.T.gpw == 8;
            // ... and two of them above 64 games per CU: the 128-register build of the 64-leaf network pass spills ~40 registers and still beats
            // the two-kernel form (first ply at 32768 games of Gobang 9x9 512x8: 12.3 vs 14.1 ms, 24576: 11.2 vs 12.8)
            const bool x8 = b8 && h->L > 64 * h->cus;
            const int lv = cmp_level(h);                          // rows by the root's legal rank (see k_search_small above)
            // ... or, there, ONE 128-game workgroup per CU with 4 lanes per tree (k_search_big4): a layer's weights stream once per 128 leaves
            big_fn k4 = nullptr;
            if (x8 && h->big4 != 0 && (h->L + 127) / 128 <= h->cus && h->small_gpw <= 0) {
                if (lv < 0) k4 = h->k_big4[0];
                else for (int i = 1; i < 3; ++i) if (h->k_big4[i] && h->big4_kpr[i] == 2 * h->cmp[lv].kpr) k4 = h->k_big4[i];
                const size_t win = (std::max((size_t)8 * (size_t)eager_lds_layout(h->V, 16, 4 * h->big4_kpl).total, (size_t)128 * big_rowb) + 15) & ~(size_t)15;
                if (win + 4 * 144 + 8 * 256 > (size_t)(160 * 1024)) k4 = nullptr;
            }
            const int twb = b8 ? 8 : 4;
            if (b8) wgs = (h->L + 63) / 64;
            if (k4) { wgs = (h->L + 127) / 128; S.T.gpw = 16; }
            S.V = V; S.tree_lds = k4 ? eager_lds_layout(h->V, 16, 4 * h->big4_kpl).total : (int)h->reg_lds;
            S.xch_off = (int)((std::max((size_t)8 * (size_t)S.tree_lds, (size_t)(k4 ? 128 : 8 * twb) * big_rowb) + 15) & ~(size_t)15);   // (tree waves and helper waves have tables of their own)
            size_t shared = (size_t)S.xch_off + 4 * 144;
            const int wgcu = k4 ? 1 : (x8 ? 2 : (b8 ? 1 : occ + 1));
            S.nxw_off = place_nxw(shared, (size_t)(160 * 1024) / (size_t)wgcu, twb, k4 ? 16 : 8, h->V);
            const size_t room = (size_t)(160 * 1024) / (size_t)wgcu > shared ? (size_t)(160 * 1024) / (size_t)wgcu - shared : 0;
            S.wl_off = (int)shared; S.wl_bytes = (int)std::min({(size_t)((k4 ? 16 : 8) * h->V * 4), (room / (size_t)twb) & ~(size_t)15, (size_t)h->wl_lds_max});
            const size_t lds = shared + (size_t)twb * S.wl_bytes;
            std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
            if ((h->profiling & 1) && h->prof_this) { ev = next_events(h, h->ev_tree, h->ev_tree_used); hipEventRecord(ev->first, h->stream); }
            h->tree_kpr = lv < 0 ? 0 : h->cmp[lv].kpr;
            hipLaunchKernelGGL(k4 ? k4 : x8 ? (lv < 0 ? h->k_big8x : h->cmp[lv].b8x) : b8 ? (lv < 0 ? h->k_big8 : h->cmp[lv].b8) : (lv < 0 ? h->k_big[occ] : h->cmp[lv].b[occ]), dim3((unsigned)wgs), dim3(NB_THREADS), lds, h->stream, S);
            if (lv >= 0) {
                PlyPar Q; memset(&Q, 0, sizeof Q);
                Q.G = h->G; Q.L = h->L; Q.V = h->V; Q.states = h->states; Q.policy_final = h->policy_final;
                hipLaunchKernelGGL(h->k_spread, dim3((unsigned)((h->L + 3) / 4)), dim3(256), 0, h->stream, Q);
            }
            { char kb[48] = ""; if (lv >= 0) snprintf(kb, sizeof kb, ",rows by legal rank KPR=%d", h->tree_kpr);
              char b[200];
              if (k4) snprintf(b, sizeof b, "k_search_big4<KPL=%d,H=512,G=4%s> (whole mcts_single per launch, one 128-game workgroup per CU, 16 games per tree wave)", h->big4_kpl, kb);
              else snprintf(b, sizeof b, "k_search_big<KPL=%d,H=512,WG=%d%s%s> (whole mcts_single per launch, %d games per workgroup, %d per tree wave)",
                            h->reg_kpl, x8 ? 2 : (b8 ? 1 : occ + 1), b8 ? ",TW=8" : "", kb, twb * S.T.gpw, S.T.gpw);
              h->form_tree = b; h->form_nn = k4 ? "inside k_search_big4 (mlp_big_body<512,8>)" : b8 ? "inside k_search_big (mlp_big_body<512,4>)" : "inside k_search_big (mlp_big_body<512,2>)"; }
            if (ev) hipEventRecord(ev->second, h->stream);
            HIPCHK(h, hipGetLastError());
            h->cnt_live = true;
            h->need_reset = true; h->injected = false;
            if (h->prof_this) h->total_rollouts += (uint64_t)h->L * (uint64_t)V;
            return AGZ_OK;
        }
    }
    int K = 1;
    if (h->aux[0]) {
        // measured (128x6, V = 64): 3 chains -9 % at 32768 and 24576 games, 2 chains -11 % at 16384, nothing below ~12000
        K = h->chains > 0 ? h->chains : (h->L >= 20000 ? 3 : (h->L >= 12000 ? 2 : 1));
        const int kmax = (h->L + 255) / 256;               // at least 256 games per chain
        if (K > kmax) K = kmax;
        if (K > agz_engine::KCH) K = agz_engine::KCH;
        if (K < 1) K = 1;
    }
    const int chunk = ((h->L + K - 1) / K + 127) / 128 * 128;
    { char b[160]; snprintf(b, sizeof b, "%s (one launch per rollout, %d sub-batch chain%s)", "k_rollout_eager", K, K > 1 ? "s" : "");
      h->form_tree = b; }
    if (K > 1) HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    for (int c = 1; c < K; ++c) if (c * chunk < h->L) HIPCHK(h, hipStreamWaitEvent(h->aux[c - 1], h->ev_fork, 0));
    for (int k = 0; k <= V; ++k) {
        for (int c = 0; c < K; ++c) {
            const int s0 = c * chunk, s1 = (c + 1) * chunk < h->L ? (c + 1) * chunk : h->L;
            if (s0 >= s1) break;
            hipStream_t st = c == 0 ? h->stream : h->aux[c - 1];
            if (k < V) {
                rc = launch_rollout(h, (uint32_t)k, k == 0, k > 0, 1, k == V - 1, 0, 0, s0, s1, st); if (rc) return rc;
                rc = launch_network(h, which, s0, s1, st); if (rc) return rc;
            } else {
                rc = launch_rollout(h, (uint32_t)V, 0, 1, 0, 0, 0, 0, s0, s1, st, 1); if (rc) return rc;
            }
        }
    }
    for (int c = 1; c < K; ++c) if (c * chunk < h->L) {
        HIPCHK(h, hipEventRecord(h->ev_join[c - 1], h->aux[c - 1]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[c - 1], 0));
    }
    h->need_reset = true; h->injected = false;
    if (h->prof_this) h->total_rollouts += (uint64_t)h->L * (uint64_t)V;
    return AGZ_OK;
}
int agz_search(agz_engine* h, int V, float cpuct, int training, uint32_t step) { return agz_search_actor(h, 0, V, cpuct, training, step); }

// ---- stepwise (teacher-forced parity) ---------------------------------------------------------------
int agz_search_begin(agz_engine* h, float cpuct, int training, uint32_t step) {
    if (!h) return AGZ_ERR_ARG;
    if (h->chain_live && h->chain_L > 0) { h->fail("agz_search_begin: %d games of a chain of self-play calls are in flight (end the chain with next_ngames = 0, or set new roots)", h->chain_L); return AGZ_ERR_STATE; }
    h->cpuct = cpuct; h->training = training; h->step = step; h->need_reset = true; h->injected = false; h->step_last = false; h->tree_kpr = 0;
    if (h->L > 0) { HIPCHK(h, hipSetDevice(h->cfg.device)); HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->slot_ply, (int)step, (size_t)h->L, h->stream)); }
    h->rd_rec_bytes = h->tp.rec_bytes; h->rd_off_rk = 16 + h->tp.A2 * 4; h->rd_off_el = h->tp.off_q; h->rd_off_vis = h->tp.off_vis;
    return AGZ_OK;
}
int agz_rollout_select(agz_engine* h, uint32_t rollout, int last) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = launch_rollout(h, rollout, h->need_reset ? 1 : 0, 0, 1, last, 0, 0);
    h->need_reset = false; h->step_last = last != 0; h->step_rollout = rollout;
    return rc;
}
int agz_rollout_eval(agz_engine* h) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = launch_network(h, 0); if (rc) return rc;
    if (h->L > 0)
        hipLaunchKernelGGL(h->k_soft, dim3((unsigned)((h->L + 3) / 4)), dim3(256), 0, h->stream, (const float*)h->logits, h->LGS,
                           h->prior_eval, h->G.A, h->L, (int)(h->cfg.nn_mode == AGZ_NN_EXACT));
    HIPCHK(h, hipGetLastError());
    h->injected = true;
    return AGZ_OK;
}
int agz_get_eval(agz_engine* h, float* prior, float* v) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (prior) HIPCHK(h, hipMemcpy(prior, h->prior_eval, (size_t)h->L * h->G.A * 4, hipMemcpyDeviceToHost));
    if (v) HIPCHK(h, hipMemcpy(v, h->v_eval, (size_t)h->L * 4, hipMemcpyDeviceToHost));
    return AGZ_OK;
}
int agz_get_logits(agz_engine* h, float* logits, float* v) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (logits && h->L > 0)
        HIPCHK(h, hipMemcpy2D(logits, (size_t)h->G.A * 4, h->logits, (size_t)h->LGS * 4, (size_t)h->G.A * 4, (size_t)h->L, hipMemcpyDeviceToHost));
    if (v) HIPCHK(h, hipMemcpy(v, h->v_eval, (size_t)h->L * 4, hipMemcpyDeviceToHost));
    return AGZ_OK;
}
int agz_inject_eval(agz_engine* h, const float* prior, const float* v) {
    if (!h || !prior || !v) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->prior_eval, prior, (size_t)h->L * h->G.A * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->v_eval, v, (size_t)h->L * 4, hipMemcpyHostToDevice));
    h->injected = true;
    return AGZ_OK;
}
int agz_rollout_expand_backup(agz_engine* h) {
    if (!h) return AGZ_ERR_ARG;
    if (!h->injected) { h->fail("agz_rollout_expand_backup: call agz_rollout_eval or agz_inject_eval first"); return AGZ_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->injected = false;
    h->total_rollouts += (uint64_t)h->L;
    // eager kernel: the backup recomputes the rows the next descent samples from and keeps policy_final = the root's row as of
    // now (copy_pol :330-339 reads it at the start of the last rollout); after the last select nothing is recomputed any more
    return launch_rollout(h, h->step_rollout + 1u, 0, 1, 0, 1, 1, 0, 0, -1, nullptr, h->step_last ? 1 : 0);
}
int agz_search_end(agz_engine* h) {
    if (!h) return AGZ_ERR_ARG;
    h->need_reset = true;
    return AGZ_OK;
}

// ---- getters ----------------------------------------------------------------------------------------
static int fetch(agz_engine* h, void* dst, const void* src, size_t bytes) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (bytes) HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return AGZ_OK;
}
int agz_get_policy(agz_engine* h, float* out) {
    if (!h || !out) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return fetch(h, out, h->policy_final, (size_t)h->L * h->G.A * 4);
}
static int planes_getter(agz_engine* h, float* out, int use_leaf) {
    if (!h || !out) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->L == 0) return AGZ_OK;
    hipLaunchKernelGGL(k_planes, dim3((unsigned)h->L), dim3(128), 0, h->stream, (const Pos*)h->states, (const uint32_t*)h->leaf, use_leaf, h->V,
                       h->G.VS, h->L, h->scratch_f);
    HIPCHK(h, hipGetLastError());
    return fetch(h, out, h->scratch_f, (size_t)h->L * 2 * h->G.VS * 4);
}
int agz_get_batch(agz_engine* h, float* out) { return planes_getter(h, out, 0); }
int agz_get_leaf_batch(agz_engine* h, float* out) { return planes_getter(h, out, 1); }
static int stats_getter(agz_engine* h, float* out, int want_q) {
    if (!h || !out) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->L == 0) return AGZ_OK;
    if (h->tree_kpr) { h->fail("root statistics are not available after a ply-loop search with rows by legal rank (AGZ_NO_COMPACT=1 keeps rows by action)"); return AGZ_ERR_STATE; }
    hipLaunchKernelGGL(k_root_stats, dim3((unsigned)h->L), dim3(128), 0, h->stream, (const uint8_t*)h->recs, (const uint32_t*)h->meta, h->V,
                       h->rd_rec_bytes, h->rd_off_rk, h->rd_off_el, h->rd_off_vis, h->G.A, h->L, want_q ? (float*)nullptr : h->scratch_f,
                       want_q ? h->scratch_f : (float*)nullptr);
    HIPCHK(h, hipGetLastError());
    return fetch(h, out, h->scratch_f, (size_t)h->L * h->G.A * 4);
}
int agz_get_root_visits(agz_engine* h, float* out) { return stats_getter(h, out, 0); }
int agz_get_root_q(agz_engine* h, float* out) { return stats_getter(h, out, 1); }
int agz_get_leaf(agz_engine* h, int32_t* out) {
    if (!h || !out) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return fetch(h, out, h->leaf, (size_t)h->L * 4);
}
int agz_get_node_count(agz_engine* h, int32_t* out) {
    if (!h || !out) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return fetch(h, out, h->ncount, (size_t)h->L * 4);
}
static int fold_counters(agz_engine* h) {      // adds the per-slot counters of the LAST search to the running totals
    if (h->L == 0) return AGZ_OK;
    std::vector<uint32_t> a((size_t)h->L), b((size_t)h->L);
    int rc = fetch(h, a.data(), h->cnt_p, (size_t)h->L * 4); if (rc) return rc;
    rc = fetch(h, b.data(), h->cnt_new, (size_t)h->L * 4); if (rc) return rc;
    for (int i = 0; i < h->L; ++i) { h->acc_p += a[i]; h->acc_new += b[i]; }
    h->cnt_live = false;
    return AGZ_OK;
}
int agz_get_counters(agz_engine* h, uint64_t* sum_p, uint64_t* sum_new, uint64_t* rollouts) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // counters of the last search are folded lazily: acc_* holds completed folds (selfplay folds every ply)
    uint64_t p = h->acc_p, n = h->acc_new;
    { unsigned long long d[2] = {0, 0}; int rc0 = fetch(h, d, h->d_acc, 16); if (rc0) return rc0; p += d[0]; n += d[1]; }
    if (h->L > 0 && h->cnt_live) {
        std::vector<uint32_t> a((size_t)h->L), b((size_t)h->L);
        int rc = fetch(h, a.data(), h->cnt_p, (size_t)h->L * 4); if (rc) return rc;
        rc = fetch(h, b.data(), h->cnt_new, (size_t)h->L * 4); if (rc) return rc;
        for (int i = 0; i < h->L; ++i) { p += a[i]; n += b[i]; }
    }
    if (sum_p) *sum_p = p;
    if (sum_new) *sum_new = n;
    if (rollouts) *rollouts = h->total_rollouts;
    return AGZ_OK;
}

int agz_set_profiling(agz_engine* h, int enable) {
    if (!h) return AGZ_ERR_ARG;
    h->profiling = enable;
    return AGZ_OK;
}
int agz_get_kernel_times(agz_engine* h, double* tree_ms, double* nn_ms, int64_t* tree_launches, int reset) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    if (tree_ms) *tree_ms = h->tree_ms;
    if (nn_ms) *nn_ms = h->nn_ms;
    if (tree_launches) *tree_launches = h->tree_launches;
    if (reset) { h->nn_leaves = 0; h->tree_ms = h->nn_ms = h->tree_busy_ms = 0; h->tree_launches = 0; h->acc_p = h->acc_new = 0; h->total_rollouts = 0; hipMemsetAsync(h->d_acc, 0, 16, h->stream);
                 h->age_ranked_searches = h->age_searches = h->age_pushed = 0; }
    return AGZ_OK;
}

int agz_get_age_stats(agz_engine* h, uint64_t out[3]) {
    if (!h || !out) return AGZ_ERR_ARG;
    out[0] = h->age_searches; out[1] = h->age_ranked_searches; out[2] = h->age_pushed;
    return AGZ_OK;
}
int agz_get_nn_leaves(agz_engine* h, uint64_t* leaves) {
    if (!h || !leaves) return AGZ_ERR_ARG;
    *leaves = h->nn_leaves;
    return AGZ_OK;
}
int agz_get_tree_busy_ms(agz_engine* h, double* busy_ms) {
    if (!h || !busy_ms) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    *busy_ms = h->tree_busy_ms;
    return AGZ_OK;
}

// ---- self-play --------------------------------------------------------------------------------------
static void fill_plypar(agz_engine* h, PlyPar& T, int ply, int tau_plies, bool all_actions) {
    memset(&T, 0, sizeof T);
    T.all_actions = all_actions ? 1 : 0;
    T.G = h->G; T.L = h->L; T.V = h->V; T.ply = ply; T.tau_plies = tau_plies; T.seed = h->cfg.seed; T.game_id_base = h->cfg.game_id_base;
    T.states = h->states; T.game_id = h->game_id; T.slot_ply = h->slot_ply; T.policy_final = h->policy_final; T.newpos = h->newpos; T.alive = h->alive;
    T.next_game = h->d_stats + 6; T.identity = h->d_count;
    T.sample_games = h->sample_games; T.max_plies = h->G.max_plies;
    T.s_boards = h->s_boards; T.s_policy = h->s_policy; T.s_move = h->s_move; T.s_net = h->s_net; T.net_tag = h->net_tag; T.g_nplies = h->g_nplies; T.g_result = h->g_result;
    T.g_final = h->g_final; T.stats = h->d_stats;
}

// next_games < 0: a call of its own (agz_selfplay, agz_duel).  next_games >= 0: a call of a CHAIN (agz_selfplay_chain): game ids run on from
// the previous call of the chain, the call returns when ITS ngames games have finished, and while they run out it starts up to next_games
// games of the next call in the slots that come free — they stay in flight when the call returns and the next call goes on with them, so
// only the last call of a chain (next_games = 0) ends on a batch that runs out.
static int finish_call(agz_engine* h, int ngames, bool chain, unsigned long long k0, unsigned long long started, int in_flight, bool persist,
                       int64_t rollouts, int ply, double search_ms, std::chrono::steady_clock::time_point t0, agz_selfplay_stats* st);
static bool persist_shape(const agz_engine* h);
static int run_games_persist(agz_engine* h, int ngames, int V, float cpuct, int tau_plies, agz_selfplay_stats* st, int next_games);

static int run_games(agz_engine* h, int ngames, int V, float cpuct, int tau_plies, int training, int duel_first, bool duel,
                     agz_selfplay_stats* st, int next_games = -1) {
    int rc = check_search_args(h, V); if (rc) return rc;
    const bool chain = next_games >= 0;
    const bool cont = chain && h->chain_live;                                   // the chain goes on (possibly with games in flight)
    if (!duel && training) {
        // one launch per call (agz_selfplay_small.hpp) where the kernel exists: by default for calls that refill their slots on an engine whose
        // every workgroup is resident at more than 96 slots per CU (the batch the eight-wave workgroups are the fastest form of); a chain
        // stays in the form it began in (the persistent form does not compact its slots)
        const long long want_ = chain ? (long long)ngames + next_games : (long long)ngames;
        const bool refill_ = chain || want_ > h->Lmax;
        const bool use = cont ? h->chain_persist : (persist_shape(h) && (h->persist > 0 || (h->persist < 0 && refill_ && h->Lmax > (h->net[0].H == 512 ? 64 : 96) * h->cus)));
        if (use) {
            if (!persist_shape(h)) { h->fail("the chain's games are in flight in uncompacted slots (persistent form) and the engine's network / mode no longer allows that form"); return AGZ_ERR_STATE; }
            return run_games_persist(h, ngames, V, cpuct, tau_plies, st, next_games);
        }
    }
    const unsigned long long k0 = cont ? h->chain_k0 : 0ull;                    // this call's games: k0 .. k0 + ngames - 1 of the chain
    const unsigned long long pool_end = k0 + (unsigned long long)ngames + (unsigned long long)(chain ? next_games : 0);   // games that may be started
    // more games than slots (self-play only): the first Lmax games start together, and a slot whose game has ended takes the next game
    // that has not started yet (k_advance) until all ngames have been started — the batch stays full, every search of the generation
    // runs at the size the chip is filled by, and each game's samples are the ones a lock-step run over ngames slots gives (results are
    // keyed by game id and the game's own ply)
    const long long want = chain ? (long long)ngames + next_games : (long long)ngames;
    // AGZ_RESERVE_CUS=n: the launch leaves the slots of n CUs (128 each) without a workgroup, so that the kernels of another stream — the RCCL
    // all-gather of the call before, agz_comm_* — find n CUs' worth of wave slots and LDS free for the whole launch instead of waiting for a
    // persistent workgroup to end
    const int LM = h->Lmax - h->reserve_slots;
    const int slots = want < LM ? (int)want : LM;
    if (ngames < 1 || (duel && ngames > h->Lmax)) { h->fail("ngames=%d outside [1,%d]", ngames, h->Lmax); return AGZ_ERR_ARG; }
    if (ngames > slots && ngames > h->sample_games) { h->fail("ngames=%d exceeds the sample capacity of the engine (%d games: agz_config.sample_capacity_games)", ngames, h->sample_games); return AGZ_ERR_ARG; }
    if (chain && want > h->sample_games) { h->fail("agz_selfplay_chain: ngames + next_ngames = %lld exceeds the sample capacity of the engine (%d games: agz_config.sample_capacity_games)", want, h->sample_games); return AGZ_ERR_ARG; }
    if (chain && pool_end + (unsigned long long)h->cfg.game_id_base > 0xffffffffull) { h->fail("agz_selfplay_chain: game ids exhausted"); return AGZ_ERR_ARG; }
    auto t0 = std::chrono::steady_clock::now();
    unsigned long long started = 0, finished = 0;
    if (!cont) {
        rc = agz_set_roots(h, nullptr, 0, nullptr, slots); if (rc) return rc;   // Position() for every game (:479)
        HIPCHK(h, hipMemsetAsync(h->d_stats, 0, 16 * sizeof(unsigned long long), h->stream));
        HIPCHK(h, hipMemsetAsync(h->g_nplies, 0, (size_t)h->sample_games * 4, h->stream));
        HIPCHK(h, hipMemsetAsync(h->slot_ply, 0, (size_t)slots * 4, h->stream));
        started = (unsigned long long)slots;
        HIPCHK(h, hipMemcpyAsync(h->d_stats + 6, &started, 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        // the games the last call left in flight go on in their slots; per-call counters start over; the games of this call that finished
        // early (during the last call) are already counted; the ring entries of the games that may start now are cleared
        HIPCHK(h, hipSetDevice(h->cfg.device));
        h->L = h->chain_L; h->need_reset = true;
        started = h->chain_started;
        const unsigned long long cap = (unsigned long long)h->sample_games;
        for (unsigned long long a = started; a < pool_end;) {                    // [started, pool_end) in ring order, piece by piece
            const unsigned long long r = a % cap, n = std::min(pool_end - a, cap - r);
            HIPCHK(h, hipMemsetAsync(h->g_nplies + r, 0, (size_t)n * 4, h->stream));
            a += n;
        }
        {   // the games of THIS call that are already over (they ran, early, while the call before it was busy): their entries say so
            std::vector<int32_t> np((size_t)ngames);
            for (size_t a = 0; a < np.size();) {
                const size_t r = (size_t)((k0 + a) % cap), n = std::min(np.size() - a, (size_t)cap - r);
                HIPCHK(h, hipMemcpyAsync(np.data() + a, h->g_nplies + r, n * 4, hipMemcpyDeviceToHost, h->stream));
                a += n;
            }
            HIPCHK(h, hipStreamSynchronize(h->stream));
            finished = 0; for (int32_t x : np) finished += x > 0 ? 1ull : 0ull;
        }
        const unsigned long long z[10] = {0, 0, 0, 0, 0, 0, started, 0, finished, 0};
        HIPCHK(h, hipMemcpyAsync(h->d_stats, z, sizeof z, hipMemcpyHostToDevice, h->stream));
        // slots without a game (the last call's pool ran dry before its games were over) take new games right away
        const int fill = (int)std::min<unsigned long long>((unsigned long long)(slots > h->L ? slots - h->L : 0), pool_end > started ? pool_end - started : 0ull);
        if (fill > 0) {
            std::vector<Pos> roots((size_t)fill, start_pos(h->G));
            std::vector<uint32_t> ids((size_t)fill), zero((size_t)fill, 0u);
            for (int i = 0; i < fill; ++i) ids[i] = h->cfg.game_id_base + (uint32_t)(started + (unsigned long long)i);
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipMemcpy2D(h->states + (size_t)h->L * h->V, (size_t)h->V * sizeof(Pos), roots.data(), sizeof(Pos), sizeof(Pos), (size_t)fill, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(h->game_id + h->L, ids.data(), (size_t)fill * 4, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(h->slot_ply + h->L, zero.data(), (size_t)fill * 4, hipMemcpyHostToDevice));
            h->L += fill; started += (unsigned long long)fill;
            HIPCHK(h, hipMemcpy(h->d_stats + 6, &started, 8, hipMemcpyHostToDevice));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->chain_live = false;                                                      // (set again when a chained call returns in good order)
    h->sp_games = ngames; h->sp_nsamples = 0; h->sp_maxplies = 0;
    h->sp_k0 = (uint32_t)k0; h->sp_ring0 = chain ? (uint32_t)(k0 % (unsigned long long)h->sample_games) : 0u;
    double search_ms = 0; int64_t rollouts = 0; int ply = 0;
    uint32_t* const hcount = h->hcount;
    const bool refill = chain ? true : ngames > slots;
    // the youngest game alive bounds every root's stone count from below: all games have started (and the youngest was at ply 0 in round
    // first_all) -> every root of round r holds at least r - first_all stones
    int first_all = (refill && started < pool_end) ? -1 : (cont ? -2 : 0);      // round in which the last started game is at ply 0 (-1: games still start; -2: unknown — a chain's games of unknown age, no bound)
    bool fold_pending = false;
    // Run-ahead: while games still start, a ply cannot change the number of games in flight as long as more games are waiting than
    // slots exist (every game that ends is replaced) — the host then queues the next ply WITHOUT waiting for this one's count (up to 8
    // plies ahead; the search timings of those plies are read from their own event pairs at the next wait).  `started_ub` bounds the
    // games started so far from above between two waits.
    unsigned long long started_ub = started;
    int ahead = 0; bool any_fold = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tpool;                       // search-timing event pairs of the plies not yet waited for
    size_t tused = 0;
    tpool.push_back({h->ev_ply0, h->ev_ply1});
    h->in_ply_loop = true;
    while (h->L > 0) {                                                          // :494
        const int which = duel ? ((ply & 1) == 0 ? duel_first : 1 - duel_first) : 0;   // :592-596
        if (fold_pending) { hipStreamWaitEvent(h->stream, h->ev_fold, 0); fold_pending = false; }   // (the counters of the last search are folded before this one overwrites them)
        if (tused == tpool.size()) { hipEvent_t a = nullptr, b = nullptr; if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { h->fail("hipEventCreate failed"); rc = AGZ_ERR_HIP; break; } tpool.push_back({a, b}); }
        const hipEvent_t e0 = tpool[tused].first, e1 = tpool[tused].second; ++tused;
        if (hipEventRecord(e0, h->stream) != hipSuccess) { h->fail("hipEventRecord failed"); rc = AGZ_ERR_HIP; break; }
        h->legal_bound = ((h->G.fam == F_LINE || h->G.fam == F_HEX) && first_all >= 0) ? h->G.A - (ply - first_all) : 1 << 30;   // every game started from Position(): a game at ply p has p stones on the board
        rc = agz_search_actor(h, which, V, cpuct, training, (uint32_t)ply);      // :503
        h->legal_bound = 1 << 30;
        if (rc) break;
        if (hipEventRecord(e1, h->stream) != hipSuccess) { h->fail("hipEventRecord failed"); rc = AGZ_ERR_HIP; break; }
        rollouts += (int64_t)h->L * V;
        const bool fold = h->profiling && h->prof_this;                         // descent counters of this search -> d_acc: beside the ply kernels, on a
        if (fold) {                                                             // stream of its own (the next search waits for it, long after it has run)
            hipStream_t fs = h->fold_stream ? h->fold_stream : h->stream;
            if (h->fold_stream) hipStreamWaitEvent(fs, e1, 0);
            hipLaunchKernelGGL(k_fold_counters, dim3((unsigned)std::min(64, (h->L + 255) / 256)), dim3(256), 0, fs, (const uint32_t*)h->cnt_p,
                               (const uint32_t*)h->cnt_new, h->L, h->d_acc);
            if (h->fold_stream) { hipEventRecord(h->ev_fold, fs); fold_pending = true; }
            any_fold = true;
        }
        PlyPar T; fill_plypar(h, T, ply, tau_plies, duel);
        T.refill_total = refill ? (uint32_t)pool_end : 0u;
        T.ring = chain ? 1 : 0; T.k_cur_end = (uint32_t)(k0 + (unsigned long long)ngames);
        hipLaunchKernelGGL(h->k_adv, dim3((unsigned)((h->L + 3) / 4)), dim3(256), 0, h->stream, T);          // :513-549
        const bool sleep = h->ply_sleep && h->ev_adv && hipEventRecord(h->ev_adv, h->stream) == hipSuccess;
        const uint32_t seq = ++h->ply_seq ? h->ply_seq : ++h->ply_seq;        // (never 0)
        hipLaunchKernelGGL(k_scan_alive, dim3(1), dim3(1024), 0, h->stream, (const uint32_t*)h->alive, h->newslot, h->L, h->d_count, h->hflag_dev, seq,
                           (const unsigned long long*)(h->d_stats + 6), h->d_stats + 7, (const unsigned long long*)(h->d_stats + 8));
        hipLaunchKernelGGL(k_compact, dim3((unsigned)((h->L + 255) / 256)), dim3(256), 0, h->stream, T, (const uint32_t*)h->newslot,
                           (const uint32_t*)h->game_id, h->game_id2, h->slot_ply2);           // :550-561
        if (refill && ahead < h->run_ahead && started_ub + 2ull * (unsigned long long)h->L <= pool_end) {
            // this ply and the next cannot exhaust the games that wait: the batch stays as it is — queue the next ply now
            started_ub += (unsigned long long)h->L; ++ahead;
            { uint32_t* s = h->game_id; h->game_id = h->game_id2; h->game_id2 = s; h->tp.game_id = h->game_id; }
            { uint32_t* s = h->slot_ply; h->slot_ply = h->slot_ply2; h->slot_ply2 = s; h->tp.slot_ply = h->slot_ply; }
            ++ply;
            continue;
        }
        bool have = false;
        if (h->hflag_dev) {
            // the number of games left, as soon as the scan kernel has it: polled from host-visible memory while the compaction still runs
            // (the next search is queued behind it on the stream); a stream that ends without the word falls back to the copy below
            volatile unsigned long long* const f = h->hflag;
            // the search takes milliseconds: the host thread sleeps through it (a blocking event after k_advance — no core is burnt
            // per engine) and polls only for the scan that follows (~10 us)
            if (sleep) (void)hipEventSynchronize(h->ev_adv);
            for (uint32_t spin = 0;; ++spin) {
                const unsigned long long w = __atomic_load_n(f, __ATOMIC_ACQUIRE);
                if ((uint32_t)(w >> 32) == seq) { *hcount = (uint32_t)w; started = __atomic_load_n(f + 1, __ATOMIC_RELAXED); finished = __atomic_load_n(f + 2, __ATOMIC_RELAXED); have = true; break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();                                            // (a polite spin: the core is shared with the host's other threads)
#else
                std::this_thread::yield();
#endif
                if ((spin & 1023u) == 1023u) {
                    const hipError_t q = hipStreamQuery(h->stream);
                    if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
                }
            }
            if (have && hipEventSynchronize(e1) != hipSuccess) have = false;   // (the search's end event is long past)
        }
        if (!have) {
            if (hipMemcpyAsync(hcount, h->d_count, 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipMemcpyAsync(&started, h->d_stats + 6, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipMemcpyAsync(&finished, h->d_stats + 8, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { h->fail("ply loop failed: %s", hipGetErrorString(hipGetLastError())); rc = AGZ_ERR_HIP; break; }
        }
        for (size_t i = 0; i < tused; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, tpool[i].first, tpool[i].second) == hipSuccess) search_ms += ms; }
        tused = 0; ahead = 0; started_ub = started;
        if (any_fold) { h->cnt_live = false; drain_events(h); any_fold = false; }
        { uint32_t* s = h->game_id; h->game_id = h->game_id2; h->game_id2 = s; h->tp.game_id = h->game_id; }
        { uint32_t* s = h->slot_ply; h->slot_ply = h->slot_ply2; h->slot_ply2 = s; h->tp.slot_ply = h->slot_ply; }
        h->L = (int)*hcount;
        ++ply;
        if (first_all == -1 && started >= pool_end) first_all = ply;          // the last game was started in this round's k_advance: ply 0 in the next search
        if (chain && finished >= (unsigned long long)ngames) break;            // this call's games are all over; what is in flight belongs to the next call
        if (ply > 255 && !refill) { h->fail("game exceeded 255 plies"); rc = AGZ_ERR_STATE; break; }
        if (ply > 255 * ((ngames + slots - 1) / slots + 1)) { h->fail("ply loop does not end"); rc = AGZ_ERR_STATE; break; }
    }
    h->in_ply_loop = false;
    if (fold_pending) hipStreamWaitEvent(h->stream, h->ev_fold, 0);
    for (size_t i = 1; i < tpool.size(); ++i) { hipEventDestroy(tpool[i].first); hipEventDestroy(tpool[i].second); }
    if (rc) return rc;
    return finish_call(h, ngames, chain, k0, started, h->L, false, rollouts, ply, search_ms, t0, st);
}

// the end of a self-play / duel call, shared by the ply loop above and the persistent form below: the call's sample count and longest
// game, W / D / L, the chain's bookkeeping, the statistics, the faults
static int finish_call(agz_engine* h, int ngames, bool chain, unsigned long long k0, unsigned long long started, int in_flight, bool persist,
                       int64_t rollouts, int ply, double search_ms, std::chrono::steady_clock::time_point t0, agz_selfplay_stats* st) {
    unsigned long long hs[16];
    HIPCHK(h, hipMemcpy(hs, h->d_stats, sizeof hs, hipMemcpyDeviceToHost));
    {   // nsamples = sum of plies per game; kept (with the longest game) for agz_get_samples_packed
        std::vector<int32_t> np((size_t)(ngames < h->sample_games ? ngames : h->sample_games));
        std::vector<int8_t> res(chain ? np.size() : 0);
        const size_t cap = (size_t)h->sample_games, r0 = (size_t)h->sp_ring0;
        for (size_t a = 0; a < np.size();) {                                    // the call's games in ring order (one piece unless the ring wraps)
            const size_t r = (r0 + a) % cap, n = std::min(np.size() - a, cap - r);
            HIPCHK(h, hipMemcpy(np.data() + a, h->g_nplies + r, n * 4, hipMemcpyDeviceToHost));
            if (chain) HIPCHK(h, hipMemcpy(res.data() + a, h->g_result + r, n, hipMemcpyDeviceToHost));
            a += n;
        }
        int64_t n = 0; int mx = 0; for (int32_t x : np) { n += x; mx = x > mx ? x : mx; }
        h->sp_nsamples = n; h->sp_maxplies = mx;
        if (chain) {
            // the device counters of a chained call mix this call's games with the early ones of the next: W / D / L and the plies of
            // THIS call's games come from their own entries (mcts_gpu.jl:535: tot_length += round, the ply of the last move)
            hs[0] = hs[1] = hs[2] = hs[3] = 0;
            for (size_t i = 0; i < np.size(); ++i) { hs[res[i] == 1 ? 0 : (res[i] == 0 ? 1 : 2)] += 1; hs[3] += (unsigned long long)(np[i] > 0 ? np[i] - 1 : 0); }
            h->chain_live = true; h->chain_k0 = k0 + (unsigned long long)ngames; h->chain_started = persist ? hs[6] : started; h->chain_L = in_flight;
            h->chain_persist = persist;
        }
    }
    if (st) {
        memset(st, 0, sizeof *st);
        st->wins = (int64_t)hs[0]; st->draws = (int64_t)hs[1]; st->losses = (int64_t)hs[2]; st->total_plies = (int64_t)hs[3];
        st->faults = (int32_t)hs[4]; st->rollouts = rollouts; st->plies = ply; st->search_seconds = search_ms * 1e-3;   // (plies: rounds of the loop — with refilled slots more than the longest game)
        st->total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        st->nsamples = h->sp_nsamples;
    }
    if (hs[5]) { h->fail("%llu expansion(s) found a root with more legal actions than the rows by legal rank hold (legal_bound violated)", hs[5]); return AGZ_ERR_STATE; }
    if (hs[4]) { h->fail("%llu illegal sampled move(s) (\"faute\", mcts_gpu.jl:526-529)", hs[4]); return AGZ_ERR_ILLEGAL_MOVE; }
    return AGZ_OK;
}

// ---- one launch per call: the persistent self-play kernel (agz_selfplay_small.hpp) -------------------------------------------------
// the wide-trunk form with ONE 128-game workgroup per CU (k_selfplay_big4): built for the game shape, every workgroup resident, the 128-row
// activation tile (shared with the eight waves' tree tables) + flags + a work list of at least 64 entries per wave inside a CU's LDS
static bool use_big4(const agz_engine* h) {
    const DevNet& n = h->net[0];
    if (!h->k_persist_big4 || !n.wbig || n.H != 512 || h->big4 == 0) return false;
    if ((h->Lmax + 127) / 128 > h->cus) return false;
    if (h->big4 < 0 && h->Lmax <= 64 * h->cus) return false;                    // (up to 64 slots per CU the 8-lane form has a CU to itself at 256 registers)
    const int big_rowb = (2 * std::max(n.H, 32 * n.k0r) + 255) & ~255;
    const size_t xch_off = (std::max((size_t)8 * (size_t)eager_lds_layout(h->V, 16, 4 * h->big4_kpl).total, (size_t)128 * big_rowb) + 15) & ~(size_t)15;
    return xch_off + 4 * 144 + 16 + 8 * 256 <= (size_t)(160 * 1024);
}
static bool persist_shape(const agz_engine* h) {
    const DevNet& n = h->net[0];
    if (h->persist == 0 || h->cfg.nn_mode != AGZ_NN_BF16 || !n.loaded) return false;
    if (h->V > 128 || (h->V & 3) != 0 || h->no_fused_nn || 8 * h->reg_kpl > h->LGS) return false;
    if ((h->Lmax + 63) / 64 > 2 * h->cus) return false;                        // every workgroup resident: two 64-game workgroups per CU
    // ... which also needs their LDS to fit side by side (V = 128 trees on a 128-wide trunk do not: the call then runs one launch per ply)
    if (n.H == 128) {
        if (!(h->k_persist || h->k_persist_nar) || !n.w16w) return false;
        const bool nar = h->k_persist_nar && h->narrow_mode >= 0;
        // (lane-groups per wave, games per wave: half of the groups in the sparse form of the narrow kernel)
        const int G = nar ? h->persist_nar_g : 8, NG = 64 / G, GPW = (nar && h->nar_sparse > 0 && h->k_persist_nar_sp) ? NG / 2 : NG;
        const int tw = (nar || h->persist_tw4) ? 4 : 8, gpwg = tw * GPW;
        const int wgs = (h->Lmax + gpwg - 1) / gpwg, per_cu = (wgs + h->cus - 1) / h->cus;
        if (per_cu > (gpwg == 32 ? 4 : 2)) return false;
        const size_t cu_lds = (size_t)(160 * 1024) / (size_t)per_cu;
        const size_t tree_lds = nar ? (size_t)eager_lds_layout(h->V, NG, G * h->persist_nar_kpl).total : h->reg_lds;
        const int rs = small_io_row_bytes(n);
        const size_t io_off = (std::max((size_t)tw * tree_lds, (size_t)gpwg * 2 * (n.H * 2 + 16)) + 15) & ~(size_t)15;
        return io_off + (size_t)tw * GPW * rs + 16 + (size_t)tw * 64 <= cu_lds;   // (flags, at least 16 work-list entries per wave; the next-word tables where they fit)
    }
    if (n.H == 512 && use_big4(h)) return true;
    if (n.H == 512) {
        if (!h->k_persist_big[0] || !n.wbig || h->big8 < 0) return false;
        const size_t cu_lds = (size_t)(160 * 1024) / (size_t)((h->Lmax + 63) / 64 <= h->cus ? 1 : 2);
        const int big_rowb = (2 * std::max(n.H, 32 * n.k0r) + 255) & ~255;
        const size_t xch_off = (std::max((size_t)8 * h->reg_lds, (size_t)8 * 8 * big_rowb) + 15) & ~(size_t)15;
        return xch_off + 4 * 144 + 16 + 8 * 64 <= cu_lds;
    }
    return false;
}

static int run_games_persist(agz_engine* h, int ngames, int V, float cpuct, int tau_plies, agz_selfplay_stats* st, int next_games) {
    const bool chain = next_games >= 0;
    const bool cont = chain && h->chain_live;
    const unsigned long long k0 = cont ? h->chain_k0 : 0ull;
    const unsigned long long pool_end = k0 + (unsigned long long)ngames + (unsigned long long)(chain ? next_games : 0);
    const long long want = chain ? (long long)ngames + next_games : (long long)ngames;
    // AGZ_RESERVE_CUS=n: the launch leaves the slots of n CUs (128 each) without a workgroup, so that the kernels of another stream — the RCCL
    // all-gather of the call before, agz_comm_* — find n CUs' worth of wave slots and LDS free for the whole launch instead of waiting for a
    // persistent workgroup to end
    const int LM = h->Lmax - h->reserve_slots;
    const int slots = want < LM ? (int)want : LM;
    if (ngames < 1) { h->fail("ngames=%d outside [1,%d]", ngames, h->sample_games); return AGZ_ERR_ARG; }
    if (ngames > slots && ngames > h->sample_games) { h->fail("ngames=%d exceeds the sample capacity of the engine (%d games: agz_config.sample_capacity_games)", ngames, h->sample_games); return AGZ_ERR_ARG; }
    if (chain && want > h->sample_games) { h->fail("agz_selfplay_chain: ngames + next_ngames = %lld exceeds the sample capacity of the engine (%d games: agz_config.sample_capacity_games)", want, h->sample_games); return AGZ_ERR_ARG; }
    if (chain && pool_end + (unsigned long long)h->cfg.game_id_base > 0xffffffffull) { h->fail("agz_selfplay_chain: game ids exhausted"); return AGZ_ERR_ARG; }
    auto t0 = std::chrono::steady_clock::now();
    unsigned long long started = 0, finished = 0;
    int rc;
    if (!cont) {
        rc = agz_set_roots(h, nullptr, 0, nullptr, slots); if (rc) return rc;   // Position() for every game (:479), ids game_id_base + slot
        HIPCHK(h, hipMemsetAsync(h->d_stats, 0, 16 * sizeof(unsigned long long), h->stream));
        HIPCHK(h, hipMemsetAsync(h->g_nplies, 0, (size_t)h->sample_games * 4, h->stream));
        HIPCHK(h, hipMemsetAsync(h->slot_ply, 0, (size_t)h->Lmax * 4, h->stream));
        HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->alive, 1, (size_t)slots, h->stream));
        if (slots < h->Lmax) HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)(h->alive + slots), 0, (size_t)(h->Lmax - slots), h->stream));   // (the reserved slots too)
        started = (unsigned long long)slots;
        HIPCHK(h, hipMemcpyAsync(h->d_stats + 6, &started, 8, hipMemcpyHostToDevice, h->stream));
        if (h->mq_buf && h->mq_dirty) {                                         // games an abandoned chain left on their way between workgroups are dropped
            HIPCHK(h, hipMemsetAsync(h->mq_buf, 0, (size_t)h->mq_cap * sizeof(MigEntry), h->stream));
            HIPCHK(h, hipMemsetAsync(h->mq_ctr, 0, 16, h->stream));
            h->mq_dirty = false;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        // the games the last call left in flight go on in their slots (alive[]) or wait in the migration queue; per-call counters start over; the games of this call that
        // finished early are already counted; the ring entries of the games that may start now are cleared; slots without a game take the
        // games that wait when the kernel starts
        HIPCHK(h, hipSetDevice(h->cfg.device));
        started = h->chain_started;
        const unsigned long long cap = (unsigned long long)h->sample_games;
        for (unsigned long long a = started; a < pool_end;) {
            const unsigned long long r = a % cap, n = std::min(pool_end - a, cap - r);
            HIPCHK(h, hipMemsetAsync(h->g_nplies + r, 0, (size_t)n * 4, h->stream));
            a += n;
        }
        {
            std::vector<int32_t> np((size_t)ngames);
            for (size_t a = 0; a < np.size();) {
                const size_t r = (size_t)((k0 + a) % cap), n = std::min(np.size() - a, (size_t)cap - r);
                HIPCHK(h, hipMemcpyAsync(np.data() + a, h->g_nplies + r, n * 4, hipMemcpyDeviceToHost, h->stream));
                a += n;
            }
            HIPCHK(h, hipStreamSynchronize(h->stream));
            finished = 0; for (int32_t x : np) finished += x > 0 ? 1ull : 0ull;
        }
        const unsigned long long z[10] = {0, 0, 0, 0, 0, 0, started, 0, finished, 0};
        HIPCHK(h, hipMemcpyAsync(h->d_stats, z, sizeof z, hipMemcpyHostToDevice, h->stream));
    }
    h->L = LM;                                                                  // the launch covers every slot (the reserved ones aside); alive[] says which hold a game
    h->need_reset = true; h->tree_kpr = 0;
    h->chain_live = false;
    h->sp_games = ngames; h->sp_nsamples = 0; h->sp_maxplies = 0;
    h->sp_k0 = (uint32_t)k0; h->sp_ring0 = chain ? (uint32_t)(k0 % (unsigned long long)h->sample_games) : 0u;
    h->cpuct = cpuct; h->training = 1; h->step = 0;
    HIPCHK(h, hipMemsetAsync(h->d_pacc, 0, 16 * sizeof(unsigned long long), h->stream));
    // ---- the launch: the search parameters of the 64-game workgroups of k_search_small / k_search_big (agz_search_actor), the ply step's,
    // the call's pool
    DevNet& n = h->net[0];
    const bool big = n.H == 512;
    const bool nar = !big && h->k_persist_nar && h->narrow_mode >= 0;           // few-action games: 4 lanes per tree, 16 trees per wave, four waves
    // age classes: stone-placing games on boards whose rows by legal rank are built, V large enough for the expansion's compaction buffer
    // (2 V >= 8 KPR floats of the lane-group's edge table), refilled slots (a game can only leave a slot that a new game takes)
    const bool age = !big && !nar && h->k_persist_age && h->age_on && !h->no_compact && 2 * h->V >= 8 * h->age_kpr && (chain || (long long)ngames > (long long)slots);
    if (!big && !nar && !h->k_persist) { h->fail("no persistent self-play kernel for this game shape"); return AGZ_ERR_UNSUPPORTED; }
    const bool tw4 = !big && !nar && h->persist_tw4;
    const bool big4 = big && use_big4(h);                                       // one 128-game workgroup per CU, 4 lanes per tree
    const bool nsp = nar && h->nar_sparse > 0 && h->k_persist_nar_sp;                // ... or 8 games per wave of 16 lane-groups, four waves per SIMD (round 6)
    const int G = nar ? h->persist_nar_g : (big4 ? 4 : 8), NG = 64 / G, GPW = nsp ? NG / 2 : NG, tw = (nar || tw4) ? 4 : 8, gpwg = tw * GPW;
    const unsigned wgs = (unsigned)((LM + gpwg - 1) / gpwg);
    PersistTail X; memset(&X, 0, sizeof X);
    fill_plypar(h, X.P, 0, tau_plies, false);
    X.P.L = LM;
    X.P.refill_total = (chain || (long long)ngames > (long long)slots) ? (uint32_t)pool_end : 0u;
    X.P.ring = chain ? 1 : 0; X.P.k_cur_end = (uint32_t)(k0 + (unsigned long long)ngames);
    X.ngames_cur = chain ? (uint32_t)ngames : 0u;
    X.acc = h->d_pacc;
    if (h->mq_buf && !big && (age || h->mq_dirty)) {                            // (a chain that began with age classes keeps the queue: games may wait in it)
        X.P.mq.ctr = h->mq_ctr; X.P.mq.buf = h->mq_buf; X.P.mq.mask = h->mq_cap - 1u;
        X.P.mq.backlog_max = (uint32_t)(h->age_backlog > 0 ? h->age_backlog : std::max(64, LM / 16));
        if (X.P.mq.backlog_max > h->mq_cap / 2u) X.P.mq.backlog_max = h->mq_cap / 2u;
        X.P.mq.age = (uint32_t)std::max(0, h->G.A - 8 * h->age_kpr);
        X.old16 = (uint32_t)h->age_old16; X.class_by_block = (h->age_by_block ? 1u : 0u) | (h->age_by_wave ? 2u : 0u);
        // (the ring must outlast the worst backlog: the bound plus one game per wave of the launch, agz_plystep.hpp)
        if ((unsigned long long)X.P.mq.backlog_max + (unsigned long long)wgs * (unsigned long long)tw >= (unsigned long long)h->mq_cap) {
            h->fail("migration queue of %u entries is too small for a backlog of %u games and %u waves", h->mq_cap, X.P.mq.backlog_max, wgs * (unsigned)tw);
            return AGZ_ERR_STATE;
        }
        h->mq_dirty = true;
    }
    const bool age_kernel = X.P.mq.buf != nullptr;
    TreePar T = h->tp;
    T.L = LM; T.slot0 = 0; T.step = 0; T.cpuct = cpuct; T.training = 1;
    T.fastdiv = fastdiv_range(h);
    T.inject = 0; T.capture = 0; T.rollout = 0; T.do_reset = 1; T.do_expand = 0; T.do_select = 1; T.last = 0;
    T.gpw = GPW;
    const size_t cu_lds_all = (size_t)(160 * 1024);
    int wgcu = 2;
    h->in_ply_loop = true;
    if (big) {
        PersistBigPar Q; memset(&Q, 0, sizeof Q);
        BigSearchPar& S = Q.S;
        S.T = T;
        const int big_rowb = (2 * std::max(n.H, 32 * n.k0r) + 255) & ~255;
        BigPar& B = S.B;
        B.planes = (const uint16_t*)h->planes; B.INP = n.INP; B.wh = n.wbig;
        B.whead = n.w16 + (size_t)(n.INP / 32) * (n.H / 16) * 512 + (size_t)n.T * (n.H / 32) * (n.H / 16) * 512;
        B.bias_head = n.bias_head; B.logits = h->logits; B.LGS = h->LGS; B.vout = h->v_eval;
        B.L = LM; B.T = n.T; B.A = h->G.A; B.AOP = n.AOP; B.K0R = n.k0r; B.ROWB = big_rowb;
        wgcu = (big4 || (int)wgs <= h->cus) ? 1 : 2;
        S.V = V; S.tree_lds = big4 ? eager_lds_layout(h->V, NG, G * h->big4_kpl).total : (int)h->reg_lds;
        S.xch_off = (int)((std::max((size_t)8 * (size_t)S.tree_lds, (size_t)gpwg * big_rowb) + 15) & ~(size_t)15);
        size_t shared = (size_t)S.xch_off + 4 * 144 + 16;                        // ... + the workgroup's two flag words
        Q.X = X; Q.X.flag_off = (int)shared - 16;
        const size_t cu_lds = cu_lds_all / (size_t)wgcu;
        S.nxw_off = place_nxw(shared, cu_lds, 8, NG, h->V);
        const size_t room = cu_lds > shared ? cu_lds - shared : 0;
        S.wl_off = (int)shared; S.wl_bytes = (int)std::min({(size_t)(NG * h->V * 4), (room / 8) & ~(size_t)15, (size_t)h->wl_lds_max});
        const size_t lds = shared + (size_t)8 * S.wl_bytes;
        if (lds > cu_lds) { h->in_ply_loop = false; h->fail("persistent self-play kernel: %zu bytes of LDS per workgroup", lds); return AGZ_ERR_UNSUPPORTED; }
        h->rd_rec_bytes = h->tp.rec_bytes;
        hipEventRecord(h->ev_ply0, h->stream);
        hipLaunchKernelGGL(big4 ? h->k_persist_big4 : h->k_persist_big[wgcu - 1], dim3(wgs), dim3(NB_THREADS), lds, h->stream, Q);
    } else {
        PersistPar Q; memset(&Q, 0, sizeof Q);
        SmallPar& S = Q.S;
        S.T = T;
        S.F.planes = (const uint16_t*)h->planes; S.F.INP = n.INP; S.F.w16 = n.w16w; S.F.bias_head = n.bias_head;
        S.F.logits = h->logits; S.F.LGS = h->LGS; S.F.vout = h->v_eval; S.F.L = LM; S.F.T = n.T; S.F.A = h->G.A; S.F.AOP = n.AOP;
        S.F.gpw = 0; S.F.tw = tw; S.F.rb = GPW;
        S.V = V; S.tree_lds = nar ? eager_lds_layout(h->V, NG, G * h->persist_nar_kpl).total : (int)h->reg_lds;
        const int rs = small_io_row_bytes(n);
        S.io_prowb = rs; S.io_lgs = rs / 4; S.io_bw = GPW * rs;
        S.io_off = (int)((std::max((size_t)tw * (size_t)S.tree_lds, (size_t)gpwg * 2 * (n.H * 2 + 16)) + 15) & ~(size_t)15);
        S.xch_off = S.io_off + tw * S.io_bw;                                       // (no helper waves here: nothing is exchanged)
        // ... + the workgroup's two flag words + the tree waves' next-word tables (agz_tree_eager.hpp NXL: 2 V bytes per game)
        size_t shared = (size_t)S.xch_off + 16;
        Q.X = X; Q.X.flag_off = S.xch_off;
        wgcu = ((int)wgs + h->cus - 1) / h->cus;
        const size_t cu_lds = cu_lds_all / (size_t)wgcu;
        S.nxw_off = place_nxw(shared, cu_lds, tw, NG, h->V);
        const size_t room = cu_lds > shared ? cu_lds - shared : 0;
        S.wl_off = (int)shared; S.wl_bytes = (int)std::min({(size_t)(NG * h->V * 4), (room / (size_t)tw) & ~(size_t)15, (size_t)h->wl_lds_max});
        const size_t lds = shared + (size_t)tw * S.wl_bytes;
        if (lds > cu_lds) { h->in_ply_loop = false; h->fail("persistent self-play kernel: %zu bytes of LDS per workgroup", lds); return AGZ_ERR_UNSUPPORTED; }
        h->rd_rec_bytes = nar ? (uint32_t)eager_rec_bytes(G * h->persist_nar_kpl, h->V) : h->tp.rec_bytes;
        hipEventRecord(h->ev_ply0, h->stream);
        hipLaunchKernelGGL(nar ? (nsp ? h->k_persist_nar_sp : h->k_persist_nar) : (tw4 ? (age_kernel ? h->k_persist_tw4_age : h->k_persist_tw4) : (age_kernel ? h->k_persist_age : h->k_persist)), dim3(wgs), dim3(64 * tw), lds, h->stream, Q);
    }
    hipEventRecord(h->ev_ply1, h->stream);
    const bool sleep = h->ply_sleep && h->ev_adv && hipEventRecord(h->ev_adv, h->stream) == hipSuccess;
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = sleep ? hipEventSynchronize(h->ev_adv) : hipSuccess;   // (a blocking event: the host thread sleeps through the call)
    if (le == hipSuccess) le = hipStreamSynchronize(h->stream);
    h->in_ply_loop = false;
    if (le != hipSuccess) { h->fail("persistent self-play launch failed: %s", hipGetErrorString(le)); return AGZ_ERR_HIP; }
    float ms = 0; hipEventElapsedTime(&ms, h->ev_ply0, h->ev_ply1);
    unsigned long long acc[16], mqc[2] = {0, 0};
    HIPCHK(h, hipMemcpy(acc, h->d_pacc, sizeof acc, hipMemcpyDeviceToHost));
    if (age_kernel) HIPCHK(h, hipMemcpy(mqc, h->mq_ctr, sizeof mqc, hipMemcpyDeviceToHost));
    const int waiting = (int)(mqc[0] - mqc[1]);                                 // games on their way between workgroups: in flight like those in slots
    h->age_ranked_searches += acc[4]; h->age_searches += acc[2]; h->age_pushed += acc[5];
    const int64_t rollouts = (int64_t)acc[2] * V;
#ifdef AGZ_BIG4STAMPS
    fprintf(stderr, "[big4stamps] cycles summed over waves: tree step %llu  barrier before the pass %llu  network pass %llu  barrier behind it %llu\n", acc[12], acc[13], acc[14], acc[15]);
    fprintf(stderr, "[big4stamps] inside the pass: planes + first barrier %llu  k-loops %llu  barrier %llu  epilogues %llu  barrier %llu  head %llu\n", acc[6], acc[7], acc[8], acc[9], acc[10], acc[11]);
#endif
#ifdef AGZ_PSTAMPS
    fprintf(stderr, "[pstamps] cycles summed over waves: flag/barrier %llu  search %llu  counters %llu  ply step %llu   (ply step share %.2f %%)\n", acc[8], acc[9], acc[10], acc[11],
            100.0 * (double)acc[11] / (double)(acc[8] + acc[9] + acc[10] + acc[11] + 1));
#endif
    if (big) {
        char b[320];
        if (big4) snprintf(b, sizeof b, "k_selfplay_big4<KPL=%d,H=512,G=4> (persistent: one launch per self-play call, ONE workgroup per CU loops over the plies of its 128 games)", h->big4_kpl);
        else snprintf(b, sizeof b, "k_selfplay_big<KPL=%d,H=512,WG=%d> (persistent: one launch per self-play call, a workgroup loops over the plies of its 64 games)", h->reg_kpl, wgcu);
        h->form_tree = b; h->form_nn = big4 ? "inside k_selfplay_big4 (mlp_big_body<512,8>)" : "inside k_selfplay_big (mlp_big_body<512,4>)";
    } else {
        char ab[96] = ""; if (age_kernel) snprintf(ab, sizeof ab, "; age classes: rows by legal rank KPR=%d in workgroups whose games are all at ply >= %u", h->age_kpr, X.P.mq.age);
        char b[320]; snprintf(b, sizeof b, "k_selfplay_small<KPL=%d,H=128,TW=%d,WV=%d,G=%d%s> (persistent: one launch per self-play call, a workgroup loops over the plies of its %d games%s)",
                              nar ? h->persist_nar_kpl : h->reg_kpl, tw, (nar && !nsp) ? 2 : 4, G, age_kernel ? ",AGE" : (nsp ? ",SPARSE" : ""), gpwg, ab); h->form_tree = b; h->form_nn = "inside k_selfplay_small (mlp_wave_body<128>)";
    }
    h->acc_p += acc[0]; h->acc_new += acc[1]; h->total_rollouts += (uint64_t)rollouts; h->cnt_live = false;
    h->tree_ms += ms; h->tree_busy_ms += ms; h->tree_launches += 1;
    const int rounds = (int)((acc[2] + (unsigned long long)LM - 1) / (unsigned long long)LM);   // searches per slot, rounded up
    if (!chain && waiting) { h->fail("persistent self-play: %d games left in the migration queue at the end of a call of its own", waiting); return AGZ_ERR_STATE; }
    return finish_call(h, ngames, chain, k0, started, (int)acc[3] + waiting, true, rollouts, rounds, (double)ms, t0, st);
}

int agz_selfplay(agz_engine* h, int ngames, int V, float cpuct, int tau_plies, agz_selfplay_stats* stats) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->net[0].loaded) { h->fail("no network loaded"); return AGZ_ERR_STATE; }
    return run_games(h, ngames, V, cpuct, tau_plies, 1, 0, false, stats);
}
int agz_selfplay_chain(agz_engine* h, int ngames, int next_ngames, int V, float cpuct, int tau_plies, agz_selfplay_stats* stats) {
    if (!h) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->net[0].loaded) { h->fail("no network loaded"); return AGZ_ERR_STATE; }
    if (next_ngames < 0) { h->fail("agz_selfplay_chain: next_ngames=%d", next_ngames); return AGZ_ERR_ARG; }
    return run_games(h, ngames, V, cpuct, tau_plies, 1, 0, false, stats, next_ngames);
}
int agz_duel(agz_engine* h, int ngames, int V, float cpuct, int tau_plies, int first, int64_t wdl[3]) {
    if (!h || !wdl) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->net[0].loaded || !h->net[1].loaded) { h->fail("agz_duel needs both network slots"); return AGZ_ERR_STATE; }
    agz_selfplay_stats st{};
    wdl[0] = wdl[1] = wdl[2] = 0;
    int rc = run_games(h, ngames, V, cpuct, tau_plies, 0, first ? 1 : 0, true, &st);
    if (rc == AGZ_OK || rc == AGZ_ERR_ILLEGAL_MOVE) { wdl[0] = st.wins; wdl[1] = st.draws; wdl[2] = st.losses; }
    return rc;
}

// ---- samples ----------------------------------------------------------------------------------------
static int pack_samples(agz_engine* h, void* dev_out, int64_t capacity_records, int64_t* n_out, bool wait);
int agz_get_samples_packed(agz_engine* h, void* dev_out, int64_t capacity_records, int64_t* n_out) { return pack_samples(h, dev_out, capacity_records, n_out, true); }
int agz_get_samples_packed_async(agz_engine* h, void* dev_out, int64_t capacity_records, int64_t* n_out) { return pack_samples(h, dev_out, capacity_records, n_out, false); }
static int pack_samples(agz_engine* h, void* dev_out, int64_t capacity_records, int64_t* n_out, bool wait) {
    if (!h || !n_out) return AGZ_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // the number of samples and the longest game are known since the generation ended (run_games); the order table is built on the
    // device into a buffer the engine keeps: no allocation, no copy of the per-game ply counts per call
    const int64_t n = h->sp_games > 0 ? h->sp_nsamples : 0;
    *n_out = n;
    if (!dev_out) return AGZ_OK;                         // size query
    if (n > capacity_records) { h->fail("sample buffer too small: %lld > %lld", (long long)n, (long long)capacity_records); return AGZ_ERR_ARG; }
    if (n == 0) return AGZ_OK;
    const int G = h->sp_games < h->sample_games ? h->sp_games : h->sample_games;
    hipLaunchKernelGGL(k_sample_order, dim3((unsigned)h->sp_maxplies), dim3(1024), 0, h->stream, (const int32_t*)h->g_nplies, G, h->d_order,
                       (unsigned long long*)nullptr, h->sp_ring0, (uint32_t)h->sample_games);
    PackPar T;
    T.A = h->G.A; T.VS = h->G.VS; T.FS = h->G.FS; T.max_plies = h->G.max_plies; T.rec_bytes = h->info.rec_bytes;
    T.game_id_base = h->cfg.game_id_base; T.s_boards = h->s_boards; T.s_policy = h->s_policy; T.s_move = h->s_move; T.s_net = h->s_net;
    T.g_nplies = h->g_nplies; T.g_result = h->g_result; T.g_final = h->g_final; T.order = h->d_order; T.n = n;
    T.ring0 = h->sp_ring0; T.cap = (uint32_t)h->sample_games; T.k0 = h->sp_k0;
    T.out = (uint8_t*)dev_out;
    hipLaunchKernelGGL(k_pack_samples, dim3((unsigned)std::min<int64_t>((n + 3) / 4, (int64_t)h->cus * 32)), dim3(256), 0, h->stream, T);
    HIPCHK(h, hipGetLastError());
    if (!wait) return AGZ_OK;                            // (the records are complete when the engine's stream gets here: agz_stream)
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->fail("k_pack_samples failed: %s", hipGetErrorString(e)); return AGZ_ERR_HIP; }
    return AGZ_OK;
}
int agz_get_samples(agz_engine* h, int8_t* state, float* policy, int8_t* player, float* value, int8_t* fstate,
                    uint32_t* game_id, int32_t* ply, int32_t* move) {
    if (!h) return AGZ_ERR_ARG;
    int64_t n = 0;
    int rc = agz_get_samples_packed(h, nullptr, 0, &n); if (rc) return rc;
    if (n == 0) return AGZ_OK;
    const size_t rb = (size_t)h->info.rec_bytes;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if ((size_t)n * rb > h->stage_cap) {                 // staging buffers are kept: a host loop allocates them once, not per generation
        hipFree(h->stage_dev); h->stage_dev = nullptr;
        if (h->stage_host) { hipHostFree(h->stage_host); h->stage_host = nullptr; }
        h->stage_cap = 0;
        const size_t cap = (size_t)n * rb + (size_t)n * rb / 8;
        HIPCHK(h, dmalloc(&h->stage_dev, cap));
        HIPCHK(h, hipHostMalloc((void**)&h->stage_host, cap, hipHostMallocDefault));
        h->stage_cap = cap;
    }
    rc = agz_get_samples_packed(h, h->stage_dev, n, &n); if (rc) return rc;
    if (hipMemcpyAsync(h->stage_host, h->stage_dev, (size_t)n * rb, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { h->fail("sample D2H failed"); return AGZ_ERR_HIP; }
    return agz_unpack_records(&h->info, h->stage_host, n, state, policy, player, value, fstate, game_id, ply, move);
}
// Host-only helper: n packed records (agz_get_samples_packed layout, in host memory) -> PoolSample-layout arrays (any may be NULL).
// No handle, no device: a host loop that copies the records of generation k to pinned memory on a side stream unpacks them here
// while generation k + 1 runs on the engine's stream.
int agz_unpack_records(const agz_game_info* info, const void* records, int64_t n, int8_t* state, float* policy, int8_t* player,
                       float* value, int8_t* fstate, uint32_t* game_id, int32_t* ply, int32_t* move) {
    if (!info || (!records && n > 0) || n < 0) return AGZ_ERR_ARG;
    const int A = info->A, VS = info->VS, FS = info->FS;
    const size_t rb = (size_t)info->rec_bytes;
    const uint8_t* const host = (const uint8_t*)records;
    auto unpack = [=](int64_t s0, int64_t s1) {          // records [s0, s1) -> the caller's arrays (PoolSample layout)
        for (int64_t s = s0; s < s1; ++s) {
            const uint8_t* r = host + (size_t)s * rb;
            if (game_id) memcpy(&game_id[s], r, 4);
            if (ply) memcpy(&ply[s], r + 4, 4);
            if (move) memcpy(&move[s], r + 8, 4);
            if (value) memcpy(&value[s], r + 12, 4);
            if (player) player[s] = (int8_t)r[16];
            if (policy) memcpy(policy + (size_t)s * A, r + 20, (size_t)A * 4);
            if (state) memcpy(state + (size_t)s * 2 * VS, r + 20 + 4 * A, (size_t)2 * VS);
            if (fstate) memcpy(fstate + (size_t)s * FS, r + 20 + 4 * A + 2 * VS, (size_t)FS);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), (n + 65535) / 65536);   // one thread per 64 K records, at most 16
    if (nt <= 1) unpack(0, n);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(unpack, n * t / nt, n * (t + 1) / nt);
        for (auto& x : th) x.join();
    }
    return AGZ_OK;
}

// Known-answer test hook for the game plugins as the DEVICE runs them (Game<FAM,NC>::canPlay / play / isOver of agz_games.hpp, the
// code every tree and ply kernel is instantiated from): breadth-first perft from Position() on the GPU.  nodes = positions after
// exactly `depth` plies (finished games are not extended); terminal[0..2] = finished games met at any ply <= depth with result
// +1 / 0 / -1 — the usual perft definition (the test suite's CPU checker uses the same one), so published counts (Othello 4, 12, 56, 244, 1396, 8200, ...; TicTacToe's
// 255168 games; Connect4 7^d ...) pin the device code directly.
int agz_perft(const agz_config* cfg, int depth, int64_t* nodes, int64_t terminal[3]) {
    if (!cfg || !nodes || depth < 0 || depth > 64) return AGZ_ERR_ARG;
    GamePar P;
    if (make_game_par(cfg->game, cfg->n, cfg->nvict, P) != 0) return AGZ_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || cfg->device < 0 || cfg->device >= ndev) { g_create_error = "agz_perft: no such HIP device (libagz has no CPU path)"; return AGZ_ERR_HIP; }
    if (hipSetDevice(cfg->device) != hipSuccess) return AGZ_ERR_HIP;
    typedef void (*perft_fn)(const GamePar, const Pos*, unsigned long long, Pos*, unsigned long long, unsigned long long*, int);
    perft_fn k = nullptr;
#define X(F, R, C) if (P.fam == F && P.NC == C) k = k_perft_level<F, C>;
    AGZ_COMBOS(X)
#undef X
    if (!k) return AGZ_ERR_UNSUPPORTED;
    const unsigned long long cap = 1ull << 22;                             // positions per stored level (320 MB per buffer)
    Pos *buf[2] = {nullptr, nullptr}; unsigned long long* cnt = nullptr;
    int rc = AGZ_OK;
    if (hipMalloc((void**)&buf[0], cap * sizeof(Pos)) != hipSuccess || hipMalloc((void**)&buf[1], cap * sizeof(Pos)) != hipSuccess ||
        hipMalloc((void**)&cnt, 6 * sizeof(unsigned long long)) != hipSuccess) rc = AGZ_ERR_NOMEM;
    unsigned long long hc[6] = {0, 0, 0, 0, 0, 0};
    if (rc == AGZ_OK) {
        Pos root; memset(&root, 0, sizeof root);
        for (int i = 0; i < 3; ++i) { root.p[i] = P.start_p[i]; root.o[i] = P.start_o[i]; root.lg[i] = P.start_lg[i]; }
        root.player = (int8_t)P.start_player; root.aux = (int8_t)P.start_aux;
        hipMemcpy(buf[0], &root, sizeof root, hipMemcpyHostToDevice);
        hipMemset(cnt, 0, 6 * sizeof(unsigned long long));
        unsigned long long n = 1;
        for (int d = 0; d <= depth && n > 0; ++d) {
            const int remaining = depth - d;
            hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, P, (const Pos*)buf[d & 1], n, buf[(d + 1) & 1], cap, cnt, remaining);
            if (hipMemcpy(hc, cnt, sizeof hc, hipMemcpyDeviceToHost) != hipSuccess) { rc = AGZ_ERR_HIP; break; }
            if (hc[5]) { g_create_error = "agz_perft: a level exceeds the position buffer"; rc = AGZ_ERR_NOMEM; break; }
            n = hc[4];
            unsigned long long z = 0; hipMemcpy(cnt + 4, &z, 8, hipMemcpyHostToDevice);
            if (remaining <= 1) break;
        }
    }
    hipFree(buf[0]); hipFree(buf[1]); hipFree(cnt);
    if (rc != AGZ_OK) return rc;
    *nodes = (int64_t)hc[0];
    if (terminal) { terminal[0] = (int64_t)hc[1]; terminal[1] = (int64_t)hc[2]; terminal[2] = (int64_t)hc[3]; }
    return AGZ_OK;
}

}  // extern "C"

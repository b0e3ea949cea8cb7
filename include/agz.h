/*
 * agz.h — C ABI of libagz: MI355X-native batched AlphaZero self-play search.
 *
 * Drop-in boundary for the mcts_gpu.jl / selfplay.jl hot path of fabricerosay/AlphaGPU.
 * Every entry point names the reference interface it replaces (file:line under /root/reference).
 * Plain pointers and sizes only; the handle owns all device memory; host buffers are caller-owned.
 * One handle = one device + one HIP stream; a handle is not thread-safe; handles are independent.
 * All functions return 0 on success or a negative agz_status; agz_last_error() gives the message.
 *
 * Index conventions: actions, nodes and slots are 0-based (reference index k <-> k-1).
 * Host arrays are row-major with the game/slot index slowest, which is byte-identical to the
 * reference's column-major (A,L) / (2VS,L) Julia arrays.
 */
#ifndef AGZ_H
#define AGZ_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct agz_engine agz_engine;

enum agz_status {
    AGZ_OK = 0, AGZ_ERR_ARG = -1, AGZ_ERR_HIP = -2, AGZ_ERR_STATE = -3, AGZ_ERR_NOMEM = -4,
    AGZ_ERR_ILLEGAL_MOVE = -5,   /* the reference's "faute" abort, mcts_gpu.jl:526-529 */
    AGZ_ERR_UNSUPPORTED = -6
};

/* game plugins: Gobang.jl, 4IARow.jl, Hex.jl, Reversi8x8.jl, Reversi6x6.jl */
enum agz_game_kind { AGZ_GOBANG = 0, AGZ_CONNECT4 = 1, AGZ_HEX = 2, AGZ_REVERSI8 = 3, AGZ_REVERSI6 = 4,
                     AGZ_EXTRA_GAME = 5 /* a game plugged in at build time: -DAGZ_EXTRA_GAME_HPP, INTEGRATION.md "Adding a game" */ };

/* network evaluation modes */
enum agz_nn_mode {
    AGZ_NN_BF16 = 0,   /* bf16 MFMA contraction, fp32 accumulate (throughput mode) */
    AGZ_NN_EXACT = 1   /* fp32, k-ordered fma chains + source-order softmax: bit-identical to the CPU oracle */
};

typedef struct {
    int32_t  game;          /* agz_game_kind */
    int32_t  n;             /* board size N (Gobang <= 13, Hex <= 12); mainGobang.jl:24 `const N` */
    int32_t  nvict;         /* stones in a row to win (Gobang); mainGobang.jl:26 `const Nvict` */
    int32_t  max_games;     /* Lmax: slots allocated (mcts_gpu.jl:350 init(positions,visits)) */
    int32_t  max_visits;    /* V: tree nodes per slot (= rollouts per move), <= 256 */
    int32_t  device;        /* HIP device ordinal */
    uint64_t seed;          /* Philox key: search uniforms (CUDA.rand, mcts_gpu.jl:397) and move sampling (:520) */
    uint32_t game_id_base;  /* global id of slot 0; results depend on game ids, never on slot/GPU placement */
    int32_t  nn_mode;       /* agz_nn_mode */
    int32_t  sample_capacity_games; /* games whose samples are retained by agz_selfplay (0 = max_games); the most games one agz_selfplay call may play */
    int32_t  reserved[3];
} agz_config;

typedef struct {            /* constants every game module exports: Gobang.jl:2,8-11 */
    int32_t A;              /* maxActions */
    int32_t VS;             /* VectorizedState */
    int32_t FS;             /* FeatureSize */
    int32_t ML;             /* maxLengthGame */
    int32_t max_plies;      /* hard bound on plies per game (sample slots per game) */
    int32_t pos_image_bytes;/* sizeof(Position) in Julia: 104 or 152 */
    int32_t rec_bytes;      /* bytes of one packed sample record (agz_get_samples_packed) */
    int32_t reserved;
} agz_game_info;

/* positions handed over in the reference's own memory image (Vector{Position}) or as compact 80-byte
 * records {u64 bplayer[3], bopponent[3], legalplay[3]; i8 player, aux; pad[6]} */
enum agz_pos_format { AGZ_POS_JULIA = 0, AGZ_POS_COMPACT = 1 };

int  agz_query_game(const agz_config *cfg, agz_game_info *out);
int  agz_create(const agz_config *cfg, agz_engine **out);          /* mcts_gpu.jl:342-357 init */
void agz_destroy(agz_engine *h);
const char *agz_last_error(const agz_engine *h);                   /* NULL handle: last create error */
int  agz_get_info(const agz_engine *h, agz_game_info *out);
/* Replaces the Philox key of every later search / self-play / duel call of this handle.  The reference draws fresh
 * CUDA.rand / StatsBase randomness on every call (mcts_gpu.jl:397,520): a host loop that keeps one engine alive passes a
 * new seed per generation (the Python and Julia wrappers do), a parity test keeps it fixed. */
int  agz_set_seed(agz_engine *h, uint64_t seed);

/* convert_back(net) -> snetwork2 (DenseNet.jl:331-333, 279-286).  Host fp32, Flux (out,in) column-major:
 * W0 H x in, Wres T consecutive H x H blocks, Wp A x H, bp A, Wv 1 x H, bv 1.  in = 2*VS. */
int  agz_set_network(agz_engine *h, int H, int T, const float *W0, const float *Wres,
                     const float *Wp, const float *bp, const float *Wv, const float *bv);
/* The tag (0..255) stored with every sample that later self-play searches produce: byte 17 of a packed record (agz_get_samples_packed),
 * `net` of agz_get_sample_tags.  A host loop that changes the network between the calls of a chain (agz_selfplay_chain) numbers its
 * networks with it: the games a call starts early for the NEXT call play their first plies with the network of the running call
 * (the reference plays every game of a generation with one actor, selfplay.jl:34-56), and the tag of each sample says which.  Default 0. */
int  agz_set_network_tag(agz_engine *h, uint32_t tag);
/* second actor for duels (mcts_gpu.jl:581 mcts(actor1,actor2,...)); which = 0 or 1 */
int  agz_set_network_slot(agz_engine *h, int which, int H, int T, const float *W0, const float *Wres,
                          const float *Wp, const float *bp, const float *Wv, const float *bv);
/* Flux glorot_uniform / zero-bias random init from Philox (DenseNet.jl:193-198 ressimplesf): host helper */
int  agz_init_weights(uint64_t seed, int in, int H, int T, int A,
                      float *W0, float *Wres, float *Wp, float *bp, float *Wv, float *bv);

/* re_init(positions, vnodes, L, ...) mcts_gpu.jl:359-373 (+ Position() for NULL positions :479).
 * game_ids may be NULL (slot i -> game_id_base + i). */
int  agz_set_roots(agz_engine *h, const void *positions, int format, const uint32_t *game_ids, int L);

/* mcts_single(actor, visits, ..., L; training, cpuct) mcts_gpu.jl:376-462.  step = ply index (keys the uniforms). */
int  agz_search(agz_engine *h, int V, float cpuct, int training, uint32_t step);
int  agz_search_actor(agz_engine *h, int which, int V, float cpuct, int training, uint32_t step);

/* ---- stepwise form of mcts_single for teacher-forced parity (one call per reference kernel group) ---- */
int  agz_search_begin(agz_engine *h, float cpuct, int training, uint32_t step);   /* :380-387 */
int  agz_rollout_select(agz_engine *h, uint32_t rollout, int last);               /* :397-407 kdescendTree! + decoder */
int  agz_rollout_eval(agz_engine *h);                                             /* :414-417 actor + softmax! */
int  agz_get_eval(agz_engine *h, float *prior /*[L][A]*/, float *v /*[L]*/);      /* softmaxed priors as used */
int  agz_inject_eval(agz_engine *h, const float *prior, const float *v);          /* replaces the actor output */
int  agz_get_logits(agz_engine *h, float *logits /*[L][A]*/, float *v /*[L]*/);   /* raw actor output of the last STAND-ALONE network launch
                                                                                     (agz_rollout_eval; DenseNet.jl:294-304, before softmax!);
                                                                                     the one-launch searches keep logits in LDS */
int  agz_rollout_expand_backup(agz_engine *h);                                    /* :424-431 expand + backUp */
int  agz_search_end(agz_engine *h);                                               /* :441-444 decoder_roots, copy_pol */

/* ---- results of the last search ---- */
int  agz_get_policy(agz_engine *h, float *out);        /* policy_final [L][A]   (copy_pol :330-339) */
int  agz_get_batch(agz_engine *h, float *out);         /* root planes  [L][2VS] (decoder_roots :225-246) */
int  agz_get_leaf_batch(agz_engine *h, float *out);    /* leaf planes  [L][2VS] (decoder :202-223) */
int  agz_get_root_visits(agz_engine *h, float *out);   /* visits[:,1,:] [L][A] */
int  agz_get_root_q(agz_engine *h, float *out);        /* q[:,1,:]      [L][A] */
int  agz_get_leaf(agz_engine *h, int32_t *out);        /* leaf          [L] */
int  agz_get_node_count(agz_engine *h, int32_t *out);  /* newindex      [L] */
int  agz_get_counters(agz_engine *h, uint64_t *sum_p, uint64_t *sum_new, uint64_t *rollouts); /* for the roofline */

/* ---- whole-generation self-play on the device: mcts(actor,visits,ngames,buffer;cpuct) mcts_gpu.jl:477-579 ---- */
typedef struct {
    int64_t nsamples;       /* samples produced (= sum of plies over games) */
    int64_t total_plies;    /* "tot_length" :535 */
    int64_t wins, draws, losses;      /* res == 1 / 0 / -1  :541-547 */
    int64_t rollouts;       /* sum over plies of L_ply * V */
    int32_t plies;          /* number of lock-step rounds played */
    int32_t faults;         /* illegal sampled moves ("faute") */
    double  search_seconds; /* HIP-event time inside mcts_single ("temps mcts" :566) */
    double  total_seconds;  /* wall time of the call */
} agz_selfplay_stats;
/* ngames may exceed max_games (up to sample_capacity_games): the first max_games games start together and a slot whose game has ended
 * takes the next game that has not started yet, so that every search runs on a full batch.  Each game's samples are exactly those of a
 * lock-step run over ngames slots (results are keyed by game id and the game's own ply, never by slot or by the round a game started in);
 * stats->plies then counts the rounds of the loop. */
/* Execution form (chosen by the engine, results never depend on it): one launch per ply with the ply step in kernels of its own
 * (k_advance / k_scan_alive / k_compact: lock-step calls, small engines, exact mode), or — calls with refilled slots on engines of more than
 * 96 slots per CU (64 for 512-wide trunks), bf16 mode — ONE launch per call in which every workgroup loops over the plies of its own 64 games
 * (k_selfplay_small / k_selfplay_big; AGZ_PERSIST): stats->plies is then the searches per slot rounded up, stats->search_seconds the launch. */
int  agz_selfplay(agz_engine *h, int ngames, int V, float cpuct, int tau_plies, agz_selfplay_stats *stats);
/* A CHAIN of self-play calls (a host loop that calls mcts(actor, visits, ngames, buffer) generation after generation, selfplay.jl:34):
 * like agz_selfplay, but the caller says how many games its NEXT call will play.  Game ids run on from call to call (game k of the chain
 * has id game_id_base + k, this call returns games k0 .. k0 + ngames - 1), and while this call's games run out the slots that come
 * free start up to next_ngames games of the next call(s) instead of idling; they stay in flight when the call returns — as soon as its own
 * ngames games are over — and the next call of the chain goes on with them.  (next_ngames may exceed the size of the next call: the games
 * after it then start early too.  A host loop of SHORT calls — one generation each — needs that: a call's longest games take two
 * generations' worth of searches to finish, and the slots that come free meanwhile must have games to take.)  Only the last call of a chain (next_ngames = 0) ends on a
 * batch that runs out.  Every game's samples are those of one lock-step run over all the chain's games with the engine's seed (keyed by
 * game id and the game's own ply): the seed must stay the same while games are in flight (agz_set_seed fails otherwise), the network may
 * change between calls — the games started early for the next call are then searched by the OLD network on their first plies (the
 * reference plays a generation with one actor, selfplay.jl:34-56): agz_set_network_tag makes that visible per sample, next_ngames = 0
 * avoids it.  Needs sample_capacity_games >= ngames + next_ngames; fetch a call's samples before the next call of the chain
 * (their storage is reused).  agz_selfplay, agz_duel, agz_set_roots end a chain (games in flight are dropped).
 * stats: nsamples / wins / draws / losses / total_plies of THIS call's games; rollouts, plies, seconds of the work done inside the call. */
int  agz_selfplay_chain(agz_engine *h, int ngames, int next_ngames, int V, float cpuct, int tau_plies, agz_selfplay_stats *stats);
/* duelnetwork half: mcts(actor1,actor2,visits,ngames;cpuct) :581-651; first = actor to move at ply 0 */
int  agz_duel(agz_engine *h, int ngames, int V, float cpuct, int tau_plies, int first, int64_t wdl[3]);

/* Samples of the last agz_selfplay in PoolSample order (ply-major, then game id): mainGobang.jl:34-82.
 * Any pointer may be NULL.  state [n][2VS] i8, policy [n][A] f32, player [n] i8, value [n] f32,
 * fstate [n][FS] i8, game_id [n] u32, ply [n] i32, move [n] i32. */
int  agz_get_samples(agz_engine *h, int8_t *state, float *policy, int8_t *player, float *value,
                     int8_t *fstate, uint32_t *game_id, int32_t *ply, int32_t *move);
/* Packed records in DEVICE memory for the RCCL all-gather: writes n records of rec_bytes
 * (see agz_game_info) to dev_out; returns n via *n_out.  Record: {u32 game_id, i32 ply, i32 move, f32 value,
 * i8 player, u8 net_tag, i8 pad[2], f32 policy[A], i8 state[2VS], i8 fstate[FS], pad to 16 B}   (net_tag: agz_set_network_tag). */
int  agz_get_samples_packed(agz_engine *h, void *dev_out, int64_t capacity_records, int64_t *n_out);
/* The same without waiting: the pack kernels are queued on the engine's stream (agz_stream) and the call returns; the records are complete
 * once that stream has reached this point (an event recorded on it, agz_synchronize, or any later blocking call of the handle). */
int  agz_get_samples_packed_async(agz_engine *h, void *dev_out, int64_t capacity_records, int64_t *n_out);
/* Host-only helper (no handle, no device): n packed records in HOST memory -> the PoolSample-layout arrays of agz_get_samples
 * (push_buffer / update_buffer, mainGobang.jl:54-80).  Lets a host loop copy the records of generation k to pinned memory on a side
 * stream and unpack them while generation k+1 runs. */
int  agz_unpack_records(const agz_game_info *info, const void *records, int64_t n, int8_t *state, float *policy, int8_t *player,
                        float *value, int8_t *fstate, uint32_t *game_id, int32_t *ply, int32_t *move);

/* ---- the exchange step of a SHARDED generation: RCCL all-gather, over xGMI, of the packed sample records -------------------
 * One engine per GPU plays its own shard of game ids (agz_config.game_id_base = rank x games per rank; nothing is exchanged while games
 * run: every uniform is keyed by the global game id, so the union of the shards IS the single-GPU run over all games).  At the end of a
 * call the ranks gather their packed records (agz_get_samples_packed layout) and every rank holds the generation's samples, which
 * mcts(actor, visits, ngames, buffer) pushes into ONE PoolSample (mcts_gpu.jl:513-516; caller selfplay.jl:34).  The reference has no
 * counterpart (single device).  RCCL is bound at run time (dlopen: AGZ_RCCL_LIB, librccl.so.1); the host ships the 128-byte id from one
 * rank to the others by its own means (a file, a socket, MPI, torch.distributed).
 * Memory, allocated once at agz_comm_create: 2 x (1 + world) x (32 + capacity_records x rec_bytes) bytes per rank (two slots: the
 * exchange of call k overlaps call k + 1).  8 ranks x one Gobang 9x9 generation (32768 games, ~39 plies, 592-byte records: ~0.76 GB per
 * rank) = 6.1 GB gathered per slot: size capacity_records by the records of ONE call (16 GB of gathered records hold two generations).
 * BASELINE config 5 (Reversi 8x8, 8 ranks x 32768 games, one generation per call; measured on an MI355X, scratch/mem_cfg5.py): 480-byte records, at most
 * 128 plies per game (passes included; ~60 on average) -> capacity 32768 x 128 records = 2.01 GB per block, 2 x 9 x 2.01 = 36.2 GB of exchange
 * buffers per rank next to the engine's 7.0 GB (node records 2.4 GB, the sample store of three generations in flight, states, planes): 14 % of the
 * 309 GB the device reports.  A capacity of 70 plies per game (bench.py --exchange-plies 70: a call that outgrows it fails on every rank alike) is 19.8 GB.
 * A rank whose self-play call FAILED (an illegal sampled move, any error) must still enter the exchange — the others would wait inside
 * ncclAllGather for ever: agz_comm_post_status hands its return code to the next collective (such a rank sends no records), every
 * rank reads all codes after the wait (agz_comm_get_statuses) and they fail, or go on, together. */
typedef struct agz_comm agz_comm;
#define AGZ_COMM_ID_BYTES 128
int  agz_comm_unique_id(void *id /* [AGZ_COMM_ID_BYTES] */);       /* ncclGetUniqueId: on ONE rank */
int  agz_comm_create(agz_engine *h, int rank, int world, const void *id, int64_t capacity_records, agz_comm **out);   /* ncclCommInitRank on the engine's device; collective */
void agz_comm_destroy(agz_comm *c);
const char *agz_comm_last_error(const agz_comm *c);                /* NULL: the last agz_comm_create / agz_comm_unique_id error of this thread */
/* Blocking form, as SURVEY §8e states it: all-gather of the ranks' record counts, then of the records of the engine's last self-play
 * call padded to the largest count.  counts[world] (may be NULL) receives every rank's record count. */
int  agz_allgather_samples(agz_engine *h, agz_comm *c, int64_t *counts);
/* Pipelined form: ONE collective per call, issued without waiting for any other rank — every rank sends 32 + send_records x rec_bytes
 * bytes (its record count travels in the 32-byte header), send_records being a number ALL ranks agree on (0 = the capacity; a host loop
 * predicts it from the counts of the calls before).  At most two collectives in flight.  _wait returns the counts of the OLDEST one; if a
 * rank produced more than send_records records, every rank — they all see the same counts — gathers the rest in a second, blocking
 * collective inside _wait (from the exchange's own copy of the records: the engine may have played on).  *max_count: the largest count. */
int  agz_allgather_samples_start(agz_engine *h, agz_comm *c, int64_t send_records);
int  agz_allgather_samples_wait(agz_comm *c, int64_t *counts, int64_t *max_count);
/* (_start does not block the host: the pack kernels run on the engine's stream, the exchange's stream waits for them through an event, the
 *  32-byte header {record count, send_records, status} is copied from pinned memory.  _wait blocks until the oldest collective is complete.) */
/* The status word this rank sends with its NEXT collective (default 0; reset to 0 when sent): the return code of its self-play call.
 * A rank that posts a status != 0 takes part with a record count of 0. */
int  agz_comm_post_status(agz_comm *c, int status);
/* statuses[world]: the status words of the exchange last waited for (agz_allgather_samples_wait / agz_allgather_samples). */
int  agz_comm_get_statuses(agz_comm *c, int32_t *statuses);
/* agz_comm_post_status + agz_allgather_samples + agz_comm_get_statuses in one call: what a rank of a sharded run calls after its
 * agz_selfplay* call WHATEVER that call returned (`status` = its return code).  Returns AGZ_OK when the exchange itself worked. */
int  agz_allgather_samples_status(agz_engine *h, agz_comm *c, int status, int64_t *counts, int32_t *statuses);
/* The records of rank `rank` from the exchange last waited for: n records from record `first` on into host memory (-> agz_unpack_records),
 * or their address in device memory (valid until the next-but-one agz_allgather_samples_start; NULL if the exchange needed its second
 * collective: the records are then in two pieces). */
int  agz_comm_fetch_records(agz_comm *c, int rank, void *host_dst, int64_t first, int64_t n);
const void *agz_comm_records_device(agz_comm *c, int rank);

/* Known-answer test hook: breadth-first perft from Position() computed ON THE DEVICE with the game plugin code the kernels are
 * instantiated from (canPlay / play / isOver: Gobang.jl:25-70, 4IARow.jl:25-81, Hex.jl:37-67, Reversi8x8.jl:84-121).  *nodes = positions
 * after exactly `depth` plies (finished games are not extended); terminal[0..2] (may be NULL) = finished games met at any ply <= depth
 * with result +1 / 0 / -1.  Uses cfg->game, n, nvict, device. */
int  agz_perft(const agz_config *cfg, int depth, int64_t *nodes, int64_t terminal[3]);

/* stream / timing plumbing */
void *agz_stream(agz_engine *h);                       /* the engine's hipStream_t */
int  agz_synchronize(agz_engine *h);
/* HIP-event timings (ms) accumulated since the last reset: [0] tree kernels, [1] network kernels, [2] launches */
int  agz_get_kernel_times(agz_engine *h, double *tree_ms, double *nn_ms, int64_t *tree_launches, int reset);
/* with sub-batch chains several tree-kernel launches run side by side: busy time (ms) = length of the union of the
 * launch intervals since the last reset of agz_get_kernel_times (== tree_ms when launches never overlap) */
int  agz_get_tree_busy_ms(agz_engine *h, double *busy_ms);
/* leaves evaluated by the stand-alone network launches that agz_get_kernel_times' nn_ms covers (since its last reset) */
int  agz_get_nn_leaves(agz_engine *h, uint64_t *leaves);
/* the persistent self-play kernels since the last reset of agz_get_kernel_times: out[0] = searches of a game (one per game and ply),
 * out[1] = those run with node rows by the root's legal rank (workgroups whose games were all old: the age classes of agz_selfplay_small.hpp),
 * out[2] = games that changed workgroup through the migration queue */
int  agz_get_age_stats(agz_engine *h, uint64_t out[3]);
/* names of the kernels the last search ran (the engine picks the execution form by batch size and network width) */
int  agz_get_search_form(agz_engine *h, char *tree_kernel, char *nn_kernel, int cap);
int  agz_set_profiling(agz_engine *h, int mode);       /* HIP events per launch: bit 0 tree kernel, bit 1 network kernel;
                                                          bit 2: instrument (events and agz_get_counters) every 4th search only */

/*
 * Environment switches read by agz_create (all optional; results never depend on them, the tests use them to reach every
 * kernel build at small sizes):
 *   AGZ_SMALL_MAXL=n       128-wide trunk: 16 games per workgroup of the one-launch search up to n games (default 8192), 32 above;
 *                          0 together with AGZ_SMALL4_MAXL=0 disables the one-launch form (two kernels per rollout)
 *   AGZ_SMALL4_MAXL=n      largest batch of the 32-games-per-workgroup one-launch search
 *   AGZ_SMALL4_OCC=0|1|2   force the 2 / 3 / 4 workgroups-per-CU register budget of that kernel
 *   AGZ_SMALL_GPW=1..8    games per tree wave of the one-launch forms (sparse waves for small batches)
 *   AGZ_BIG_MAXL=n         512-wide trunk: one-launch search (k_search_big) up to n games (0 disables; from the default 16384 on the limit
 *                          is 128 games per CU: two 64-game workgroups per CU, see AGZ_BIG8)
 *   AGZ_CHAINS=k           two-kernel form: k sub-batches on parallel streams (default 2-3 from 12000 games)
 *   AGZ_REG3_MAX_WAVES=n   two-kernel form: largest grid that uses the 3-waves-per-SIMD build of the tree kernel
 *   AGZ_NN_WAVE_LT, AGZ_NN_WAVE_DEPTH   tile count / prefetch depth of k_mlp_wave
 *   AGZ_NO_FUSED_NN=1      per-layer network kernels (k_layer_bf16) and no one-launch search
 *   AGZ_BIG8=0             k_search_big above 32 games per CU: two 32-game workgroups per CU instead of one 64-game workgroup, and the
 *                          two-kernel form above 64 games per CU instead of two 64-game workgroups per CU
 *   AGZ_TW8=1|0            one-launch search beyond 96 games per CU: 64-game workgroups of eight waves for every board shape (1) or
 *                          for none (0); default: the 9x9 shapes they were measured on
 *   AGZ_NARROW=-1|0|4|2    one-launch search with narrow lane-groups (4 or 2 lanes per game tree: 16 / 32 trees per wave, workgroups of four tree
 *                          waves): never (-1), by batch size (0, default), or only that group width
 *   AGZ_NARROW_SPARSE=0    a few-action game (Connect4) at full batch — persistent self-play and the one-launch search above 64 games per CU —: sixteen games per
 *                          wave of sixteen 4-lane groups, two waves per SIMD (round 5's form) instead of EIGHT games per wave and four waves per SIMD, the
 *                          groups without a game on work items (default, round 6); =2: the one-launch search in that form at every batch size (tests)
 *   AGZ_NARROW_MINL=n      ... for batches of at least n games (tests: 0)
 *   AGZ_NARROW_OCC=0|1     ... force its two- / one-wave-per-SIMD register budget (tests)
 *   AGZ_NO_COMPACT=1       the ply loop keeps node rows by action on Gobang / Hex 9x9 (default: rows by the root's legal rank from ply 17 on,
 *                          same results; root statistics cannot be read back after such a search)
 *   AGZ_NO_FASTDIV=1       IEEE '/' everywhere in the tree kernel (agz_fastdiv.hpp off)
 *   AGZ_PLY_SPIN=1         ply loop: the host thread spins on the scan's host-visible word through the whole search instead of sleeping on a
 *                          blocking event until the ply's k_advance has run
 *   AGZ_RUN_AHEAD=n        ply loop with refilled slots: plies queued before the host waits for a ply's counters (default 8; 8 and 32 measure the same)
 *   AGZ_NO_HOST_FLAG=1     ply loop: fetch the number of games left with a copy + stream synchronisation instead of polling the
 *                          host-visible word the scan kernel publishes
 *   AGZ_BIG_MT=2|4|8       256 / 512-wide trunk, stand-alone network launches: 16-leaf tiles per workgroup (default by launch size)
 *   AGZ_PERSIST=1|0        self-play calls as ONE launch (k_selfplay_small: every workgroup loops over the plies of its own games, no kernel
 *                          boundary, no host wake-up per ply) wherever such a kernel exists — 128-wide trunk, bf16 mode, all slots resident —
 *                          (1), or never (0); default: calls that refill their slots (agz_selfplay with more games than slots,
 *                          agz_selfplay_chain) on engines of more than 96 slots per CU
 *   AGZ_PERSIST_TW=8       persistent self-play with 64-game workgroups of eight waves, two per CU, instead of 32-game workgroups of four (default)
 *   AGZ_BIG4=0|1           512-wide trunks: ONE 128-game workgroup per CU with 4 lanes per tree and the network pass on 128 leaves (k_selfplay_big4,
 *                          k_search_big4) never (0) / wherever it fits (1); default: above 64 games per CU, where the 8-lane form runs two 64-game
 *                          workgroups per CU at 128 registers
 *   AGZ_AGE=0              persistent self-play without age classes (every workgroup keeps node rows by action; nothing migrates)
 *   AGZ_AGE_WAVE=1         age classes: every WAVE of a workgroup picks rows by legal rank from its own eight games (default: the workgroup as a whole)
 *   AGZ_AGE_OLD16=n        ... n of 16 CU pairs prefer old games (default 8);  AGZ_AGE_CLASS=block: odd workgroups prefer old games (tests);
 *                          AGZ_AGE_BACKLOG=n: young-preferring workgroups keep their old games while n games wait in the migration queue
 *   AGZ_RCCL_LIB=path      the RCCL library agz_comm_* binds (default: an RCCL already in the process, librccl.so.1, /opt/rocm/lib/librccl.so.1)
 *   AGZ_WL_LDS_BYTES=n     one-launch forms: at most n bytes of LDS per tree wave for the work list of a rollout (the rest of
 *                          the list lives in global memory; default: what the resident workgroups leave free)
 *   AGZ_RESERVE_CUS=n      self-play calls use 128 n slots fewer than the engine has: the persistent launches then leave n CUs' worth of wave slots and LDS
 *                          free for the whole launch — room for the RCCL kernels of the exchange (agz_comm_*) that overlaps the next call
 *   AGZ_NXL=0              one-launch forms: the descent reads the next words from the node records (global memory) instead of the copy the tree
 *                          waves keep in LDS (16 bits per node and game, wherever the workgroup's LDS has the room; round 6, +6 % on the headline)
 */
#ifdef __cplusplus
}
#endif
#endif

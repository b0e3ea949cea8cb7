/*
 * agz_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, strict-IEEE (no FMA contraction, no fast-math), single-threaded
 * restatement of the fabricerosay/AlphaGPU self-play hot path:
 *   Bitboard.jl, Gobang.jl, 4IARow.jl, Hex.jl, Reversi8x8.jl, Reversi6x6.jl,
 *   mcts_gpu.jl (select / encode / expand / backup / mcts_single / mcts),
 *   DenseNet.jl snetwork2 forward, mainGobang.jl PoolSample, fast_mcts.jl.
 * Every function cites the reference file:line it follows.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures
 *  and cannot be executed here (no Julia).  The game rules are
 * pinned by independent known-answer tests (tests/test_oracle_games.py); the
 * search semantics are pinned only by this restatement and by hand-derived
 * micro-cases.  Randomness (CURAND / StatsBase in the reference, unseeded)
 * is replaced by the Philox4x32-10 streams defined below.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (alphagpu_amd/) never links or calls it.
 *
 * Conventions: actions, nodes and bit indices are 0-BASED here; reference
 * (Julia) index k corresponds to k-1.
 */
#ifndef AGZ_ORACLE_H
#define AGZ_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { AGZO_GOBANG = 0, AGZO_CONNECT4 = 1, AGZO_HEX = 2, AGZO_REVERSI8 = 3, AGZO_REVERSI6 = 4 };

typedef struct { uint64_t c[3]; } agzo_bb;                 /* Bitboard.jl:5-9 chunks */

typedef struct {                                           /* Gobang.jl:16-21, Hex.jl:16-21, Reversi8x8.jl:73-78 */
    agzo_bb bplayer, bopponent, legalplay;
    int8_t  player;
    int8_t  aux;                                           /* round (Gobang/Connect4) or lp (Hex) */
    int8_t  pad[6];
} agzo_pos;                                                /* 80 bytes */

typedef struct {
    int kind, n, nvict;
    int d1, d2, len;                                       /* bitboard dims / length */
    int A, VS, FS, ML;                                     /* maxActions, VectorizedState, FeatureSize, maxLengthGame */
} agzo_game;

typedef struct agzo_net_bf16 agzo_net_bf16;                /* weights rounded to bf16 (agzo_net_bf16_create) */
typedef struct {                                           /* DenseNet.jl:279-286 (snetwork2), weights in Flux (out,in) column-major */
    int in, H, T, A;
    const float *W0;                                       /* H x in  */
    const float *Wres;                                     /* T blocks of H x H */
    const float *Wp, *bp;                                  /* A x H, A */
    const float *Wv, *bv;                                  /* 1 x H, 1 */
    const agzo_net_bf16 *bf16;                             /* non-NULL: agzo_search / agzo_selfplay / agzo_duel evaluate the network
                                                              as the product's bf16 MFMA mode does (bit-level model) */
} agzo_net;

/* ---- games ---- */
int  agzo_game_init(agzo_game *g, int kind, int n, int nvict);
void agzo_pos_init(const agzo_game *g, agzo_pos *p);
int  agzo_can_play(const agzo_game *g, const agzo_pos *p, int a);
void agzo_play(const agzo_game *g, const agzo_pos *p, int a, agzo_pos *out);
int  agzo_is_over(const agzo_game *g, const agzo_pos *p, int *result);
int  agzo_bb_get(const agzo_bb *b, int bit);
long agzo_perft(const agzo_game *g, const agzo_pos *p, int depth, long *terminal);
/* raw bitboard ops exposed for the naive-model cross-check */
void agzo_bb_shift(const agzo_game *g, const agzo_bb *b, int op, agzo_bb *out); /* op: 0 right 1 left 2 down 3 up */
/* Julia memory image <-> agzo_pos (104 / 152 byte records) */
int  agzo_pos_image_bytes(const agzo_game *g);
void agzo_pos_to_image(const agzo_game *g, const agzo_pos *p, void *img);
void agzo_pos_from_image(const agzo_game *g, const void *img, agzo_pos *p);

/* ---- randomness ---- */
void  agzo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float agzo_uniform_search(uint64_t seed, uint32_t game_id, uint32_t step, uint32_t rollout, uint32_t depth); /* (0,1] */
float agzo_uniform_move(uint64_t seed, uint32_t game_id, uint32_t step);                                     /* [0,1) */
void  agzo_init_weights(uint64_t seed, int in, int H, int T, int A,
                        float *W0, float *Wres, float *Wp, float *bp, float *Wv, float *bv);

/* ---- network ---- */
float agzo_expf(float x);
void  agzo_encode(const agzo_game *g, const agzo_pos *p, float *planes);              /* mcts_gpu.jl:202-223 */
void  agzo_forward(const agzo_net *net, const float *planes, float *logits, float *v); /* DenseNet.jl:294-304 */
void  agzo_softmax(float *x, int n);                                                   /* mcts_gpu.jl:417 */
/* the same forward as the product's bf16 MFMA kernels compute it (bit-level model of v_mfma_f32_16x16x32_bf16, see .c) */
float agzo_mfma_dot(const uint16_t *a, const uint16_t *b, int K, float acc);          /* K multiple of 8, bf16 bit patterns */
agzo_net_bf16 *agzo_net_bf16_create(const agzo_net *net);
void  agzo_net_bf16_destroy(agzo_net_bf16 *n);
void  agzo_forward_bf16(const agzo_net_bf16 *n, const float *planes, float *logits, float *v);
float agzo_exp2_spec(float x);
void  agzo_softmax_bf16mode(float *x, int n);

/* ---- batched search (mcts_gpu.jl semantics) ---- */
typedef struct agzo_tree agzo_tree;
agzo_tree *agzo_tree_create(const agzo_game *g, int Lmax, int V);
void agzo_tree_destroy(agzo_tree *t);
void agzo_tree_set_roots(agzo_tree *t, const agzo_pos *positions, const uint32_t *game_ids, int L); /* re_init :359-373 */
void agzo_search_reset(agzo_tree *t);                                                  /* :380-387 */
void agzo_select(agzo_tree *t, uint64_t seed, uint32_t step, uint32_t rollout, float cpuct); /* :100-199 */
void agzo_set_reference_keying(int on);   /* test switch: the descent's uniform keyed by the VISITING rollout, as prob[cpt,i] in the reference (:178) */
void agzo_encode_leaves(agzo_tree *t, float *batch);                                   /* :202-223 */
/* expand / backup of rollout `rollout` also fix the uniform the next visit of every row they make samples with (see agzo_select) */
void agzo_expand(agzo_tree *t, const float *prior, int training, uint64_t seed, uint32_t step, uint32_t rollout); /* :250-302 */
void agzo_backup(agzo_tree *t, const float *v, uint64_t seed, uint32_t step, uint32_t rollout);                   /* :306-328 */
/* mcts_single :376-462.  If prior_inject/v_inject are non-NULL they hold V x L x A / V x L teacher-forced
 * network outputs (softmaxed priors) and the net is not evaluated.  If prior_capture/v_capture are non-NULL
 * the evaluated (softmaxed) outputs are stored there. */
void agzo_search(agzo_tree *t, const agzo_net *net, int V, float cpuct, int training,
                 uint64_t seed, uint32_t step,
                 const float *prior_inject, const float *v_inject,
                 float *prior_capture, float *v_capture);
/* getters (row-major [L][A] etc.) */
void agzo_get_policy(const agzo_tree *t, float *out);        /* policy_final  [L][A]   :330-339 */
void agzo_get_root_planes(const agzo_tree *t, float *out);   /* decoder_roots [L][2VS] :225-246 */
void agzo_get_root_visits(const agzo_tree *t, float *out);   /* visits[:,1,:] [L][A] */
void agzo_get_root_q(const agzo_tree *t, float *out);        /* q[:,1,:]      [L][A] */
void agzo_get_root_policy_row(const agzo_tree *t, float *out); /* policy[:,1,:] as of now */
void agzo_get_leaf(const agzo_tree *t, int *out);            /* leaf          [L] (0-based) */
void agzo_get_newindex(const agzo_tree *t, int *out);        /* nodes used    [L] */
long agzo_get_counters(const agzo_tree *t, long *sum_p, long *sum_new);

/* ---- self-play (mcts_gpu.jl:477-579 + PoolSample mainGobang.jl:34-82) ---- */
typedef struct {
    long   nsamples, capacity;
    int    A, VS, FS;
    int8_t *state;     /* [n][2VS] */
    float  *policy;    /* [n][A]   */
    int8_t *player;    /* [n]      */
    float  *value;     /* [n]      */
    int8_t *fstate;    /* [n][FS]  */
    uint32_t *game_id; /* [n]      */
    int32_t  *ply;     /* [n]      */
    int32_t  *move;    /* [n] chosen action (0-based) */
    long   wins, draws, losses, total_plies, faults;
} agzo_samples;
agzo_samples *agzo_samples_create(const agzo_game *g, long capacity);
void agzo_samples_destroy(agzo_samples *s);
int  agzo_selfplay(const agzo_game *g, const agzo_net *net, int ngames, int V, float cpuct,
                   int tau_plies, uint64_t seed, uint32_t game_id_base, agzo_samples *out);
/* ... with the actor chosen per (game, ply) (tags: [ngames][tag_stride] indices into nets; NULL: nets[0]) — a chain of calls whose
 * network changes between calls (include/agz.h agz_selfplay_chain, agz_set_network_tag) */
int  agzo_selfplay_tagged(const agzo_game *g, const agzo_net *const *nets, int nnets, const uint8_t *tags, int tag_stride, int ngames, int V,
                          float cpuct, int tau_plies, uint64_t seed, uint32_t game_id_base, agzo_samples *out);

/* ---- duel: mcts(actor1,actor2,visits,ngames;cpuct) mcts_gpu.jl:581-651 (training=false, sample over ALL
 * actions for round < tau_plies (15), argmax after; actor by ply parity).  first = 0: net1 moves first. ---- */
int  agzo_duel(const agzo_game *g, const agzo_net *net1, const agzo_net *net2, int ngames, int V, float cpuct,
               int tau_plies, uint64_t seed, uint32_t game_id_base, int first, long wdl[3],
               int32_t *moves, int max_plies, int32_t *nplies);

/* ---- CPU baseline: fast_mcts.jl single-tree search ---- */
void agzo_fmcts(const agzo_game *g, const agzo_net *net, const agzo_pos *pos, int readout, float c,
                uint64_t seed, uint32_t game_id, float *policy_out, float *value_out);
/* plays `ngames` games to the end with `readout` readouts per move on `threads` OpenMP threads;
 * returns total rollouts performed. */
long agzo_fmcts_selfplay(const agzo_game *g, const agzo_net *net, int ngames, int readout, float c,
                         int tau_plies, uint64_t seed, int threads, int max_plies);

#ifdef __cplusplus
}
#endif
#endif

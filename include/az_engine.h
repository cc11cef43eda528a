/*
 * az_engine.h - C ABI of the MI355X batched self-play engine.
 *
 * The reference has no FFI: its plug-in boundary is three Python modules
 * chosen by YAML keys (env_path / state_converter_path / gameplay_path,
 * AlphaZero/config/go.yaml:3-6) plus the evaluator/search/player contracts.
 * The Python modules in alphazero_amd/ keep those contracts and bind the
 * entry points below through ctypes; each entry point names the reference
 * interface it stands in for.
 *
 * Conventions: opaque handle, int status (0 = ok, <0 = AZ_ERR_*),
 * caller-owned buffers (host pointers unless the name ends in _dev), no
 * exceptions across the boundary, one host thread per handle (one handle per
 * GPU).  Actions are flat indices x*W + y; PASS is H*W.  A "slot" is one game
 * (state + the two search trees of its two players); a "side" is 0 for
 * player_1 (moves first) and 1 for player_2.
 */
#ifndef AZ_ENGINE_H
#define AZ_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_GO 0
#define AZ_REVERSI 1
#define AZ_MNK 2

#define AZ_OK 0
#define AZ_ERR_ARG (-1)
#define AZ_ERR_HIP (-2)
#define AZ_ERR_ILLEGAL_MOVE (-3)   /* the module-level IllegalMove of the env modules */
#define AZ_ERR_TREE_FULL (-4)      /* a tree reached tree_edge_cap / tree_block_cap, or the shared edge pool ran out of pages */
#define AZ_ERR_RNG_UNDERFLOW (-5)  /* a replay stream ran dry */
#define AZ_ERR_DEPTH (-6)          /* a playout went deeper than max_depth */
#define AZ_ERR_STATE (-7)          /* call not valid in the handle's current phase */
#define AZ_ERR_NO_WEIGHTS (-8)

#define AZ_MAX_CELLS 361
#define AZ_MAX_ACTIONS 362
#define AZ_CELL_PAD 384            /* per-board stride in bytes */
#define AZ_NPLANES 17              /* history_step*planes_per_step + additional_planes */

#define AZ_TOWER_FP32 0
#define AZ_TOWER_BF16 1
#define AZ_TOWER_F16 2   /* IEEE half precision: same matrix-core rate and bytes as bf16, 11 significant bits instead of 8 */

typedef struct az_engine az_engine;

typedef struct az_config {
    uint32_t struct_size;      /* sizeof(az_config) as the caller sees it.  az_engine_create refuses any other value than the
                                  library's own, so a binding written against another version of this header fails at the
                                  first call instead of being read past its end.  az_abi_sizes reports the library's sizes. */
    int32_t game;              /* AZ_GO / AZ_REVERSI / AZ_MNK  (config key env_path) */
    int32_t board_height, board_width, k;
    int32_t n_slots;           /* concurrent games on this GPU */
    int32_t max_playout;       /* ext_config['player']['max_playout'] */
    int32_t transform_types;   /* 0/1 = off, 8 = dihedral (square boards) */
    int32_t dirichlet_before;  /* noise while turns <= this */
    int32_t max_turn;          /* game stops when turns > this */
    double d_alpha, d_epsilon;
    int32_t tree_edge_cap;     /* most edges one tree may hold; 0 = 3*(max_playout+1)*O.  The edges themselves come in pages from
                                  one pool that all trees of the engine share (tree_pool_edges below) */
    int32_t tree_block_cap;    /* expanded nodes per tree; 0 = 8*(max_playout+1) */
    int32_t max_depth;         /* playout path bound; 0 = 1024 */
    int32_t rng_replay;        /* 1: draws come from az_rng_inject streams; 0: device generator */
    uint64_t seed;
    int32_t record_trajectories; /* keep per-ply moves + visit counts for az_trajectory_pop */
    /* tower (AlphaZero/network/model.py:39-88) */
    int32_t num_blocks;        /* residual blocks; 0 = no tower (priors must be injected) */
    int32_t channels;          /* 256 in the reference (model.py:47,55-56) */
    int32_t tower_dtype;       /* AZ_TOWER_FP32 / AZ_TOWER_BF16 / AZ_TOWER_F16: storage type of the tower's weights and activations
                                  (accumulation is float32 in the 16-bit forms, float64 in the float32 form) */
    int32_t device;            /* HIP device ordinal */
    int32_t leaves_per_round;  /* 0/1: one playout in flight per tree, as the reference (search/mcts.py:295-296) - the only
                                  setting held to the reference's visit counts.  K > 1: K leaves of each tree per round
                                  (single-game / play use: K times the evaluator batch); per-leaf arrays then have
                                  K * n_slots rows, row j * n_slots + slot for the j-th leaf of a slot */
    double virtual_loss;       /* K > 1: subtracted from W of every edge on a path in flight, given back when its value arrives */
    int32_t tree_full_policy;  /* what happens when a tree reaches tree_edge_cap / tree_block_cap or the shared edge pool has no
                                  page left.  0: the run stops with AZ_ERR_TREE_FULL (the reference has no limit, so
                                  nothing else is exact); 1: the node stays a leaf for that playout, its value is still
                                  backed up, and az_counters.expansions_skipped counts the event (production runs);
                                  2: the limits of all trees and the pool are doubled and the stopped playout is completed
                                  when the call that ran the round returns - exact like 0, without the limit (single games) */
    int32_t search_f64;        /* 0: N/W/P arithmetic in float32 (a float32 evaluator: the network, nn_eval_parallel.py:75-92);
                                  1: in float64 (an evaluator returning Python floats / float64: evaluator/dummy_eval.py:26-28).
                                  With 1 the tower cannot feed the search; use the *_f64 entry points */
    int32_t enforce_superko;   /* Go: GameState(enforce_superko=True), positional superko (env/go.py:303-330) in the rules entry
                                  points and inside the search, as the reference's GTP games run (interface/gtp_wrapper.py:88);
                                  self-play runs without it (game/gameplay.py:22).  A slot then keeps its move list and the
                                  hash of every position, up to max(max_turn + 2, 1024) + max_depth moves */
    double c_punt;             /* exploration constant of the selection score, the module global c_punt of search/mcts.py:9 used at
                                  :88; 0 = the reference's 5.0.  az_search_set_c_punt changes it between searches */
    /* Switches that only serve A/B timing and tests; 0 everywhere is the production setting. */
    int32_t tower_no_graphs;   /* 1: small-batch forwards are launched kernel by kernel, never replayed as captured HIP graphs */
    int32_t tower_no_duo;      /* 1: the one-workgroup slab convolution also where the two-workgroups-per-CU kernel applies */
    int32_t tower_group_boards;/* boards per group when a large batch goes through the tower in groups: 0 = automatic,
                                  -1 = never (whole batch layer by layer), N = groups of N boards */
    int32_t match_parallel_towers; /* 0: the two towers of a two-network match run one after the other on the engine's stream.
                                  1: the second network's tower runs beside the first on a stream of another priority (a hardware
                                  queue of its own; +60 % match throughput when neither share fills the chip).  Off by default:
                                  with many streams created earlier in the process, rounds of two concurrently running towers
                                  have been seen to stop making progress on this runtime */
    /* The pool of tree edges.  The reference's tree is a heap of Python objects without a limit (search/mcts.py:30-52); here
     * a tree numbers its edges 0 .. n-1 and a page table maps each page of that range to a page of one pool per engine, so
     * memory follows what the trees hold at the moment, not n_trees times the largest tree. */
    int64_t tree_pool_edges;   /* edges in the pool; 0 = automatic: n_trees * tree_edge_cap (every tree can reach its cap at once)
                                  when tree_edge_cap is given, half of that when tree_edge_cap is 0 as well - the default cap is
                                  generous: the fullest tree of whole 400-ply Go 19x19 games reached 33-56 % of it.  A pool that
                                  runs out is handled like a full tree (tree_full_policy).  Never fewer than one page per tree:
                                  a tree's first page is its own, so a root's children always have a place */
    int32_t tree_page_edges;   /* edges per page, a power of two not below the number of actions; 0 = 4 * (actions rounded up
                                  to a power of two), at least 64 */
    int32_t reserved0;         /* 0 */
} az_config;

/* Fixed-layout copy of one game state (attributes of the GameState classes of env/go.py, env/reversi.py, env/mnk.py). */
typedef struct az_state_blob {
    int8_t board[AZ_CELL_PAD];
    int8_t history[7][AZ_CELL_PAD];   /* board_history, oldest first */
    int32_t current_player, ko, turns, n_history, last1, last2;
    int32_t is_end_of_game, winner, winner_set;
    int32_t passes_black, passes_white;
    int32_t num_black_prisoners, num_white_prisoners, stones_played;
} az_state_blob;

typedef struct az_counters {
    uint64_t steps;            /* select/evaluate/backup rounds */
    uint64_t evals;            /* leaf evaluations consumed (= MCTS sims incl. root expansions) */
    uint64_t moves;            /* plies played by the self-play machine */
    uint64_t games_finished;
    uint64_t levels;           /* tree levels descended */
    uint64_t children_scanned; /* edges read by select */
    uint64_t edges_created;    /* edges written by expand */
    uint64_t rule_moves;       /* do_move replays inside playouts */
    double tower_ms;           /* HIP-event time of all tower forwards since reset */
    double select_ms, finish_ms;
    uint64_t tower_calls;
    uint64_t tower_evals;      /* rows pushed through the tower */
    double conv_ms;            /* HIP-event time of the residual-tower 3x3 conv launches (az_tower_profile on) */
    uint64_t conv_launches;
    double conv_flops_per_launch; /* mean over the timed launches of 2 * boards in the launch * H*W * 9 * C^2 */
    uint64_t expansions_skipped;  /* tree_full_policy 1: expansions dropped because a tree was at its limit or the pool empty */
    uint64_t peak_tree_edges;     /* most edges / expanded nodes any one tree has held since the last reset ... */
    uint64_t peak_tree_blocks;
    uint64_t tree_edge_cap;       /* ... against the per-tree limits in force (az_config.tree_edge_cap / tree_block_cap) */
    uint64_t tree_block_cap;
    uint64_t tree_page_edges;     /* the shared pool of tree edges: edges per page, */
    uint64_t tree_pool_pages;     /* pages in the pool, */
    uint64_t tree_pool_pages_in_use;  /* pages the trees hold now (never fewer than one each), */
    uint64_t tree_pool_pages_peak;    /* and the most they have held at once since the last reset */
} az_counters;

/* ---- lifetime ------------------------------------------------------- */
/* Replaces constructing GameState/Player/Game objects one process per game
 * (AlphaZero/train/parallel/selfplay.py:71-108). */
az_engine *az_engine_create(const az_config *cfg);
/* sizeof of the three structs of this header as the library was compiled: out[0] = az_config, out[1] = az_state_blob,
 * out[2] = az_counters.  A binding in another language checks its own struct definitions against these before the first
 * az_engine_create (no handle needed).  No counterpart in the reference, which has no FFI. */
void az_abi_sizes(int32_t out[3]);
void az_engine_destroy(az_engine *h);
const char *az_last_error(az_engine *h);       /* h may be NULL (creation errors) */
int az_engine_dims(az_engine *h, int32_t *n_slots, int32_t *cells, int32_t *actions);

/* ---- game state (env/go.py, env/reversi.py, env/mnk.py GameState) ----- */
/* GameState(): initial position, both trees cleared, search idle */
int az_games_reset(az_engine *h, const int32_t *slots, int32_t n);
/* GameState.copy(): ref_semantics=1 applies the reference's copy() quirks
 * (Go drops pass counters and the end flag, env/go.py:242-272) */
int az_state_copy(az_engine *dst, int32_t dst_slot, az_engine *src, int32_t src_slot, int32_t ref_semantics);
int az_state_get(az_engine *h, int32_t slot, az_state_blob *out);
int az_state_set(az_engine *h, int32_t slot, const az_state_blob *in);
/* GameState.is_legal for every action, tensor-index order, PASS last */
int az_rules_legal_mask(az_engine *h, const int32_t *slots, int32_t n, uint8_t *mask /* n*O */);
/* GameState.do_move(action, color): colors may be NULL (current player);
 * status[i] = AZ_OK or AZ_ERR_ILLEGAL_MOVE (state untouched); ended[i] = return value */
int az_rules_do_move(az_engine *h, const int32_t *slots, int32_t n, const int32_t *actions,
                     const int32_t *colors, int32_t *status, int32_t *ended);
/* GameState.get_winner(): +1 / -1 / 0, 2 = None (mnk before the end) */
int az_rules_winner(az_engine *h, const int32_t *slots, int32_t n, int32_t *winner);
/* GameState.transform(id) on board + history (env/go.py:573-594) */
int az_state_transform(az_engine *h, int32_t slot, int32_t transform_id);

/* ---- features (processing/state_converter.py:48-82) ------------------- */
/* StateTensorConverter.state_to_tensor on copy().transform(id): int8 (n,17,H,W) */
int az_features(az_engine *h, const int32_t *slots, int32_t n, const int32_t *transform_ids, int8_t *planes);

/* ---- search (search/mcts.py MCTSearch; player/mcts_player.py Player) --- */
/* MCTSearch._calc_move(state, dirichlet) on tree `side` of each slot: arms the search;
 * it then advances one leaf per az_search_step_* round. */
int az_search_begin(az_engine *h, const int32_t *slots, int32_t n, const int32_t *sides, const int32_t *dirichlet);
/* One descent per searching tree (MCTSearch._playout down to the leaf, mcts.py:173-217);
 * n_leaves = trees that need an evaluation this round (0 = all searches finished). */
int az_search_step_select(az_engine *h, int32_t *n_leaves);
/* The evaluator's view of this round's leaves (NNEvaluator.eval input,
 * evaluator/nn_eval_parallel.py:85): planes int8 (n_slots,17,H,W), valid[n_slots] */
int az_leaf_batch(az_engine *h, int8_t *planes, uint8_t *valid);
/* The evaluator's answers as the last finish step used them (what NNEvaluator.eval returned for each leaf,
 * evaluator/nn_eval_parallel.py:88-92): priors (leaves_per_round*n_slots, O) and values (leaves_per_round*n_slots)
 * float32 host arrays in the evaluator's (transformed) index order.  Rows of leaves that took no evaluation (idle
 * slots; with leaf compaction on, finished games below the root) keep whatever an earlier round left there.
 * Either pointer may be NULL.  Float32 engines only. */
int az_leaf_outputs(az_engine *h, float *priors, float *values);
/* expand / terminal handling / backup (mcts.py:217-232,196-202,283-291).
 * priors (n_slots,O) and values (n_slots) float32 host arrays, in the evaluator's
 * (transformed) index order; both NULL = run the tower on the leaf batch. */
int az_search_step_finish(az_engine *h, const float *priors, const float *values);
/* n_rounds of (select, tower, finish) queued on the engine's stream with one synchronisation at the end: the reference's
 * `for _ in range(max_playout): self._playout(...)` (search/mcts.py:295-296) without a host round trip per playout.
 * Needs tower weights; slots whose search is complete ignore the extra rounds. */
int az_search_run(az_engine *h, int32_t n_rounds);
/* root children in insertion order (MCTreeNode._children of _root): returns count */
int az_search_root_stats(az_engine *h, int32_t slot, int32_t side, int32_t *actions, int32_t *visits,
                         float *total_value, float *prior, int32_t *root_visits);
/* math_helper.weighted_random_choice over _get_search_probs with uniform draw u */
int az_search_choose(az_engine *h, int32_t slot, int32_t side, double u, int32_t *move);
/* MCTSearch.update_with_move for one tree / Player.ack for both trees of a slot */
int az_search_advance_tree(az_engine *h, int32_t slot, int32_t side, int32_t move);
int az_search_advance(az_engine *h, const int32_t *slots, int32_t n, const int32_t *moves);
/* Raise the limits of every tree to edge_cap edges / block_cap expanded nodes (0 = twice the present value; never
 * shrinks) and enlarge the shared edge pool by the same factor as edge_cap; the trees keep their contents.  What
 * tree_full_policy 2 does by itself. */
int az_search_grow_trees(az_engine *h, int32_t edge_cap, int32_t block_cap);
/* Enlarge only the shared edge pool, to pool_edges edges (0 = twice the present size; never shrinks, and never beyond what all
 * trees at their limit would need); limits and trees stay as they are.  A production driver calls it between rounds when
 * az_counters.tree_pool_pages_peak comes close to tree_pool_pages, so that no expansion is ever skipped for want of a page
 * (tree_full_policy 1) without sizing the pool for the worst case up front.  No counterpart in the reference, whose tree is an
 * unbounded heap (search/mcts.py:30-52). */
int az_search_grow_pool(az_engine *h, int64_t pool_edges);
/* nodes (edges) and expanded nodes currently held by a tree */
int az_search_tree_size(az_engine *h, int32_t slot, int32_t side, int32_t *edges, int32_t *blocks);
/* Assigning the module global c_punt of search/mcts.py:9: the value every later selection step uses (c_punt > 0). */
int az_search_set_c_punt(az_engine *h, double c_punt);

/* ---- self-play machine (game/gameplay.py Game.start over all slots) ---- */
int az_selfplay_start(az_engine *h);            /* reset every slot and arm player_1's first search */
/* n rounds of select -> tower -> finish -> (move, re-root, next search) */
int az_selfplay_run(az_engine *h, int32_t n_rounds);
/* injected-evaluator variant of one round: select, then the caller evaluates
 * az_leaf_batch, then finish(priors, values) which also plays due moves */
int az_selfplay_select(az_engine *h, int32_t *n_leaves);
int az_selfplay_finish(az_engine *h, const float *priors, const float *values);
/* per-slot progress: plies played, 1 if the game is over and waits for az_trajectory_pop */
int az_selfplay_status(az_engine *h, int32_t *plies /* n_slots */, int32_t *done /* n_slots */);
/* Game.get_history raw material for a finished slot: moves[n_plies], visit counts
 * (n_plies,O) scattered by action, winner.  restart=1 re-arms the slot with a new game.
 * Returns n_plies, or AZ_ERR_STATE if the slot's game is not finished. */
int az_trajectory_pop(az_engine *h, int32_t slot, int16_t *moves, int32_t *visits, int32_t *winner, int32_t restart);
int az_trajectory_max_plies(az_engine *h);
/* The same for n finished slots in one call (one synchronisation, one packed copy): the games' rows lie back to back,
 * game i at rows row_start[i] .. row_start[i] + plies_i of moves[total_rows] and visits[total_rows][O]; row_start must be
 * the running sum of the plies az_selfplay_status reported and total_rows their sum (checked). winners[n]. */
int az_trajectory_pop_many(az_engine *h, const int32_t *slots, int32_t n, const int32_t *row_start, int32_t total_rows,
                           int16_t *moves, int32_t *visits, int32_t *winners, int32_t restart);
/* The same with the packed rows left in this GPU's memory: moves_dev[total_rows] and visits_dev[total_rows][O] are device
 * pointers of the caller (e.g. torch tensors), so finished games can be compacted and sent on to the replay buffer's
 * rank without a pass through host memory; winners[n] is a host array.  Returns after the engine's stream is idle. */
int az_trajectory_pop_many_dev(az_engine *h, const int32_t *slots, int32_t n, const int32_t *row_start, int32_t total_rows,
                               int16_t *moves_dev, int32_t *visits_dev, int32_t *winners, int32_t restart);
/* az_trajectory_pop_many for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc); it has no counterpart in the reference.  Same arguments and results, but the device-side
 * staging of the packed rows is a fixed array inside the library (one per GPU, AZ_STATIC_POP_BYTES, held until the
 * process ends) instead of a buffer grown on demand, and the call makes no device or host allocation.  Limits:
 * n <= AZ_STATIC_POP_SLOTS and total_rows * (2 + 4 * O) <= AZ_STATIC_POP_BYTES - 16384, else AZ_ERR_ARG with a message.
 * Every call on a GPU shares that GPU's array: calls on one GPU that overlap in time (two handles, two threads) race. */
#define AZ_STATIC_POP_SLOTS 256
#define AZ_STATIC_POP_BYTES (8u << 20)
int az_trajectory_pop_many_static(az_engine *h, const int32_t *slots, int32_t n, const int32_t *row_start, int32_t total_rows,
                                  int16_t *moves, int32_t *visits, int32_t *winners, int32_t restart);

/* ---- RNG (numpy.random.randint / random.gammavariate / random.choices) - */
/* replay streams for one slot: transform ids, gamma draws, move-choice uniforms */
int az_rng_inject(az_engine *h, int32_t slot, const int32_t *tids, int64_t n_tids, const double *etas,
                  int64_t n_etas, const double *unis, int64_t n_unis);
int az_rng_cursor(az_engine *h, int32_t slot, int64_t *i_tids, int64_t *i_etas, int64_t *i_unis);

/* ---- tower (network/model.py, network/main.py:111-127 Network.response) - */
size_t az_tower_blob_floats(az_engine *h);      /* float32 count of the folded-weight blob */
/* folded float32 weights, layout documented in alphazero_amd/network/model.py */
int az_tower_load_weights(az_engine *h, const float *blob, size_t n_floats);
/* The same blob already in this GPU's memory (e.g. the tensor an RCCL broadcast delivered): padding, conversion to the
 * compute type and the copies into the tower's weight arrays run on the GPU, nothing passes through host memory.  net = 0 / 1
 * as az_tower_load_weights_net.  The caller's writes to blob_dev must be complete (its stream synchronised); the call
 * returns after the engine's stream has finished reading it.  Replaces saver.restore of a checkpoint path in every worker
 * (train/parallel/selfplay.py:110-122, evaluator/nn_eval_parallel.py:100-106). */
int az_tower_load_weights_dev(az_engine *h, int32_t net, const float *blob_dev, size_t n_floats);
/* float64 forms of the three calls that carry W / P / priors / values, for engines created with search_f64 = 1
 * (the float32 forms return AZ_ERR_ARG on such an engine, and these on a float32 engine) */
int az_search_step_finish_f64(az_engine *h, const double *priors, const double *values);
int az_selfplay_finish_f64(az_engine *h, const double *priors, const double *values);
int az_search_root_stats_f64(az_engine *h, int32_t slot, int32_t side, int32_t *actions, int32_t *visits,
                             double *total_value, double *prior, int32_t *root_visits);

/* Gating matches between two weight sets (train/parallel/evaluator.py:70-98: Game(nn_eval_chal, nn_eval_best, ...) or
 * the two swapped).  Network 0 is the one az_tower_load_weights fills; network 1 is created on first load.
 * az_match_assign: in slot slots[i] the side that moves first (player_1, Black) is evaluated by network
 * net_of_first[i] (0/1), the other side by the other network; from then on az_selfplay_run / az_selfplay_finish
 * with NULL priors route every leaf to the network owning the searching tree.  n = 0 returns to one network. */
int az_tower_load_weights_net(az_engine *h, int32_t net, const float *blob, size_t n_floats);
int az_match_assign(az_engine *h, const int32_t *slots, int32_t n, const int32_t *net_of_first);
/* Which network evaluates this round's leaf of every slot, for a caller that computes the priors itself
 * (az_selfplay_select -> az_leaf_batch -> here -> az_selfplay_finish(priors, values)): net_of_slot[n_slots] = 0 / 1, or -1
 * for a slot whose leaf needs no evaluation (idle, or a finished game below the root).  Runs the same device-side split
 * as the tower path and returns its lists, i.e. nn_eval_1 / nn_eval_2 of Game(nn_eval_1, nn_eval_2, ...) by side to move
 * (game/gameplay.py:19-20, 62-65). */
int az_match_leaf_nets(az_engine *h, int32_t *net_of_slot);
/* x int8 (B,17,H,W) host; out_p (B,O), out_v (B) float32 host */
int az_tower_forward(az_engine *h, const int8_t *planes, int32_t B, float *out_p, float *out_v);
/* the same with all three arrays already in this GPU's memory (e.g. torch.Tensor.data_ptr()); returns after the
 * engine's stream has finished, so the outputs can be read from any stream */
int az_tower_forward_dev(az_engine *h, const int8_t *planes_dev, int32_t B, float *out_p_dev, float *out_v_dev);
double az_tower_flops_per_eval(az_engine *h);
/* Name (as a profiler's kernel trace shows it) and a short description of the convolution kernel that the residual layers of a
 * self-play round run with on this engine - it depends on the tower's type and width, the board and the slot count.
 * Measurement only; the string lives as long as the library. */
const char *az_tower_conv_kernel(az_engine *h);
/* Per-launch timing of the residual tower's convolutions inside az_selfplay_run: HIP events on the launch stream
 * around every such launch, accumulated in az_counters.conv_ms / conv_launches (measurement only).
 * on=0: off.  on=1: every round (the host then waits for each round, and every round runs the full batch).
 * on=N>=8: every N-th round only; that round runs the full batch (no leaf compaction, so the work per launch is
 * the nominal one), all other rounds run as in production and stay queued ahead of the host. */
int az_tower_profile(az_engine *h, int32_t on);
/* on=1 (the default): a round's tower pass covers only the leaf rows that need it. The reference calls its evaluator for
 * every playout and throws the answer away at a finished game (search/mcts.py:204-211 against :229-231); results are
 * the same either way, on=0 only serves comparisons. Applies where the tower's kernels take the row count from device
 * memory (256-channel bf16 tower, large batches) and never while az_tower_profile is on. */
int az_tower_leaf_compaction(az_engine *h, int32_t on);

/* ---- measurement ------------------------------------------------------ */
int az_counters_get(az_engine *h, az_counters *out);
int az_counters_reset(az_engine *h);
int az_sync(az_engine *h);

#ifdef __cplusplus
}
#endif
#endif

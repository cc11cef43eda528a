/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (plain C) of the reference's self-play hot path:
 * game rules (Go / Reversi / m,n,k), feature planes, dihedral transforms,
 * sequential PUCT search and the two-player game loop.  It exists only as
 * the checker for the HIP engine and as the CPU baseline that bench.py
 * times; nothing in the product path may link or call it.
 *
 * Parity status: pinned.  tests/test_oracle_golden.py checks every function
 * here against fixtures in tests/golden/ that were produced by running the
 * reference's own Python (oracle/gen_golden.py) and against the
 * known-answer cases of the reference's unit tests.
 *
 * Coordinates follow the reference: board[x][y], flat cell index x*W + y,
 * PASS is action index H*W (reference: AlphaZero/game/gameplay.py:89-92).
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H
#include <stdint.h>

/* Arithmetic type of the search statistics W, P and of the evaluator's outputs.  float reproduces the
 * reference with a float32 evaluator (the neural network); building with -DOZ_REAL=double reproduces it
 * with an evaluator that returns Python floats / float64 (evaluator/dummy_eval.py:26-28), where NumPy and
 * Python keep every intermediate in double. */
#ifndef OZ_REAL
#define OZ_REAL float
#endif
typedef OZ_REAL oz_real;

#ifdef __cplusplus
extern "C" {
#endif

#define OZ_MAXC 361
#define OZ_MAXO 362
#define OZ_NHIST 7
#define OZ_NPLANES 17
#define OZ_SK_MAX 1024

enum { OZ_GO = 0, OZ_REVERSI = 1, OZ_MNK = 2 };
#define OZ_BLACK 1
#define OZ_WHITE (-1)
#define OZ_NOMOVE (-2)

typedef struct {
    int game, H, W, k, O;
    int8_t board[OZ_MAXC];
    int8_t hist[OZ_NHIST][OZ_MAXC]; /* board_history, oldest first */
    int cur;                        /* player to move */
    int ko;                         /* cell index or -1 */
    int turns;
    int nhist;                      /* len(history) */
    int last1, last2;               /* history[-1], history[-2] */
    int is_end;
    int winner;                     /* mnk: stored winner; valid if winner_set */
    int winner_set;
    int passes_b, passes_w;
    int pris_b, pris_w;             /* num_black_prisoners, num_white_prisoners */
    int stones_played;              /* reversi */
    double komi;
    /* Positional superko (go.py:303-330; GTP play only: interface/gtp_wrapper.py:88).  Everything from here on is copied
     * by oz_copy only as far as it is filled, and not at all when superko is 0. */
    int superko;                    /* enforce_superko */
    int nprev;                      /* len(previous_hashes as a list): one entry per stone move */
    uint64_t cur_hash;              /* current_hash */
    int16_t sk_moves[OZ_SK_MAX];    /* history (all moves, PASS included) */
    uint64_t sk_hash[OZ_SK_MAX];    /* position hashes after every stone move so far */
} OzState;

/* ---- rules ---- */
void oz_init(OzState *s, int game, int H, int W, int k);
void oz_set_superko(OzState *s, int on);               /* GameState(enforce_superko=...): call on a fresh state */
void oz_copy(OzState *dst, const OzState *src);        /* GameState.copy() semantics */
int oz_is_legal(const OzState *s, int action);
void oz_legal_mask(const OzState *s, uint8_t *mask /*O*/);
int oz_do_move(OzState *s, int action);                /* 0 ok, -1 IllegalMove */
int oz_do_move_color(OzState *s, int action, int color); /* color 0 = current player */
int oz_winner(const OzState *s);                       /* +1/-1/0; mnk: 2 if None */

/* ---- features / transforms ---- */
void oz_transform_board(const int8_t *src, int8_t *dst, int H, int W, int tid);
void oz_features(const OzState *s, int tid, int8_t *planes /*17*H*W*/);
/* label of the i-th entry of the action list after reverse_transform */
int oz_reverse_label(int idx, int H, int W, int tid);

/* ---- search ---- */
typedef void (*oz_eval_fn)(void *ctx, const int8_t *planes, int H, int W, int O,
                           oz_real *priors, oz_real *value);

typedef struct {
    const int32_t *tids; int64_t n_tids, i_tids;       /* transform ids */
    const double *etas; int64_t n_etas, i_etas;        /* gamma draws */
    const double *unis; int64_t n_unis, i_unis;        /* uniforms for move choice */
    uint64_t seed;                                     /* used when a stream is NULL */
    int underflow;
} OzRng;

typedef struct OzTree OzTree;
OzTree *oz_tree_new(int O, int transform_types, double d_alpha, double d_epsilon);
void oz_set_c_punt(double c_punt);   /* the module global c_punt, mcts.py:9 (default 5.0) */
void oz_tree_free(OzTree *t);
/* MCTSearch._calc_move */
int oz_tree_search(OzTree *t, const OzState *s, int max_playout, int dirichlet,
                   oz_eval_fn ev, void *ctx, OzRng *rng);
/* root children in insertion order */
int oz_tree_root_stats(const OzTree *t, int32_t *actions, int32_t *visits,
                       oz_real *W, oz_real *P);
int oz_tree_root_visits(const OzTree *t);
int oz_tree_choose(const OzTree *t, double u);          /* weighted_random_choice */
int oz_tree_argmax(const OzTree *t);                    /* calc_move(prop_exp=False) */
void oz_tree_update_with_move(OzTree *t, int action);
int64_t oz_tree_num_nodes(const OzTree *t);
/* single playout from root (used by the reference's own unit tests) */
void oz_tree_playout(OzTree *t, const OzState *s, oz_eval_fn ev, void *ctx, OzRng *rng);
int oz_tree_count_expanded(const OzTree *t);

/* ---- one self-play game (game/gameplay.py Game.start) ---- */
typedef struct {
    int game, H, W, k;
    int transform_types;
    int max_playout, dirichlet_before, max_turn;
    double d_alpha, d_epsilon;
} OzGameCfg;

typedef struct {
    int n_plies;
    int winner;
    int16_t *moves;      /* [max_plies] */
    int32_t *visits;     /* [max_plies][O] visit count scattered by action index */
    int64_t n_evals;
} OzTrajectory;

int oz_play_game(const OzGameCfg *cfg, oz_eval_fn ev, void *ctx, OzRng *rng,
                 OzTrajectory *traj, int max_plies);
/* two evaluators (player_1 = ev1 moves first) and, with K > 1, the several-leaves-per-round schedule */
int oz_play_game_ex(const OzGameCfg *cfg, oz_eval_fn ev1, void *ctx1, oz_eval_fn ev2, void *ctx2, int K, double vloss,
                    OzRng *rng, OzTrajectory *traj, int max_plies);
/* K leaves per round under a virtual loss: the engine's opt-in schedule restated (not in the reference) */
#define OZ_KMAXD 1024
int oz_tree_search_k(OzTree *t, const OzState *s, int max_playout, int K, double vloss, int dirichlet,
                     oz_eval_fn ev, void *ctx, OzRng *rng);

/* deterministic hash evaluator shared with the fixture generator */
void oz_fake_eval(void *ctx, const int8_t *planes, int H, int W, int O,
                  oz_real *priors, oz_real *value);
/* uniform priors 1/O (float32), value 0 */
void oz_uniform_eval(void *ctx, const int8_t *planes, int H, int W, int O,
                     oz_real *priors, oz_real *value);

uint64_t oz_splitmix64(uint64_t *x);
double oz_gamma_small_alpha(double alpha, uint64_t *st);

#ifdef __cplusplus
}
#endif
#endif

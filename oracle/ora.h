/*
 * oracle/ora.h -- CPU (fp64) restatement of the ddrl learning hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * The arithmetic of the MLP path lives in a forked Caffe (github.com/matthieu637/caffe, unpinned,
 * absent from /root/reference and from this image); InnerProduct / ReLU / TanH / EuclideanLoss /
 * AdamSolver follow the published BVLC Caffe 1.0 algorithms.  Pinned by the reference's own
 * known-answer tests (common/src/test/CaffeUtest.cpp:24-55, 372-386) -- see tests/test_oracle_kat.py.
 * DDPG / NFAC / EWC update arithmetic is not covered by any reference test: "parity unpinned"
 * beyond those KATs.
 */
#ifndef ORA_H
#define ORA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAXL 8

/* ---- std::mt19937 + libstdc++-11 uniform_int_distribution (rng_ref.c) ---- */
typedef struct { uint32_t mt[624]; int idx; } ora_mt19937;
void     ora_mt_seed(ora_mt19937 *g, uint32_t seed);
uint32_t ora_mt_next(ora_mt19937 *g);
/* std::uniform_int_distribution<int>(0, n-1)(engine), GCC>=11 (Lemire) algorithm */
int      ora_uniform_index(ora_mt19937 *g, uint32_t n);

/* ---- MLP (mlp_ref.c) ---- */
typedef struct ora_mlp ora_mlp;
/* dims[0..nlayers]: input, hidden..., output.  hidden_act: 1 leaky(0.01) 2 tanh 3 relu.
 * out_act: 0 linear 1 leaky 2 tanh 3 relu.  momentum: 0..3 (MLP.hpp:160-173). decay<0 => off */
ora_mlp *ora_mlp_create(int nlayers, const int *dims, int hidden_act, int out_act,
                        double lr, int momentum, double decay);
/* the same with BatchNorm / Scale towers: bn_code = the batch_norm bit-field (MLP.hpp:53-79); critic_sensors = size_sensors
 * for a critic (its first_except_action variant normalises the states only, and it has no output tower), 0 for an actor.
 * The flat parameter vector is in the net's layer order: per tower [mean_sum C, var_sum C, factor][gamma C (, beta C)]
 * before the InnerProduct it feeds (learnable_params(), MLP.hpp:610-676).  Pinned by the consistency properties of
 * CaffeUtest.cpp:246-386, 1343-1447 (tests/test_oracle_bn.py); BatchNorm/Scale arithmetic = published BVLC Caffe 1.0. */
ora_mlp *ora_mlp_create_bn(int nlayers, const int *dims, int hidden_act, int out_act, double lr, int momentum, double decay,
                           int bn_code, int critic_sensors);
void   ora_mlp_set_phase(ora_mlp *n, int test);       /* copy with ::caffe::Phase::TEST: use_global_stats (MLP.hpp:284-338) */
int    ora_mlp_bn_layout(const ora_mlp *n, int *rank_on, int *offbn, int *offsc, int *chan);   /* arrays of nlayers + 1; returns scale | bias << 1 */
void   ora_mlp_layer_offsets(const ora_mlp *n, int *offw, int *offb);
void   ora_mlp_free(ora_mlp *n);
ora_mlp *ora_mlp_clone(const ora_mlp *n);
int    ora_mlp_nparams(const ora_mlp *n);
void   ora_mlp_set_weights(ora_mlp *n, const double *w);
void   ora_mlp_get_weights(const ora_mlp *n, double *w);
void   ora_mlp_get_grads(const ora_mlp *n, double *g);
void   ora_mlp_get_adam(const ora_mlp *n, double *m, double *v);
void   ora_mlp_set_adam(ora_mlp *n, const double *m, const double *v, int iter);
int    ora_mlp_iter(const ora_mlp *n);
void   ora_mlp_set_iter(ora_mlp *n, int it);
void   ora_mlp_set_lr(ora_mlp *n, double lr);
/* input gate mask (developmental state gate) and output gate mask; NULL => all ones */
void   ora_mlp_set_gates(ora_mlp *n, const double *in_gate, const double *out_gate);
/* per-row gate factors (Bernoulli developmental gates, CaffeUtest.cpp:1048-1158): queue a set [B x d0] / [B x dL]
 * (NULL = ones); each forward of batch B consumes the oldest set, which stays in force for that pass's backward and
 * replaces the per-dimension gates.  Returns 0, or -1 when the queue (16) is full. */
int    ora_mlp_push_row_gates(ora_mlp *n, const double *in_rows, const double *out_rows, int B);
int    ora_mlp_row_gates_pending(const ora_mlp *n);
void   ora_mlp_forward(ora_mlp *n, const double *x, int B, double *out);
/* backward from a diff on the post-activation output; accumulates param diffs; din may be NULL */
void   ora_mlp_backward(ora_mlp *n, const double *dout, int B, double *din);
void   ora_mlp_zero_grad(ora_mlp *n);
void   ora_mlp_apply_update(ora_mlp *n);            /* Solver::ApplyUpdate (no ++iter) */
double ora_mlp_learn_batch(ora_mlp *n, const double *x, const double *y, int B, int iters);
double ora_mlp_learn_batch_lw(ora_mlp *n, const double *x, const double *y, const double *lw, int B, int iters);
void   ora_mlp_soft_update(ora_mlp *to, const ora_mlp *from, double tau);
/* EWC (DODevMLP.hpp:604-758) */
void   ora_mlp_ewc_config(ora_mlp *n, double ewc, double ewc_decay, double fisher_beta, int best_method);
void   ora_mlp_update_fisher(ora_mlp *n, double nsample);
void   ora_mlp_regularize(ora_mlp *n);
double ora_mlp_ewc_cost(const ora_mlp *n);
void   ora_mlp_ewc_setup(ora_mlp *n);
void   ora_mlp_ewc_decay_update(ora_mlp *n);
void   ora_mlp_update_best_param(ora_mlp *n, double score);
void   ora_mlp_get_ewc(const ora_mlp *n, double *fisher, double *fisher_hat, double *wstar,
                       double *fisher_nbr, double *decay_mult);
int    ora_mlp_ewc_active(const ora_mlp *n);
void   ora_mlp_set_ewc(ora_mlp *n, const double *fisher_hat, const double *wstar);

/* ---- DDPG (ddpg_ref.c) ---- */
typedef struct ora_ddpg ora_ddpg;
typedef struct {
  int S, A, B, replay_cap;
  int n_hidden_a, hidden_a[ORA_MAXL];
  int n_hidden_q, hidden_q[ORA_MAXL];
  int hidden_act, actor_out_act, momentum;
  int inverting_grad, nstep_return, force_more_update;
  double gamma, tau, alpha_a, alpha_v, decay_v;
  double ewc, ewc_decay, ewc_fisher_beta;   /* ewc<0 => disabled */
  int ewc_for_critic, ewc_best_method;
  uint32_t seed;                            /* replay sampling engine seed */
  int td3;                                  /* TD3Ag: second critic pair (qnn2, qnn_target2) */
  int bn_actor, bn_critic;                  /* batch_norm_actor / batch_norm_critic (DeepQNAg.hpp:180-181) */
  int test_net, bn_adapt;                   /* targets as TEST-phase copies (:216-219, 231-234); :415-420 */
} ora_ddpg_cfg;
ora_ddpg *ora_ddpg_create(const ora_ddpg_cfg *c);
void   ora_ddpg_free(ora_ddpg *d);
ora_mlp *ora_ddpg_net(ora_ddpg *d, int which);   /* 0 ann 1 qnn 2 ann_target 3 qnn_target 4 qnn2 5 qnn_target2 (td3) 6 ann_testing */
/* DeepQNAg.hpp:469-472: ann_testing <- every parameter of ann (after the fupd loop) */
void   ora_ddpg_sync_testing(ora_ddpg *d);
void   ora_ddpg_insert(ora_ddpg *d, const double *s, const double *a, double r,
                       const double *s2, int goal);          /* ring push, no update */
int    ora_ddpg_replay_len(const ora_ddpg *d);
/* one pass of the fupd loop body (DeepQNAg.hpp:343-466); idx==NULL => draw from the engine.
 * idx_out (B ints, may be NULL) receives the indices used.  Returns the critic loss. */
double ora_ddpg_update(ora_ddpg *d, const int *idx, int *idx_out);
void   ora_ddpg_ewc_setup(ora_ddpg *d);
/* diagnostics of the last update */
void   ora_ddpg_last(const ora_ddpg *d, double *q_targets, double *actor_dact);

/* TD3 (td3_ref.c): TD3Ag::end_episode (agent/deepqn/include/TD3Ag.hpp:284-380) on a ddpg object created
 * with td3 = 1.  noise [B x A]: the target-policy smoothing draws (:307; the reference takes them from its
 * clock-seeded engine, so the caller supplies them).  actor_step: current_step_counter % policy_freq == 0
 * (:333).  Returns the first critic's loss; *loss2 (may be NULL) the second's. */
double ora_td3_update(ora_ddpg *d, const int *idx, int *idx_out, const double *noise, int actor_step, double *loss2);

/* ---- NFAC(lambda)-V (nfac_ref.c) ---- */
typedef struct {
  int S, A;
  int n_hidden_a, hidden_a[ORA_MAXL];
  int n_hidden_v, hidden_v[ORA_MAXL];
  int hidden_act, actor_out_act, momentum;
  int update_delta_neg, update_critic_first, gae, inverting_gradient;
  int number_fitted_iteration, stoch_iter_actor, stoch_iter_critic;
  double gamma, lambda, alpha_a, alpha_v;
  double ewc, ewc_decay, ewc_fisher_beta;
  int ewc_for_critic, ewc_best_method;
  int vnn_from_scratch;              /* OfflineCaclaAg.hpp:129,281-284 */
  int corrected_update_ac;           /* :164-173, 550-563 */
  double corrected_update_ac_factor;
} ora_nfac_cfg;
typedef struct ora_nfac ora_nfac;
ora_nfac *ora_nfac_create(const ora_nfac_cfg *c);
void   ora_nfac_free(ora_nfac *d);
ora_mlp *ora_nfac_net(ora_nfac *d, int which);   /* 0 ann 1 vnn */
/* vnn_from_scratch: the weights the re-created critics start from, n vectors of the critic's parameter count,
 * consumed one per fitted iteration (the reference draws them from Caffe's generator; the caller supplies them) */
void   ora_nfac_set_fresh(ora_nfac *d, const double *w, int n);
/* the learning branch of OfflineCaclaAg::end_episode (OfflineCaclaAg.hpp:423-594) on a trajectory
 * set of T rows; ends[n_ends] = trajectory_end_points. Returns n (positively advantaged rows). */
int    ora_nfac_update(ora_nfac *d, int T, const double *s, const double *pure_a, const double *a,
                       const double *s2, const double *r, const int *goal,
                       const int *ends, int n_ends, double *deltas_out);

#ifdef __cplusplus
}
#endif
#endif

/*
 * oracle/ddpg_ref.c -- TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * fp64 restatement of the DDPG minibatch update, DeepQNAg::end_episode
 * (agent/deepqn/include/DeepQNAg.hpp:336-473), one call of ora_ddpg_update() = one pass of the
 * `fupd` loop body (:343-466) for one agent.  Replay = std::deque<sample> with pop_front at
 * capacity (DeepQNAg.hpp:159-165); logical index 0 is the oldest sample.
 * No reference test pins this arithmetic (agent/deepqn has no unit tests): parity unpinned beyond
 * the MLP known-answer tests.
 */
#include "ora.h"
#include <stdlib.h>
#include <string.h>

#include "ddpg_ref.h"

ora_ddpg *ora_ddpg_create(const ora_ddpg_cfg *c) {
  ora_ddpg *d = (ora_ddpg *)calloc(1, sizeof(ora_ddpg));
  d->c = *c;
  int dims[ORA_MAXL + 1];
  /* actor: DeepQNAg.hpp:209-210 -> MLP actor ctor (MLP.hpp:193-279) */
  dims[0] = c->S;
  for (int i = 0; i < c->n_hidden_a; ++i) dims[i + 1] = c->hidden_a[i];
  dims[c->n_hidden_a + 1] = c->A;
  d->ann = ora_mlp_create_bn(c->n_hidden_a + 1, dims, c->hidden_act, c->actor_out_act, c->alpha_a,
                             c->momentum, -1., c->bn_actor, 0);
  /* critic: DeepQNAg.hpp:224-225 -> MLP critic ctor (MLP.hpp:86-184): input = concat(s, a), out 1 */
  dims[0] = c->S + c->A;
  for (int i = 0; i < c->n_hidden_q; ++i) dims[i + 1] = c->hidden_q[i];
  dims[c->n_hidden_q + 1] = 1;
  d->qnn = ora_mlp_create_bn(c->n_hidden_q + 1, dims, c->hidden_act, 0, c->alpha_v, c->momentum,
                             c->decay_v, c->bn_critic, c->S);
  if (c->ewc >= 0.) {
    ora_mlp_ewc_config(d->ann, c->ewc, c->ewc_decay, c->ewc_fisher_beta, c->ewc_best_method);
    if (c->ewc_for_critic >= 1)  /* DODevMLP::ewc_enabled (DODevMLP.hpp:714-720) */
      ora_mlp_ewc_config(d->qnn, c->ewc, c->ewc_decay, c->ewc_fisher_beta, c->ewc_best_method);
  }
  d->ann_target = ora_mlp_clone(d->ann);   /* DeepQNAg.hpp:216-219 */
  d->qnn_target = ora_mlp_clone(d->qnn);   /* DeepQNAg.hpp:231-234 */
  if (c->test_net) { ora_mlp_set_phase(d->ann_target, 1); ora_mlp_set_phase(d->qnn_target, 1); }
  d->ann_testing = ora_mlp_clone(d->ann);  /* :221-222 */
  ora_mlp_set_phase(d->ann_testing, 1);
  if (c->td3) {                            /* TD3Ag.hpp:221-229: a second critic of the same shape and its target */
    d->qnn2 = ora_mlp_create_bn(c->n_hidden_q + 1, dims, c->hidden_act, 0, c->alpha_v, c->momentum, c->decay_v,
                                c->bn_critic, c->S);             /* TD3Ag.hpp:221-222: batch_norm_critic like qnn */
    if (c->ewc >= 0. && c->ewc_for_critic >= 1)
      ora_mlp_ewc_config(d->qnn2, c->ewc, c->ewc_decay, c->ewc_fisher_beta, c->ewc_best_method);
    d->qnn_target2 = ora_mlp_clone(d->qnn2);
    if (c->test_net) ora_mlp_set_phase(d->qnn_target2, 1);     /* TD3Ag.hpp:224-229 */
  }
  ora_mt_seed(&d->eng, c->seed);
  d->row = 2 * c->S + 2 * c->A + 2;
  d->rep = (double *)calloc((size_t)c->replay_cap * d->row, sizeof(double));
  d->last_qt = (double *)calloc(c->B, sizeof(double));
  d->last_dact = (double *)calloc((size_t)c->B * c->A, sizeof(double));
  return d;
}

void ora_ddpg_free(ora_ddpg *d) {
  if (!d) return;
  ora_mlp_free(d->ann); ora_mlp_free(d->qnn); ora_mlp_free(d->ann_target); ora_mlp_free(d->qnn_target);
  ora_mlp_free(d->ann_testing);
  if (d->qnn2) { ora_mlp_free(d->qnn2); ora_mlp_free(d->qnn_target2); }
  free(d->rep); free(d->last_qt); free(d->last_dact); free(d);
}

ora_mlp *ora_ddpg_net(ora_ddpg *d, int which) {
  switch (which) {
    case 0: return d->ann; case 1: return d->qnn; case 2: return d->ann_target; case 3: return d->qnn_target;
    case 4: return d->qnn2; case 6: return d->ann_testing; default: return d->qnn_target2;
  }
}

/* target nets were cloned at creation; after set_weights on ann/qnn the test re-syncs explicitly */


/* DeepQNAg::insertSample (DeepQNAg.hpp:159-165) without the trailing end_episode(true) */
void ora_ddpg_insert(ora_ddpg *d, const double *s, const double *a, double r, const double *s2, int goal) {
  int S = d->c.S, A = d->c.A;
  int slot;
  if (d->len >= d->c.replay_cap) { slot = d->head; d->head = (d->head + 1) % d->c.replay_cap; }
  else { slot = (d->head + d->len) % d->c.replay_cap; d->len++; }
  double *p = d->rep + (size_t)slot * d->row;
  memcpy(p, s, sizeof(double) * S);
  memcpy(p + S, a, sizeof(double) * A);       /* pure_a slot: not used by the update */
  memcpy(p + S + A, a, sizeof(double) * A);
  memcpy(p + S + 2 * A, s2, sizeof(double) * S);
  p[2 * S + 2 * A] = r;
  p[2 * S + 2 * A + 1] = goal ? 1. : 0.;
}

int ora_ddpg_replay_len(const ora_ddpg *d) { return d->len; }

double ora_ddpg_update(ora_ddpg *d, const int *idx_in, int *idx_out) {
  const ora_ddpg_cfg *c = &d->c;
  int S = c->S, A = c->A, B = c->B;
  int *idx = (int *)malloc(sizeof(int) * B);
  int *valid_next = (int *)calloc(B, sizeof(int));
  /* sample_transition (DeepQNAg.hpp:275-288) */
  for (int i = 0; i < B; ++i)
    idx[i] = idx_in ? idx_in[i] : ora_uniform_index(&d->eng, (uint32_t)d->len);
  if (idx_out) memcpy(idx_out, idx, sizeof(int) * B);

  double *all_s = (double *)malloc(sizeof(double) * B * S);
  double *all_s2 = (double *)malloc(sizeof(double) * B * S);
  double *all_a = (double *)malloc(sizeof(double) * B * A);
  double *sa = (double *)malloc(sizeof(double) * B * (S + A));
  double *next_a = (double *)malloc(sizeof(double) * B * A);
  double *qt = (double *)malloc(sizeof(double) * B);
  double *pi = (double *)malloc(sizeof(double) * B * A);
  double *dq = (double *)malloc(sizeof(double) * B);
  double *dsa = (double *)malloc(sizeof(double) * B * (S + A));
  double *dact = (double *)malloc(sizeof(double) * B * A);

  /* gather (DeepQNAg.hpp:353-377) */
  for (int i = 0; i < B; ++i) {
    const double *p = rep_row(d, idx[i]);
    if (c->nstep_return <= 0) {
      memcpy(all_s2 + i * S, p + S + 2 * A, sizeof(double) * S);
    } else {
      int index_next_step = idx[i];
      for (int j = idx[i] + 1; j < idx[i] + c->nstep_return && j < d->len; ++j) {
        if (rep_row(d, j)[2 * S + 2 * A + 1] != 0.) break;
        index_next_step = j;
      }
      valid_next[i] = index_next_step - idx[i];
      memcpy(all_s2 + i * S, rep_row(d, index_next_step) + S + 2 * A, sizeof(double) * S);
    }
    memcpy(all_s + i * S, p, sizeof(double) * S);
    memcpy(all_a + i * A, p + S + A, sizeof(double) * A);
  }

  /* a' = ann_target(s'), q' = qnn_target(s', a')  (DeepQNAg.hpp:379-383) */
  ora_mlp_forward(d->ann_target, all_s2, B, next_a);
  for (int i = 0; i < B; ++i) {
    memcpy(sa + i * (S + A), all_s2 + i * S, sizeof(double) * S);
    memcpy(sa + i * (S + A) + S, next_a + i * A, sizeof(double) * A);
  }
  ora_mlp_forward(d->qnn_target, sa, B, qt);

  /* targets (DeepQNAg.hpp:386-405) */
  for (int i = 0; i < B; ++i) {
    const double *p = rep_row(d, idx[i]);
    double r = p[2 * S + 2 * A];
    int goal = p[2 * S + 2 * A + 1] != 0.;
    if (goal) qt[i] = r;
    else if (c->nstep_return <= 0) qt[i] = r + c->gamma * qt[i];
    else {
      double Qnext = qt[i];
      qt[i] = 0;
      double local_gamma = 1.f;
      for (int j = 0; j < valid_next[i]; ++j) {
        qt[i] += rep_row(d, idx[i] + j)[2 * S + 2 * A] * local_gamma;
        local_gamma *= c->gamma;
      }
      qt[i] += c->gamma * Qnext;
    }
  }
  memcpy(d->last_qt, qt, sizeof(double) * B);

  /* critic update: qnn->learn_batch(s, a, y, 1)  (DeepQNAg.hpp:408) */
  for (int i = 0; i < B; ++i) {
    memcpy(sa + i * (S + A), all_s + i * S, sizeof(double) * S);
    memcpy(sa + i * (S + A) + S, all_a + i * A, sizeof(double) * A);
  }
  double loss = ora_mlp_learn_batch(d->qnn, sa, qt, B, 1);

  /* actor update (DeepQNAg.hpp:411-459) */
  ora_mlp_zero_grad(d->qnn);
  ora_mlp_zero_grad(d->ann);
  ora_mlp_forward(d->ann, all_s, B, pi);
  if (c->bn_actor != 0 && c->bn_adapt)                  /* :415-420: the actions come from the TEST-phase copy */
    ora_mlp_forward(d->ann_testing, all_s, B, pi);
  for (int i = 0; i < B; ++i) memcpy(sa + i * (S + A) + S, pi + i * A, sizeof(double) * A);
  ora_mlp_forward(d->qnn, sa, B, NULL);
  for (int i = 0; i < B; ++i) dq[i] = -1.0f;            /* :424-428, NOT divided by B */
  ora_mlp_backward(d->qnn, dq, B, dsa);                 /* critic_backward :429 */
  for (int i = 0; i < B; ++i)
    for (int h = 0; h < A; ++h) {
      double diff = dsa[i * (S + A) + S + h];
      if (c->inverting_grad) {                          /* :432-450 */
        double output = pi[i * A + h];
        double mn = -1.0, mx = 1.0;
        if (diff < 0) diff *= (mx - output) / (mx - mn);
        else if (diff > 0) diff *= (output - mn) / (mx - mn);
      }
      dact[i * A + h] = diff;
    }
  memcpy(d->last_dact, dact, sizeof(double) * B * A);
  ora_mlp_backward(d->ann, dact, B, NULL);              /* actor_backward :455 */
  ora_mlp_update_fisher(d->ann, (double)B);             /* :456 */
  ora_mlp_regularize(d->ann);                           /* :457 */
  ora_mlp_apply_update(d->ann);                         /* :458 */
  ora_mlp_set_iter(d->ann, ora_mlp_iter(d->ann) + 1);   /* :459 */

  /* soft updates (DeepQNAg.hpp:462-463) */
  ora_mlp_soft_update(d->qnn_target, d->qnn, c->tau);
  ora_mlp_soft_update(d->ann_target, d->ann, c->tau);

  free(idx); free(valid_next); free(all_s); free(all_s2); free(all_a); free(sa); free(next_a);
  free(qt); free(pi); free(dq); free(dsa); free(dact);
  return loss;
}

void ora_ddpg_sync_testing(ora_ddpg *d) {
  int np = ora_mlp_nparams(d->ann);
  double *w = (double *)malloc(sizeof(double) * (np ? np : 1));
  ora_mlp_get_weights(d->ann, w);
  ora_mlp_set_weights(d->ann_testing, w);
  free(w);
}

/* develop() -> ewc_setup() on the actor, and on the critic when ewc_for_critic == 2
 * (DeepQNAg.hpp:250-261); with ewc_for_critic == 1 the critic's own inform() does it. The test
 * driver calls this at a development switch for both. */
void ora_ddpg_ewc_setup(ora_ddpg *d) {
  ora_mlp_ewc_setup(d->ann);
  ora_mlp_ewc_setup(d->qnn);
}

void ora_ddpg_last(const ora_ddpg *d, double *q_targets, double *actor_dact) {
  if (q_targets) memcpy(q_targets, d->last_qt, sizeof(double) * d->c.B);
  if (actor_dact) memcpy(actor_dact, d->last_dact, sizeof(double) * d->c.B * d->c.A);
}

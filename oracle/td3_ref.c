/*
 * oracle/td3_ref.c -- TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * fp64 restatement of the TD3 minibatch update, TD3Ag's DeepQNAg<NN>::end_episode
 * (agent/deepqn/include/TD3Ag.hpp:284-380): one call per stored transition (no fupd loop, no n-step
 * targets, no inverting gradients in this agent).  The object is an ora_ddpg created with td3 = 1
 * (ddpg_ref.c builds qnn2 / qnn_target2, TD3Ag.hpp:221-229).
 * No reference test pins this arithmetic: parity unpinned beyond the MLP known-answer tests.
 */
#include "ddpg_ref.h"
#include <stdlib.h>
#include <string.h>

double ora_td3_update(ora_ddpg *d, const int *idx_in, int *idx_out, const double *noise, int actor_step, double *loss2) {
  const ora_ddpg_cfg *c = &d->c;
  int S = c->S, A = c->A, B = c->B;
  int *idx = (int *)malloc(sizeof(int) * B);
  /* sample_transition (TD3Ag.hpp:271-276) */
  for (int i = 0; i < B; ++i) idx[i] = idx_in ? idx_in[i] : ora_uniform_index(&d->eng, (uint32_t)d->len);
  if (idx_out) memcpy(idx_out, idx, sizeof(int) * B);

  double *all_s = (double *)malloc(sizeof(double) * B * S);
  double *all_s2 = (double *)malloc(sizeof(double) * B * S);
  double *all_a = (double *)malloc(sizeof(double) * B * A);
  double *sa = (double *)malloc(sizeof(double) * B * (S + A));
  double *next_a = (double *)malloc(sizeof(double) * B * A);
  double *qt = (double *)malloc(sizeof(double) * B);
  double *qt2 = (double *)malloc(sizeof(double) * B);
  double *pi = (double *)malloc(sizeof(double) * B * A);
  double *dq = (double *)malloc(sizeof(double) * B);
  double *dsa = (double *)malloc(sizeof(double) * B * (S + A));
  double *dact = (double *)malloc(sizeof(double) * B * A);

  /* gather (:293-302) */
  for (int i = 0; i < B; ++i) {
    const double *p = rep_row(d, idx[i]);
    memcpy(all_s2 + i * S, p + S + 2 * A, sizeof(double) * S);
    memcpy(all_s + i * S, p, sizeof(double) * S);
    memcpy(all_a + i * A, p + S + A, sizeof(double) * A);
  }

  /* a' = clip(ann_target(s') + eps, -1, 1), eps ~ N(0, policy_noise) truncated to +-noise_clip (:304-310) */
  ora_mlp_forward(d->ann_target, all_s2, B, next_a);
  for (int i = 0; i < B * A; ++i) {
    double v = next_a[i] + noise[i];
    v = v > (double)-1.f ? v : (double)-1.f;     /* std::min(std::max(a + eps, -1), 1) */
    next_a[i] = v < (double)1.f ? v : (double)1.f;
  }
  for (int i = 0; i < B; ++i) {
    memcpy(sa + i * (S + A), all_s2 + i * S, sizeof(double) * S);
    memcpy(sa + i * (S + A) + S, next_a + i * A, sizeof(double) * A);
  }
  ora_mlp_forward(d->qnn_target, sa, B, qt);     /* :313 */
  ora_mlp_forward(d->qnn_target2, sa, B, qt2);   /* :314 */

  /* targets: r, or r + gamma * min(Q1', Q2') (:317-326) */
  for (int i = 0; i < B; ++i) {
    const double *p = rep_row(d, idx[i]);
    double r = p[2 * S + 2 * A];
    int goal = p[2 * S + 2 * A + 1] != 0.;
    if (goal) qt[i] = r;
    else qt[i] = r + c->gamma * (qt[i] < qt2[i] ? qt[i] : qt2[i]);
  }
  memcpy(d->last_qt, qt, sizeof(double) * B);

  /* both critics learn the same targets (:329-330) */
  for (int i = 0; i < B; ++i) {
    memcpy(sa + i * (S + A), all_s + i * S, sizeof(double) * S);
    memcpy(sa + i * (S + A) + S, all_a + i * A, sizeof(double) * A);
  }
  double loss = ora_mlp_learn_batch(d->qnn, sa, qt, B, 1);
  double l2 = ora_mlp_learn_batch(d->qnn2, sa, qt, B, 1);
  if (loss2) *loss2 = l2;

  /* delayed actor update through the FIRST critic, then all three soft updates (:333-376) */
  if (actor_step) {
    ora_mlp_zero_grad(d->qnn);
    ora_mlp_zero_grad(d->ann);
    ora_mlp_forward(d->ann, all_s, B, pi);
    if (c->bn_actor != 0 && c->bn_adapt)                  /* :339-344: the actions come from the TEST-phase copy */
      ora_mlp_forward(d->ann_testing, all_s, B, pi);
    for (int i = 0; i < B; ++i) memcpy(sa + i * (S + A) + S, pi + i * A, sizeof(double) * A);
    ora_mlp_forward(d->qnn, sa, B, NULL);
    for (int i = 0; i < B; ++i) dq[i] = -1.0f;            /* :351-352, not divided by B */
    ora_mlp_backward(d->qnn, dq, B, dsa);                 /* critic_backward :353 */
    for (int i = 0; i < B; ++i)
      for (int h = 0; h < A; ++h) dact[i * A + h] = dsa[i * (S + A) + S + h];   /* ShareDiff :358 */
    memcpy(d->last_dact, dact, sizeof(double) * B * A);
    ora_mlp_backward(d->ann, dact, B, NULL);              /* actor_backward :359 */
    ora_mlp_update_fisher(d->ann, (double)B);             /* :360 */
    ora_mlp_regularize(d->ann);                           /* :361 */
    ora_mlp_apply_update(d->ann);                         /* :362 */
    ora_mlp_set_iter(d->ann, ora_mlp_iter(d->ann) + 1);   /* :363 */
    ora_mlp_soft_update(d->qnn_target, d->qnn, c->tau);   /* :368-370 */
    ora_mlp_soft_update(d->qnn_target2, d->qnn2, c->tau);
    ora_mlp_soft_update(d->ann_target, d->ann, c->tau);
    ora_ddpg_sync_testing(d);                             /* :372-375: ann_testing follows inside the policy step */
  }

  free(idx); free(all_s); free(all_s2); free(all_a); free(sa); free(next_a);
  free(qt); free(qt2); free(pi); free(dq); free(dsa); free(dact);
  return loss;
}

/*
 * oracle/nfac_ref.c -- TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * fp64 restatement of NFAC(lambda)-V, OfflineCaclaAg::update_critic and ::end_episode
 * (agent/cacla/include/OfflineCaclaAg.hpp:242-401, 403-595), batch_norm == 0 branches only
 * (BatchNorm is outside this round's scope).
 * No reference test pins this arithmetic: parity unpinned beyond the MLP known-answer tests.
 */
#include "ora.h"
#include <stdlib.h>
#include <string.h>

struct ora_nfac {
  ora_nfac_cfg c;
  ora_mlp *ann, *vnn;
  double *fresh; int n_fresh, next_fresh;
};

ora_nfac *ora_nfac_create(const ora_nfac_cfg *c) {
  ora_nfac *d = (ora_nfac *)calloc(1, sizeof(ora_nfac));
  d->c = *c;
  int dims[ORA_MAXL + 1];
  /* OfflineCaclaAg.hpp:186: actor with loss layer, B=1 at construction */
  dims[0] = c->S;
  for (int i = 0; i < c->n_hidden_a; ++i) dims[i + 1] = c->hidden_a[i];
  dims[c->n_hidden_a + 1] = c->A;
  d->ann = ora_mlp_create(c->n_hidden_a + 1, dims, c->hidden_act, c->actor_out_act, c->alpha_a, c->momentum, -1.);
  /* OfflineCaclaAg.hpp:190: V critic, input = sensors only, decay_v = -1 */
  dims[0] = c->S;
  for (int i = 0; i < c->n_hidden_v; ++i) dims[i + 1] = c->hidden_v[i];
  dims[c->n_hidden_v + 1] = 1;
  d->vnn = ora_mlp_create(c->n_hidden_v + 1, dims, c->hidden_act, 0, c->alpha_v, c->momentum, -1.);
  if (c->ewc >= 0.) {
    ora_mlp_ewc_config(d->ann, c->ewc, c->ewc_decay, c->ewc_fisher_beta, c->ewc_best_method);
    if (c->ewc_for_critic >= 1)
      ora_mlp_ewc_config(d->vnn, c->ewc, c->ewc_decay, c->ewc_fisher_beta, c->ewc_best_method);
  }
  return d;
}

void ora_nfac_free(ora_nfac *d) {
  if (!d) return;
  ora_mlp_free(d->ann); ora_mlp_free(d->vnn); free(d->fresh); free(d);
}

void ora_nfac_set_fresh(ora_nfac *d, const double *w, int n) {
  const int np = ora_mlp_nparams(d->vnn);
  free(d->fresh);
  d->fresh = (double *)malloc(sizeof(double) * (size_t)np * (n > 0 ? n : 1));
  memcpy(d->fresh, w, sizeof(double) * (size_t)np * n);
  d->n_fresh = n; d->next_fresh = 0;
}

ora_mlp *ora_nfac_net(ora_nfac *d, int which) { return which == 0 ? d->ann : d->vnn; }

/* the "Simple computation for lambda return" backward scan (OfflineCaclaAg.hpp:328-340, 481-493) */
static void lambda_scan(int T, const double *deltas, const int *ends, int n_ends, double gl, double *diff) {
  int li = T - 1;
  double prev_delta = 0.;
  int index_ep = n_ends - 1;
  for (int it = 0; it < T; ++it) {
    if (index_ep >= 0 && ends[index_ep] - 1 == li) { prev_delta = 0.; index_ep--; }
    diff[li] = deltas[li] + prev_delta;
    prev_delta = gl * diff[li];
    --li;
  }
}

/* OfflineCaclaAg::update_critic (OfflineCaclaAg.hpp:242-401), batch_norm_critic == 0 */
static void update_critic(ora_nfac *d, int T, const double *s, const double *s2, const double *r,
                          const int *goal, const int *ends, int n_ends) {
  const ora_nfac_cfg *c = &d->c;
  if (T <= 0) return;
  double *nextV = (double *)malloc(sizeof(double) * T);
  double *V = (double *)malloc(sizeof(double) * T);
  double *v_target = (double *)malloc(sizeof(double) * T);
  double *deltas = (double *)malloc(sizeof(double) * T);
  double *diff = (double *)malloc(sizeof(double) * T);
  for (int fit = 0; fit < c->number_fitted_iteration; ++fit) {
    ora_mlp_forward(d->vnn, s2, T, nextV);                                   /* :266 */
    for (int li = 0; li < T; ++li) {                                         /* :268-278 */
      double target = r[li];
      if (!goal[li]) target += c->gamma * nextV[li];
      v_target[li] = target;
    }
    if (c->vnn_from_scratch && d->next_fresh < d->n_fresh) {                 /* :281-284: a new net, new solver, no EWC */
      const int np = ora_mlp_nparams(d->vnn);
      double *zero = (double *)calloc(np, sizeof(double));
      ora_mlp_set_weights(d->vnn, d->fresh + (size_t)np * d->next_fresh++);
      ora_mlp_set_adam(d->vnn, zero, zero, 0);
      free(zero);
    }
    if (c->lambda < 0.) {
      ora_mlp_learn_batch(d->vnn, s, v_target, T, c->stoch_iter_critic);     /* :285-286 */
    } else {
      ora_mlp_forward(d->vnn, s, T, V);                                      /* :315 */
      for (int li = 0; li < T; ++li) deltas[li] = v_target[li] - V[li];      /* :321-325 */
      lambda_scan(T, deltas, ends, n_ends, c->gamma * c->lambda, diff);      /* :328-340 */
      for (int li = 0; li < T; ++li) diff[li] = diff[li] + V[li];            /* :344-348 */
      ora_mlp_learn_batch(d->vnn, s, diff, T, c->stoch_iter_critic);         /* :352 */
    }
  }
  free(nextV); free(V); free(v_target); free(deltas); free(diff);
}

int ora_nfac_update(ora_nfac *d, int T, const double *s, const double *pure_a, const double *a,
                    const double *s2, const double *r, const int *goal,
                    const int *ends, int n_ends, double *deltas_out) {
  const ora_nfac_cfg *c = &d->c;
  int A = c->A;
  int n = 0;
  if (c->update_critic_first) update_critic(d, T, s, s2, r, goal, ends, n_ends);   /* :430-431 */
  if (T > 0) {
    double *nextV = (double *)malloc(sizeof(double) * T);
    double *mine = (double *)malloc(sizeof(double) * T);
    double *deltas = (double *)malloc(sizeof(double) * T);
    double *diff = (double *)malloc(sizeof(double) * T);
    double *actions = (double *)malloc(sizeof(double) * T * A);
    char *disable_back = (char *)calloc((size_t)T * A, 1);
    double *ac_out = (double *)malloc(sizeof(double) * T * A);
    double *ac_diff = (double *)malloc(sizeof(double) * T * A);
    ora_mlp_forward(d->vnn, s2, T, nextV);                                   /* :460 */
    ora_mlp_forward(d->vnn, s, T, mine);                                     /* :461 */
    for (int li = 0; li < T; ++li) {                                         /* :464-475 */
      double v_target = r[li];
      if (!goal[li]) v_target += c->gamma * nextV[li];
      deltas[li] = v_target - mine[li];
    }
    if (c->gae) {                                                            /* :477-502 */
      lambda_scan(T, deltas, ends, n_ends, c->gamma * c->lambda, diff);
      memcpy(deltas, diff, sizeof(double) * T);
    }
    if (deltas_out) memcpy(deltas_out, deltas, sizeof(double) * T);
    for (int li = 0; li < T; ++li) {                                         /* :507-522 */
      if (deltas[li] > 0.) { memcpy(actions + li * A, a + li * A, sizeof(double) * A); n++; }
      else if (c->update_delta_neg) memcpy(actions + li * A, pure_a + li * A, sizeof(double) * A);
      else {
        memcpy(actions + li * A, a + li * A, sizeof(double) * A);
        memset(disable_back + (size_t)li * A, 1, A);
      }
    }
    if (n > 0) {
      for (int sia = 0; sia < c->stoch_iter_actor; ++sia) {                  /* :527-572 */
        ora_mlp_forward(d->ann, s, T, ac_out);
        ora_mlp_zero_grad(d->ann);
        for (int i = 0; i < T * A; ++i) {
          if (disable_back[i]) ac_diff[i] = 0.;
          else {
            double x = actions[i] - ac_out[i];
            ac_diff[i] = -x;
            if (c->corrected_update_ac) ac_diff[i] = -x * c->corrected_update_ac_factor;   /* :561-563 */
            else if (c->inverting_gradient) {
              const double min_ = -1.0, max_ = 1.0;
              if (ac_diff[i] < 0) ac_diff[i] *= (max_ - ac_out[i]) / (max_ - min_);
              else if (ac_diff[i] > 0) ac_diff[i] *= (ac_out[i] - min_) / (max_ - min_);
            }
          }
        }
        ora_mlp_backward(d->ann, ac_diff, T, NULL);                          /* :566 */
        ora_mlp_update_fisher(d->ann, (double)n);                            /* :567 */
        ora_mlp_regularize(d->ann);                                          /* :568 */
        ora_mlp_apply_update(d->ann);                                        /* :569 */
        ora_mlp_set_iter(d->ann, ora_mlp_iter(d->ann) + 1);                  /* :570 */
      }
    }
    free(nextV); free(mine); free(deltas); free(diff); free(actions); free(disable_back);
    free(ac_out); free(ac_diff);
  }
  if (!c->update_critic_first) update_critic(d, T, s, s2, r, goal, ends, n_ends);  /* :585-587 */
  ora_mlp_ewc_decay_update(d->ann);                                          /* :593-594 */
  ora_mlp_ewc_decay_update(d->vnn);
  return n;
}

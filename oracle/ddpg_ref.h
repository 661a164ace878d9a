/* oracle/ddpg_ref.h -- TEST INFRASTRUCTURE ONLY (see ora.h): the agent object ddpg_ref.c and td3_ref.c share */
#ifndef ORA_DDPG_REF_H
#define ORA_DDPG_REF_H
#include "ora.h"
#include <stddef.h>

struct ora_ddpg {
  ora_ddpg_cfg c;
  ora_mlp *ann, *qnn, *ann_target, *qnn_target;
  ora_mlp *ann_testing;          /* TEST-phase copy of the actor (DeepQNAg.hpp:221-222) */
  ora_mlp *qnn2, *qnn_target2;   /* TD3Ag.hpp:459 (NULL for DDPG) */
  ora_mt19937 eng;
  /* ring replay */
  int len, head, row;            /* row = 2S + 2A + 2 : s, pure_a(unused here), a, s2, r, goal */
  double *rep;
  double *last_qt, *last_dact;
};

static double *rep_row(const ora_ddpg *d, int logical) {
  return d->rep + (size_t)((d->head + logical) % d->c.replay_cap) * d->row;
}
#endif

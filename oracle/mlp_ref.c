/*
 * oracle/mlp_ref.c -- TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * fp64 restatement of the Caffe-backed MLP wrapper, common/include/nn/MLP.hpp, and of the EWC part
 * of common/include/nn/DODevMLP.hpp.  The layer arithmetic itself lives in the forked Caffe
 * (github.com/matthieu637/caffe, unpinned, not in /root/reference): InnerProduct, ReLU
 * (negative_slope), TanH, EuclideanLoss and AdamSolver below restate the published BVLC Caffe 1.0
 * algorithms; the reference call sites are cited per function.
 *
 * Flat parameter order = net.learnable_params() order = [W1 (out x in, row-major), b1, W2, b2, ...]
 * (MLP.hpp:623-676; pinned by CaffeUtest.cpp:24-55).  Developmental gate weights (lr_mult 0 under
 * heuristic != 0, DODevMLP.hpp:378-401) are kept outside the flat vector as 0/1 masks, i.e. the
 * flat vector is the ignore_null_lr=true view (MLP.hpp:610-621).
 */
#include "ora.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORA_ROWQ 16
struct ora_mlp {
  int L;                       /* number of InnerProduct layers */
  int d[ORA_MAXL + 1];
  int hidden_act, out_act;
  int np;
  int offw[ORA_MAXL], offb[ORA_MAXL];
  double *w, *g, *m, *v;
  int iter;
  double lr, beta1, beta2, eps, decay;
  /* activations of the last forward: act[l] = output of layer l-1 (act[0] = gated input) */
  int B;
  double *act[ORA_MAXL + 1];
  double *in_gate, *out_gate;
  /* per-row gate factors (Bernoulli developmental gates): a queue of sets, one consumed by each forward; the set of the
   * last forward stays in force for its backward.  Test plumbing: the masks come from the caller. */
  /* BatchNorm (+ Scale) towers (MLP.hpp:53-79, 978-1021): rank r sits, in place, on act[r] = the input of InnerProduct r
   * (r < L) or the net's output (r == L, actor "very_all" only).  Parameters live in the flat vector in layer order:
   * [mean_sum C, var_sum C, factor 1] (lr_mult 0) then Scale's [gamma C, (beta C)]. */
  int bn_code, bn_rank[ORA_MAXL + 2], bn_c[ORA_MAXL + 2], bn_scale, bn_bias, bn_test;
  int offbn[ORA_MAXL + 2], offsc[ORA_MAXL + 2];
  double *xnorm[ORA_MAXL + 2], *bn_den[ORA_MAXL + 2];
  unsigned char *frozen;       /* lr_mult == 0: untouched by ApplyUpdate */
  int rq_n, rq_B[ORA_ROWQ];
  double *rq_in[ORA_ROWQ], *rq_out[ORA_ROWQ];
  double *row_in, *row_out;
  double *pre_out;             /* the net's output ahead of the output gate (the Developmental layer has a top blob of its own,
                                * DODevMLP.hpp:262-268: the last activation's backward sees the ungated values) */
  /* EWC (DODevMLP.hpp:803-822) */
  double ewc, ewc_decay, ewc_decay_mult, fisher_beta, fisher_nbr, best_score;
  int best_method, last_update_best;
  double *fisher, *prev_fisher, *best_param, *best_prev;
};

/* MLP::convertBN (MLP.hpp:53-79) + where the constructors place the towers: Tower() before every hidden InnerProduct
 * (rank i-1, :756-758), then rank H on the tower's top (:141, :220), rank H+1 on the actor's output for very_all (:238-240);
 * the critic's first_except_action normalises the states only, before the concat, and nothing else (:130-134). */
static void bn_decode(ora_mlp *n, int code, int critic_sensors) {
  int H = n->L - 1;
  n->bn_code = code;
  n->bn_bias = (code & 1) != 0;
  n->bn_scale = (code & 2) != 0;
  int b2 = (code & 4) != 0, b3 = (code & 8) != 0, b4 = (code & 16) != 0;
  for (int r = 0; r <= n->L; ++r) { n->bn_rank[r] = 0; n->bn_c[r] = n->d[r]; }
  if (b4 && !b2) { n->bn_rank[0] = 1; n->bn_c[0] = critic_sensors; }         /* first_except_action */
  else if (b4) { for (int r = 0; r <= H; ++r) n->bn_rank[r] = 1; if (critic_sensors == 0) n->bn_rank[n->L] = 1; }   /* very_all */
  else if (b2 && b3) { for (int r = 0; r <= H; ++r) n->bn_rank[r] = 1; }     /* all */
  else if (b2) n->bn_rank[0] = 1;                                            /* first */
  else if (b3) { for (int r = 0; r < H; ++r) n->bn_rank[r] = 1; }            /* all_except_last */
}

ora_mlp *ora_mlp_create_bn(int nlayers, const int *dims, int hidden_act, int out_act, double lr, int momentum, double decay,
                           int bn_code, int critic_sensors) {
  ora_mlp *n = (ora_mlp *)calloc(1, sizeof(ora_mlp));
  n->L = nlayers;
  for (int i = 0; i <= nlayers; ++i) n->d[i] = dims[i];
  n->hidden_act = hidden_act;
  n->out_act = out_act;
  bn_decode(n, bn_code, critic_sensors);
  int off = 0;
  for (int l = 0; l <= nlayers; ++l) {
    if (n->bn_rank[l]) {
      n->offbn[l] = off; off += 2 * n->bn_c[l] + 1;
      n->offsc[l] = off; if (n->bn_scale) off += n->bn_c[l] * (n->bn_bias ? 2 : 1);
    }
    if (l == nlayers) break;
    n->offw[l] = off; off += dims[l + 1] * dims[l];
    n->offb[l] = off; off += dims[l + 1];
  }
  n->np = off;
  n->w = (double *)calloc(off, sizeof(double));
  n->g = (double *)calloc(off, sizeof(double));
  n->m = (double *)calloc(off, sizeof(double));
  n->v = (double *)calloc(off, sizeof(double));
  n->frozen = (unsigned char *)calloc(off ? off : 1, 1);
  for (int l = 0; l <= nlayers; ++l)
    if (n->bn_rank[l]) {
      memset(n->frozen + n->offbn[l], 1, 2 * n->bn_c[l] + 1);               /* BatchNormLayer: lr_mult 0 on its three blobs */
      if (n->bn_scale) for (int c = 0; c < n->bn_c[l]; ++c) n->w[n->offsc[l] + c] = 1.;   /* ScaleLayer: filler 1, bias filler 0 */
    }
  n->lr = lr;
  /* MLP.hpp:160-173 */
  if (momentum == 0) { n->beta1 = 0.; n->beta2 = 0.999; }
  else if (momentum == 1) { n->beta1 = 0.9; n->beta2 = 0.999; }
  else if (momentum == 2) { n->beta1 = 0.; n->beta2 = 0.; }
  else { n->beta1 = 0.9; n->beta2 = 0.; }
  n->eps = 1e-8;               /* caffe SolverParameter.delta default */
  n->decay = decay;            /* MLP.hpp:174-175: set only if !(decay_v < 0) */
  n->ewc = -1.; n->ewc_decay = -1.; n->ewc_decay_mult = 1.;
  n->best_score = -DBL_MAX;
  return n;
}

ora_mlp *ora_mlp_create(int nlayers, const int *dims, int hidden_act, int out_act,
                        double lr, int momentum, double decay) {
  return ora_mlp_create_bn(nlayers, dims, hidden_act, out_act, lr, momentum, decay, 0, 0);
}

/* the phase of a copy (MLP.hpp:284-338): TEST => use_global_stats on every BatchNorm layer */
void ora_mlp_set_phase(ora_mlp *n, int test) { n->bn_test = test; }
int  ora_mlp_bn_layout(const ora_mlp *n, int *rank_on, int *offbn, int *offsc, int *chan) {
  for (int r = 0; r <= n->L; ++r) { rank_on[r] = n->bn_rank[r]; offbn[r] = n->offbn[r]; offsc[r] = n->offsc[r]; chan[r] = n->bn_c[r]; }
  return n->bn_scale | (n->bn_bias << 1);
}
void ora_mlp_layer_offsets(const ora_mlp *n, int *offw, int *offb) {
  for (int l = 0; l < n->L; ++l) { offw[l] = n->offw[l]; offb[l] = n->offb[l]; }
}

#define ORA_ROWQ_DROP(n) do { free((n)->row_in); free((n)->row_out); (n)->row_in = (n)->row_out = NULL; } while (0)
static void free_acts(ora_mlp *n) {
  for (int l = 0; l <= n->L; ++l) {
    free(n->act[l]); n->act[l] = NULL;
    free(n->xnorm[l]); n->xnorm[l] = NULL;
    free(n->bn_den[l]); n->bn_den[l] = NULL;
  }
}

void ora_mlp_free(ora_mlp *n) {
  if (!n) return;
  free_acts(n);
  free(n->w); free(n->g); free(n->m); free(n->v);
  free(n->in_gate); free(n->out_gate); free(n->frozen);
  ORA_ROWQ_DROP(n);
  free(n->pre_out);
  for (int i = 0; i < n->rq_n; ++i) { free(n->rq_in[i]); free(n->rq_out[i]); }
  free(n->fisher); free(n->prev_fisher); free(n->best_param); free(n->best_prev);
  free(n);
}

static double *dupv(const double *p, int n) {
  if (!p) return NULL;
  double *q = (double *)malloc(sizeof(double) * n);
  memcpy(q, p, sizeof(double) * n);
  return q;
}

/* MLP(const MLP&, copy_solver, phase) (MLP.hpp:284-349): same structure and weights */
ora_mlp *ora_mlp_clone(const ora_mlp *s) {
  ora_mlp *n = (ora_mlp *)malloc(sizeof(ora_mlp));
  *n = *s;
  n->w = dupv(s->w, s->np); n->g = dupv(s->g, s->np);
  n->m = dupv(s->m, s->np); n->v = dupv(s->v, s->np);
  n->in_gate = dupv(s->in_gate, s->d[0]);
  n->out_gate = dupv(s->out_gate, s->d[s->L]);
  n->fisher = dupv(s->fisher, s->np); n->prev_fisher = dupv(s->prev_fisher, s->np);
  n->best_param = dupv(s->best_param, s->np); n->best_prev = dupv(s->best_prev, s->np);
  for (int l = 0; l <= n->L; ++l) { n->act[l] = NULL; n->xnorm[l] = NULL; n->bn_den[l] = NULL; }
  if (s->frozen) { n->frozen = (unsigned char *)malloc(s->np ? s->np : 1); memcpy(n->frozen, s->frozen, s->np); }
  n->B = 0;
  n->rq_n = 0; n->row_in = n->row_out = NULL;
  n->pre_out = NULL;
  return n;
}

int  ora_mlp_nparams(const ora_mlp *n) { return n->np; }
void ora_mlp_set_weights(ora_mlp *n, const double *w) { memcpy(n->w, w, sizeof(double) * n->np); }
void ora_mlp_get_weights(const ora_mlp *n, double *w) { memcpy(w, n->w, sizeof(double) * n->np); }
void ora_mlp_get_grads(const ora_mlp *n, double *g) { memcpy(g, n->g, sizeof(double) * n->np); }
void ora_mlp_get_adam(const ora_mlp *n, double *m, double *v) {
  memcpy(m, n->m, sizeof(double) * n->np); memcpy(v, n->v, sizeof(double) * n->np);
}
void ora_mlp_set_adam(ora_mlp *n, const double *m, const double *v, int iter) {
  memcpy(n->m, m, sizeof(double) * n->np); memcpy(n->v, v, sizeof(double) * n->np); n->iter = iter;
}
int  ora_mlp_iter(const ora_mlp *n) { return n->iter; }
void ora_mlp_set_iter(ora_mlp *n, int it) { n->iter = it; }
void ora_mlp_set_lr(ora_mlp *n, double lr) { n->lr = lr; }

void ora_mlp_set_gates(ora_mlp *n, const double *in_gate, const double *out_gate) {
  free(n->in_gate); free(n->out_gate);
  n->in_gate = dupv(in_gate, n->d[0]);
  n->out_gate = dupv(out_gate, n->d[n->L]);
}

/* queue one set of per-row gate factors [B x d0] / [B x dL] (either may be NULL = ones) for a coming forward */
int ora_mlp_push_row_gates(ora_mlp *n, const double *in_rows, const double *out_rows, int B) {
  if (n->rq_n >= ORA_ROWQ) return -1;
  n->rq_B[n->rq_n] = B;
  n->rq_in[n->rq_n] = dupv(in_rows, B * n->d[0]);
  n->rq_out[n->rq_n] = dupv(out_rows, B * n->d[n->L]);
  n->rq_n++;
  return 0;
}
int ora_mlp_row_gates_pending(const ora_mlp *n) { return n->rq_n; }

/* Caffe ReLULayer::Forward_cpu: max(x,0) + slope*min(x,0); TanHLayer: tanh(x) */
static double act_fwd(int type, double x) {
  switch (type) {
    case 1: return (x > 0 ? x : 0.) + 0.01 * (x < 0 ? x : 0.);
    case 2: return tanh(x);
    case 3: return x > 0 ? x : 0.;
    default: return x;
  }
}
/* in-place layers: backward sees only the output y (MLP.hpp:799-803 top==bottom).
 * ReLU: dx = dy * ((y > 0) + slope * (y <= 0)); TanH: dx = dy * (1 - y*y) */
static double act_bwd(int type, double y, double dy) {
  switch (type) {
    case 1: return dy * ((y > 0 ? 1. : 0.) + 0.01 * (y <= 0 ? 1. : 0.));
    case 2: return dy * (1. - y * y);
    case 3: return dy * (y > 0 ? 1. : 0.);
    default: return dy;
  }
}

/* Caffe 1.0 BatchNormLayer::Forward_cpu on act[r] in place (eps 1e-5, moving_average_fraction 0.999), then ScaleLayer.
 * TRAIN: batch mean / biased variance; running sums: factor = factor * maf + 1, mean_sum = mean + maf * mean_sum,
 * var_sum = B/(B-1) var + maf * var_sum.  TEST (use_global_stats): mean = mean_sum / factor, var = var_sum / factor
 * (0 when factor == 0). */
#define ORA_BN_EPS 1e-5
#define ORA_BN_MAF 0.999
static void bn_forward(ora_mlp *n, int r, int B) {
  int D = n->d[r], Cn = n->bn_c[r];
  double *X = n->act[r];
  double *P = n->w + n->offbn[r];
  const double *gam = n->bn_scale ? n->w + n->offsc[r] : NULL;
  const double *bet = n->bn_scale && n->bn_bias ? n->w + n->offsc[r] + Cn : NULL;
  free(n->xnorm[r]); free(n->bn_den[r]);
  n->xnorm[r] = (double *)malloc(sizeof(double) * B * Cn);
  n->bn_den[r] = (double *)malloc(sizeof(double) * Cn);
  double sf = 0.;
  if (n->bn_test) sf = P[2 * Cn] == 0. ? 0. : 1. / P[2 * Cn];
  else P[2 * Cn] = P[2 * Cn] * ORA_BN_MAF + 1.;
  for (int c = 0; c < Cn; ++c) {
    double mean, var;
    if (n->bn_test) { mean = sf * P[c]; var = sf * P[Cn + c]; }
    else {
      mean = 0.;
      for (int i = 0; i < B; ++i) mean += X[i * D + c];
      mean /= (double)B;
      var = 0.;
      for (int i = 0; i < B; ++i) { double xc = X[i * D + c] - mean; var += xc * xc; }
      var /= (double)B;
      double bc = B > 1 ? (double)B / (double)(B - 1) : 1.;
      P[c] = mean + ORA_BN_MAF * P[c];
      P[Cn + c] = bc * var + ORA_BN_MAF * P[Cn + c];
    }
    double den = sqrt(var + ORA_BN_EPS);
    n->bn_den[r][c] = den;
    for (int i = 0; i < B; ++i) {
      double xn = (X[i * D + c] - mean) / den;
      n->xnorm[r][i * Cn + c] = xn;
      X[i * D + c] = gam ? xn * gam[c] + (bet ? bet[c] : 0.) : xn;
    }
  }
}

/* ScaleLayer::Backward_cpu (d gamma = sum dy * x_norm, d beta = sum dy, dx = dy gamma) then BatchNormLayer::Backward_cpu:
 * TRAIN dx = (dy - mean(dy) - mean(dy y) y) / sqrt(var + eps), TEST dx = dy / sqrt(var + eps).  d [B x D] in place. */
static void bn_backward(ora_mlp *n, int r, int B, double *d) {
  int D = n->d[r], Cn = n->bn_c[r];
  const double *xn = n->xnorm[r];
  for (int c = 0; c < Cn; ++c) {
    if (n->bn_scale) {
      double sg = 0., sb = 0.;
      for (int i = 0; i < B; ++i) { sg += d[i * D + c] * xn[i * Cn + c]; sb += d[i * D + c]; }
      n->g[n->offsc[r] + c] += sg;
      if (n->bn_bias) n->g[n->offsc[r] + Cn + c] += sb;
      double gam = n->w[n->offsc[r] + c];
      for (int i = 0; i < B; ++i) d[i * D + c] *= gam;
    }
    double den = n->bn_den[r][c];
    if (n->bn_test) {
      for (int i = 0; i < B; ++i) d[i * D + c] /= den;
    } else {
      double m1 = 0., m2 = 0.;
      for (int i = 0; i < B; ++i) { m1 += d[i * D + c]; m2 += d[i * D + c] * xn[i * Cn + c]; }
      m1 /= (double)B; m2 /= (double)B;
      for (int i = 0; i < B; ++i) d[i * D + c] = (d[i * D + c] - m1 - m2 * xn[i * Cn + c]) / den;
    }
  }
}

/* Net::Forward as used by computeOutBatch / computeOutVFBatch (MLP.hpp:531-579).
 * Layer l: InnerProduct y = x W^T + b (MLP.hpp:926-949) then in-place activation. */
void ora_mlp_forward(ora_mlp *n, const double *x, int B, double *out) {
  if (n->B != B) {
    free_acts(n);
    for (int l = 0; l <= n->L; ++l) n->act[l] = (double *)malloc(sizeof(double) * B * n->d[l]);
    n->B = B;
  }
  int d0 = n->d[0];
  ORA_ROWQ_DROP(n);
  if (n->rq_n > 0 && n->rq_B[0] == B) {
    n->row_in = n->rq_in[0]; n->row_out = n->rq_out[0];
    n->rq_n--;
    for (int i = 0; i < n->rq_n; ++i) { n->rq_B[i] = n->rq_B[i + 1]; n->rq_in[i] = n->rq_in[i + 1]; n->rq_out[i] = n->rq_out[i + 1]; }
  }
  for (int i = 0; i < B; ++i)
    for (int k = 0; k < d0; ++k)
      n->act[0][i * d0 + k] = x[i * d0 + k] * (n->row_in ? n->row_in[i * d0 + k] : n->in_gate ? n->in_gate[k] : 1.);
  for (int l = 0; l < n->L; ++l) {
    int K = n->d[l], N = n->d[l + 1];
    const double *W = n->w + n->offw[l], *b = n->w + n->offb[l];
    int type = (l == n->L - 1) ? n->out_act : n->hidden_act;
    /* in place on the blob the previous activation wrote (MLP.hpp:756-758): that activation's backward will see these values */
    if (n->bn_rank[l]) bn_forward(n, l, B);
    const double *X = n->act[l];
    double *Y = n->act[l + 1];
    for (int i = 0; i < B; ++i)
      for (int o = 0; o < N; ++o) {
        double s = 0.;
        for (int k = 0; k < K; ++k) s += X[i * K + k] * W[o * K + k];
        Y[i * N + o] = act_fwd(type, s + b[o]);
      }
  }
  int dl = n->d[n->L];
  if (n->bn_rank[n->L]) bn_forward(n, n->L, B);
  free(n->pre_out);
  n->pre_out = NULL;
  if (n->row_out || n->out_gate) {
    n->pre_out = dupv(n->act[n->L], B * dl);
    for (int i = 0; i < B; ++i)
      for (int o = 0; o < dl; ++o) n->act[n->L][i * dl + o] *= n->row_out ? n->row_out[i * dl + o] : n->out_gate[o];
  }
  if (out) memcpy(out, n->act[n->L], sizeof(double) * B * dl);
}

/* Net::BackwardFrom (MLP.hpp:1109-1125, DODevMLP.hpp:595-601) with force_backward (MLP.hpp:108,211):
 * InnerProduct backward accumulates dW += dY^T X, db += dY^T 1 and overwrites dX = dY W. */
void ora_mlp_backward(ora_mlp *n, const double *dout, int B, double *din) {
  int L = n->L;
  int dl = n->d[L];
  double *dy = (double *)malloc(sizeof(double) * B * dl);
  int type = n->out_act;
  double *dtop = NULL;
  if (n->bn_rank[L]) {
    /* the output gate sits behind the tower (DODevMLP.hpp:246-316: behind the net's last layer): its backward comes first */
    dtop = (double *)malloc(sizeof(double) * B * dl);
    for (int i = 0; i < B * dl; ++i)
      dtop[i] = dout[i] * (n->row_out ? n->row_out[i] : n->out_gate ? n->out_gate[i % dl] : 1.);
    bn_backward(n, L, B, dtop);
    dout = dtop;
  }
  for (int i = 0; i < B * dl; ++i) {
    double d = dout[i];
    double y = n->act[L][i];
    if (n->row_out || n->out_gate) {
      double gte = n->row_out ? n->row_out[i] : n->out_gate[i % dl];
      if (!n->bn_rank[L]) d *= gte;
      /* the activation derivative needs the pre-gate value (under a TRAIN-phase output tower d != 0 even where the row's
       * gate is 0: the tower's backward mixes the rows of a column) */
      if (n->pre_out) y = n->pre_out[i];
      else if (gte != 0.) y /= gte;
    }
    dy[i] = act_bwd(type, y, d);
  }
  free(dtop);
  for (int l = L - 1; l >= 0; --l) {
    int K = n->d[l], N = n->d[l + 1];
    const double *W = n->w + n->offw[l];
    double *gW = n->g + n->offw[l], *gb = n->g + n->offb[l];
    const double *X = n->act[l];
    for (int o = 0; o < N; ++o) {
      for (int k = 0; k < K; ++k) {
        double s = 0.;
        for (int i = 0; i < B; ++i) s += dy[i * N + o] * X[i * K + k];
        gW[o * K + k] += s;
      }
      double sb = 0.;
      for (int i = 0; i < B; ++i) sb += dy[i * N + o];
      gb[o] += sb;
    }
    double *dx = (double *)malloc(sizeof(double) * B * K);
    for (int i = 0; i < B; ++i)
      for (int k = 0; k < K; ++k) {
        double s = 0.;
        for (int o = 0; o < N; ++o) s += dy[i * N + o] * W[o * K + k];
        dx[i * K + k] = s;
      }
    if (n->bn_rank[l]) bn_backward(n, l, B, dx);
    if (l > 0) {
      for (int i = 0; i < B * K; ++i) dx[i] = act_bwd(n->hidden_act, X[i], dx[i]);
    } else {
      if (n->row_in || n->in_gate)
        for (int i = 0; i < B * K; ++i) dx[i] *= n->row_in ? n->row_in[i] : n->in_gate[i % K];
      if (din) memcpy(din, dx, sizeof(double) * B * K);
    }
    free(dy);
    dy = dx;
  }
  free(dy);
}

/* MLP::ZeroGradParameters (MLP.hpp:1086-1107) / Net::ClearParamDiffs */
void ora_mlp_zero_grad(ora_mlp *n) { memset(n->g, 0, sizeof(double) * n->np); }

/* Solver::ApplyUpdate as called at DeepQNAg.hpp:458, OfflineCaclaAg.hpp:309,569, MLP.hpp:481:
 * SGDSolver::ApplyUpdate -> Regularize (L2: diff += weight_decay * data, only when weight_decay is
 * set, MLP.hpp:174-175) -> AdamSolver::ComputeUpdateValue (t = iter+1, correction =
 * sqrt(1-beta2^t)/(1-beta1^t), eps = delta = 1e-8 added to sqrt(v)) -> Net::Update (data -= diff).
 * The caller bumps iter (set_iter(iter()+1)). */
void ora_mlp_apply_update(ora_mlp *n) {
  int t = n->iter + 1;
  double correction = sqrt(1. - pow(n->beta2, t)) / (1. - pow(n->beta1, t));
  for (int i = 0; i < n->np; ++i) {
    if (n->frozen && n->frozen[i]) continue;          /* lr_mult 0: local_rate 0, the update value is 0 */
    double g = n->g[i];
    if (!(n->decay < 0) && n->decay != 0.) g += n->decay * n->w[i];
    n->m[i] = (1. - n->beta1) * g + n->beta1 * n->m[i];
    n->v[i] = (1. - n->beta2) * (g * g) + n->beta2 * n->v[i];
    double upd = n->lr * correction * (n->m[i] / (pow(n->v[i], 0.5) + n->eps));
    n->g[i] = upd;
    n->w[i] -= upd;
  }
}

/* MLP::learn_batch (MLP.hpp:468-485).  EuclideanLoss (MLP.hpp:950-957): loss = sum (yhat-y)^2 / (2B),
 * d yhat = (yhat - y) / B.  Non-EWC: Solver::Step(iters) = {ClearParamDiffs; ForwardBackward;
 * ApplyUpdate; ++iter}.  EWC: the explicit loop with updateFisher(B) and regularize(). */
double ora_mlp_learn_batch(ora_mlp *n, const double *x, const double *y, int B, int iters) {
  int dl = n->d[n->L];
  double *out = (double *)malloc(sizeof(double) * B * dl);
  double *dout = (double *)malloc(sizeof(double) * B * dl);
  double loss = 0.;
  for (int it = 0; it < iters; ++it) {
    ora_mlp_zero_grad(n);
    ora_mlp_forward(n, x, B, out);
    loss = 0.;
    for (int i = 0; i < B * dl; ++i) {
      double df = out[i] - y[i];
      loss += df * df;
      dout[i] = df / (double)B;
    }
    loss = loss / (double)B / 2.;
    ora_mlp_backward(n, dout, B, NULL);
    if (ora_mlp_ewc_active(n) >= 0) { /* ewc_enabled() */
      ora_mlp_update_fisher(n, (double)B);
      ora_mlp_regularize(n);
    }
    ora_mlp_apply_update(n);
    n->iter += 1;
  }
  free(out); free(dout);
  return loss;
}

/* MLP::learn_batch_lw (MLP.hpp:487-506): learn_batch with setWeightedSampleVector(&lw).  The fork's EuclideanWSLoss layer
 * (MLP.hpp:958-965) is not in the tree; restated as the per-sample weighted Euclidean loss, loss = sum w_i (yhat_i - y_i)^2 / (2B),
 * d yhat_i = w_i (yhat_i - y_i) / B, which is what the reference's own tests of it pin (CaffeUtest.cpp:153-244: the fit goes to
 * the weighted mean of the labels, zero-weight rows are ignored).  Beyond that behaviour: parity unpinned. */
double ora_mlp_learn_batch_lw(ora_mlp *n, const double *x, const double *y, const double *lw, int B, int iters) {
  int dl = n->d[n->L];
  double *out = (double *)malloc(sizeof(double) * B * dl);
  double *dout = (double *)malloc(sizeof(double) * B * dl);
  double loss = 0.;
  for (int it = 0; it < iters; ++it) {
    ora_mlp_zero_grad(n);
    ora_mlp_forward(n, x, B, out);
    loss = 0.;
    for (int i = 0; i < B * dl; ++i) {
      double df = out[i] - y[i];
      double w = lw[i / dl];
      loss += w * df * df;
      dout[i] = w * df / (double)B;
    }
    loss = loss / (double)B / 2.;
    ora_mlp_backward(n, dout, B, NULL);
    if (ora_mlp_ewc_active(n) >= 0) {
      ora_mlp_update_fisher(n, (double)B);
      ora_mlp_regularize(n);
    }
    ora_mlp_apply_update(n);
    n->iter += 1;
  }
  free(out); free(dout);
  return loss;
}

/* MLP::soft_update (MLP.hpp:437-452): to = tau*from + (1.f - tau)*to  (caffe_cpu_axpby) */
void ora_mlp_soft_update(ora_mlp *to, const ora_mlp *from, double tau) {
  double b = (1.f - tau);
  for (int i = 0; i < to->np; ++i) to->w[i] = tau * from->w[i] + b * to->w[i];
}

/* ---- EWC: DODevMLP.hpp:604-758 ---- */
void ora_mlp_ewc_config(ora_mlp *n, double ewc, double ewc_decay, double fisher_beta, int best_method) {
  n->ewc = ewc; n->ewc_decay = ewc_decay; n->fisher_beta = fisher_beta; n->best_method = best_method;
}

/* returns -1 if EWC disabled, 0 if enabled but no previous task yet, 1 if the penalty is live */
int ora_mlp_ewc_active(const ora_mlp *n) {
  if (!(n->ewc >= 0.)) return -1;
  return n->best_prev ? 1 : 0;
}

/* DODevMLP::updateFisher (DODevMLP.hpp:635-658) */
void ora_mlp_update_fisher(ora_mlp *n, double nsample) {
  if (!(n->ewc >= 0.)) return;
  if (!n->fisher) { n->fisher = (double *)calloc(n->np, sizeof(double)); n->fisher_nbr = 0.; }
  for (int k = 0; k < n->np; ++k)
    n->fisher[k] = n->fisher[k] * n->fisher_beta + (n->g[k] * n->g[k]) * nsample;
  n->fisher_nbr = n->fisher_nbr * n->fisher_beta + nsample;
}

/* DODevMLP::regularize (DODevMLP.hpp:686-707) */
void ora_mlp_regularize(ora_mlp *n) {
  if (!(n->ewc >= 0.) || !n->best_prev) return;
  double f = 1.;
  if (n->ewc_decay >= 0.) f = n->ewc_decay_mult;
  for (int k = 0; k < n->np; ++k)
    n->g[k] += n->ewc * f * n->prev_fisher[k] * (n->w[k] - n->best_prev[k]);
}

/* DODevMLP::ewc_cost (DODevMLP.hpp:660-684) */
double ora_mlp_ewc_cost(const ora_mlp *n) {
  if (!(n->ewc >= 0.) || !n->best_prev) return 0.;
  double cost = 0.;
  for (int k = 0; k < n->np; ++k) {
    double x = n->best_prev[k] - n->w[k];
    cost += x * x * n->prev_fisher[k];
  }
  /* k of DODevMLP.hpp:679 counts the blobs with lr_mult != 0 only: a tower's statistics are not among them (their Fisher is 0:
   * no gradient ever reaches them, so the sum above is unaffected) */
  int k = 0;
  for (int i = 0; i < n->np; ++i) k += !(n->frozen && n->frozen[i]);
  double r = cost * n->ewc / ((double)k);
  if (n->ewc_decay < 0.) return r;
  return r * n->ewc_decay_mult;
}

/* DODevMLP::ewc_setup (DODevMLP.hpp:604-633) */
void ora_mlp_ewc_setup(ora_mlp *n) {
  if (!(n->ewc >= 0.) || !n->best_param) return;
  n->ewc_decay_mult = 1.;
  n->best_score = -DBL_MAX;
  free(n->best_prev);
  n->best_prev = dupv(n->best_param, n->np);
  for (int i = 0; i < n->np; ++i)                      /* copyWeightsFrom(best_prev, true): ignore_null_lr, the statistics stay */
    if (!(n->frozen && n->frozen[i])) n->w[i] = n->best_prev[i];
  if (!n->fisher) { n->fisher = (double *)calloc(n->np, sizeof(double)); n->fisher_nbr = 0.; }
  for (int i = 0; i < n->np; ++i) n->fisher[i] = n->fisher[i] / n->fisher_nbr;
  double fmax = n->fisher[0];
  for (int i = 1; i < n->np; ++i) if (n->fisher[i] > fmax) fmax = n->fisher[i];
  for (int i = 0; i < n->np; ++i) n->fisher[i] = n->fisher[i] / fmax;
  free(n->prev_fisher);
  n->prev_fisher = n->fisher;
  n->fisher = NULL;
}

/* DODevMLP::ewc_decay_update (DODevMLP.hpp:709-712) */
void ora_mlp_ewc_decay_update(ora_mlp *n) {
  if (n->ewc >= 0. && n->ewc_decay >= 0.) n->ewc_decay_mult *= n->ewc_decay;
}

/* DODevMLP::update_best_param_previous_task (DODevMLP.hpp:722-758) */
void ora_mlp_update_best_param(ora_mlp *n, double score) {
  if (!(n->ewc >= 0.)) return;
  int go = 0, too_old;
  switch (n->best_method) {
    case 3: n->last_update_best++; go = n->last_update_best == 1; break;
    case 2:
      too_old = n->last_update_best > 50;
      n->last_update_best++;
      go = (score > n->best_score) || too_old;
      if (go) n->last_update_best = 0;
      break;
    case 4: case 0: go = score > n->best_score; break;
    case 1: go = 1; break;
  }
  if (go) {
    n->best_score = score;
    free(n->best_param);
    n->best_param = dupv(n->w, n->np);
  }
}

void ora_mlp_get_ewc(const ora_mlp *n, double *fisher, double *fisher_hat, double *wstar,
                     double *fisher_nbr, double *decay_mult) {
  if (fisher) { if (n->fisher) memcpy(fisher, n->fisher, sizeof(double) * n->np); else memset(fisher, 0, sizeof(double) * n->np); }
  if (fisher_hat) { if (n->prev_fisher) memcpy(fisher_hat, n->prev_fisher, sizeof(double) * n->np); else memset(fisher_hat, 0, sizeof(double) * n->np); }
  if (wstar) { if (n->best_prev) memcpy(wstar, n->best_prev, sizeof(double) * n->np); else memset(wstar, 0, sizeof(double) * n->np); }
  if (fisher_nbr) *fisher_nbr = n->fisher ? n->fisher_nbr : 0.;
  if (decay_mult) *decay_mult = n->ewc_decay_mult;
}

/* test hook: overwrite previous_fisher / best_param_previous_task (re-synchronisation to the
 * device's fp32 values in the teacher-forced parity tests) */
void ora_mlp_set_ewc(ora_mlp *n, const double *fisher_hat, const double *wstar) {
  if (!n->prev_fisher) n->prev_fisher = (double *)calloc(n->np, sizeof(double));
  if (!n->best_prev) n->best_prev = (double *)calloc(n->np, sizeof(double));
  memcpy(n->prev_fisher, fisher_hat, sizeof(double) * n->np);
  memcpy(n->best_prev, wstar, sizeof(double) * n->np);
}

/*
 * oracle/gen_cmaes_golden.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Golden-vector generator: links the reference's OWN c-cmaes, compiled from where it lies
 * (/root/reference/agent/cmaes/src/cmaes.c, see oracle/Makefile target `golden`), and dumps
 * its state after each generation as JSON (%.17g => round-trips doubles exactly).
 * Runs in the dev container only; the fixtures it writes (tests/golden/cmaes_*.json) are data.
 *
 * Determinism: explicit seed >= 1, parameter file "non", and
 * sp.updateCmode.maxtime = 1.0 so that the eigen-update schedule does not depend on CPU time.
 *
 * usage: gen_cmaes_golden <case> > tests/golden/cmaes_<case>.json
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "cmaes_interface.h"
double cmaes_random_Gauss(cmaes_random_t *); /* cmaes.c:2447, not in the public header */

static double f_sphere(const double *x, int n) { double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return s; }
static double f_elli(const double *x, int n) {
  double s = 0; for (int i = 0; i < n; ++i) s += pow(1e3, (double)i / (n - 1.)) * x[i] * x[i]; return s;
}
static double f_rosen(const double *x, int n) {
  double s = 0;
  for (int i = 0; i < n - 1; ++i) s += 100. * (x[i] * x[i] - x[i + 1]) * (x[i] * x[i] - x[i + 1]) + (x[i] - 1.) * (x[i] - 1.);
  return s;
}
/* CMAESAg.hpp:239 accumulates the episode scores in float: fitness rounded to fp32 => ties */
static double f_float_steps(const double *x, int n) { return (double)(float)floor(4. * f_sphere(x, n)); }

static void dump_vec(const char *name, const double *v, int n, int last) {
  printf("    \"%s\": [", name);
  for (int i = 0; i < n; ++i) printf("%s%.17g", i ? ", " : "", v[i]);
  printf("]%s\n", last ? "" : ",");
}
static void dump_ivec(const char *name, const int *v, int n, int last) {
  printf("    \"%s\": [", name);
  for (int i = 0; i < n; ++i) printf("%s%d", i ? ", " : "", v[i]);
  printf("]%s\n", last ? "" : ",");
}

static double f_const(const double *x, int n) { (void)x; (void)n; return 1.0; }

typedef double (*fit_t)(const double *, int);

/* names of the criteria cmaes_TestForTermination (cmaes.c:1536-1667) reports, '|'-separated */
static void dump_term(cmaes_t *evo) {
  static const char *names[] = {"Fitness:", "TolFun:", "TolFunHist:", "TolX:", "TolUpX:", "ConditionNumber:",
                                "NoEffectAxis:", "NoEffectCoordinate:", "MaxFunEvals:", "MaxIter:"};
  const char *msg = cmaes_TestForTermination(evo);
  printf("    \"term\": \"");
  int first = 1;
  if (msg)
    for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]); ++i)
      if (strstr(msg, names[i])) { printf("%s%.*s", first ? "" : "|", (int)strlen(names[i]) - 1, names[i]); first = 0; }
  printf("\",\n");
}

static void run_case(int N, int lambda, long seed, int gens, fit_t f, const char *fname, int full,
                     double x0scale, double sigma0, int term) {
  cmaes_t evo;
  double *x0 = (double *)malloc(sizeof(double) * N), *sd = (double *)malloc(sizeof(double) * N);
  for (int i = 0; i < N; ++i) { x0[i] = x0scale * (i + 1) / (double)N; sd[i] = sigma0; }
  double *fit = cmaes_init(&evo, N, x0, sd, seed, lambda, "non");
  evo.sp.updateCmode.maxtime = 1.0;
  printf("{\n  \"N\": %d, \"lambda\": %d, \"mu\": %d, \"seed\": %ld, \"fitness\": \"%s\", \"x0scale\": %.17g, \"sigma0\": %.17g,\n",
         N, evo.sp.lambda, evo.sp.mu, seed, fname, x0scale, sigma0);
  printf("  \"mueff\": %.17g, \"cs\": %.17g, \"damps\": %.17g, \"ccumcov\": %.17g, \"ccov\": %.17g, \"mucov\": %.17g, \"chiN\": %.17g, \"modulo\": %.17g,\n",
         evo.sp.mueff, evo.sp.cs, evo.sp.damps, evo.sp.ccumcov, evo.sp.ccov, evo.sp.mucov, evo.chiN, evo.sp.updateCmode.modulo);
  printf("  \"weights_head\": [%.17g, %.17g], \"gens\": [\n", evo.sp.weights[0], evo.sp.weights[evo.sp.mu - 1]);
  double *diagC = (double *)malloc(sizeof(double) * N);
  for (int g = 0; g < gens; ++g) {
    double *const *pop = cmaes_SamplePopulation(&evo);
    lambda = evo.sp.lambda;
    for (int k = 0; k < lambda; ++k) fit[k] = f(pop[k], N);
    printf("   {\n");
    printf("    \"eigen_uptodate_at_sample\": %d,\n", evo.flgEigensysIsUptodate);
    dump_vec("x_first", pop[0], N, 0);
    dump_vec("x_last", pop[lambda - 1], N, 0);
    if (full) {
      printf("    \"X\": [");
      for (int k = 0; k < lambda; ++k) for (int i = 0; i < N; ++i) printf("%s%.17g", (k || i) ? ", " : "", pop[k][i]);
      printf("],\n");
    }
    { /* cheap checksum of the whole population */
      double cs1 = 0, cs2 = 0;
      for (int k = 0; k < lambda; ++k) for (int i = 0; i < N; ++i) { cs1 += pop[k][i]; cs2 += pop[k][i] * pop[k][i] * (1 + (k % 7)); }
      printf("    \"X_sum\": %.17g, \"X_wsumsq\": %.17g,\n", cs1, cs2);
    }
    dump_vec("fitness", fit, lambda, 0);
    dump_vec("D_at_sample", evo.rgD, N, 0);
    cmaes_UpdateDistribution(&evo, fit);
    dump_ivec("index", evo.index, lambda, 0);
    printf("    \"sigma\": %.17g,\n", evo.sigma);
    dump_vec("xmean", evo.rgxmean, N, 0);
    dump_vec("ps", evo.rgps, N, 0);
    dump_vec("pc", evo.rgpc, N, 0);
    for (int i = 0; i < N; ++i) diagC[i] = evo.C[i][i];
    dump_vec("diagC", diagC, N, 0);
    if (full) {
      printf("    \"C_lower\": [");
      for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) printf("%s%.17g", (i || j) ? ", " : "", evo.C[i][j]);
      printf("],\n");
    }
    if (term) dump_term(&evo);
    { double cs = 0; for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) cs += evo.C[i][j] * (1 + ((i + 2 * j) % 5));
      printf("    \"C_wsum\": %.17g\n", cs); }
    printf("   }%s\n", g + 1 < gens ? "," : "");
  }
  printf("  ]\n}\n");
  cmaes_exit(&evo);
  free(x0); free(sd); free(diagC);
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <case>\n", argv[0]); return 2; }
  const char *c = argv[1];
  if (!strcmp(c, "n5")) run_case(5, 8, 1, 20, f_sphere, "sphere", 1, 0.5, 0.5, 0);
  else if (!strcmp(c, "n12_ties")) run_case(12, 32, 7, 12, f_float_steps, "float_steps", 1, 1.0, 0.3, 0);
  else if (!strcmp(c, "n40")) run_case(40, 64, 3, 30, f_rosen, "rosen", 0, 0.2, 0.3, 0);
  else if (!strcmp(c, "n151")) run_case(151, 50, 2, 12, f_elli, "elli", 0, 0.1, 0.5, 0);
  else if (!strcmp(c, "n151_big")) run_case(151, 4096, 5, 3, f_sphere, "sphere", 0, 0.1, 0.5, 0);
  else if (!strcmp(c, "n631_big")) run_case(631, 4096, 5, 3, f_sphere, "sphere", 0, 0.1, 0.5, 0);   /* the half-cheetah actor */
  else if (!strcmp(c, "term_flat")) run_case(4, 6, 2, 30, f_const, "const", 1, 0.5, 0.5, 1);
  else if (!strcmp(c, "term_noeffect")) run_case(4, 6, 4, 8, f_sphere, "sphere", 1, 144115188075855872.0 /* 2^57: means exactly representable in fp32 */, 1.0, 1);
  else if (!strcmp(c, "term_sphere")) run_case(5, 8, 1, 20, f_sphere, "sphere", 1, 0.5, 0.5, 1);
  else if (!strcmp(c, "kat")) {
    /* SURVEY.md 8(c'): the known-answer vector (x0 = .1...5, sigma .5, seed 1, lambda 8) + tie rule
     * + the head of the gaussian stream */
    cmaes_t evo; double x0[5] = {.1, .2, .3, .4, .5}, sd[5] = {.5, .5, .5, .5, .5};
    cmaes_init(&evo, 5, x0, sd, 1, 8, "non");
    printf("{\n  \"gauss_seed1_head\": [");
    for (int i = 0; i < 32; ++i) printf("%s%.17g", i ? ", " : "", cmaes_random_Gauss(&evo.rand));
    printf("],\n");
    cmaes_exit(&evo);
    printf("  \"tie_input\": [3, 1, 2, 1, 3, 1],\n");
    /* Sorted_index is static: reach it through UpdateDistribution on a 6-individual run */
    double x6[2] = {0., 0.}, s6[2] = {1., 1.};
    double *fit = cmaes_init(&evo, 2, x6, s6, 1, 6, "non");
    cmaes_SamplePopulation(&evo);
    double tie[6] = {3, 1, 2, 1, 3, 1};
    for (int i = 0; i < 6; ++i) fit[i] = tie[i];
    cmaes_UpdateDistribution(&evo, fit);
    printf("  \"tie_index\": [%d, %d, %d, %d, %d, %d]\n}\n", evo.index[0], evo.index[1], evo.index[2], evo.index[3], evo.index[4], evo.index[5]);
    cmaes_exit(&evo);
  } else { fprintf(stderr, "unknown case %s\n", c); return 2; }
  return 0;
}

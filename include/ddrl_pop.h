/*
 * ddrl_pop.h -- C ABI of libddrl_pop.so: the MI355X-native replacement of ddrl's learning hot path
 * (the Caffe-backed MLP wrapper + the DDPG / NFAC(lambda)-V / CMA-ES update arithmetic above it),
 * batched over a POPULATION of G independent agents.
 *
 * Plain C: opaque handles, pointers and sizes only.  Every function returns 0 on success and a
 * negative DDRL_E* code on failure; ddrl_last_error() returns the message of the calling thread's
 * last failure.  The library never calls exit() (the reference does: MLP.hpp:99-100, 352-355).
 * One handle = one GPU = one HIP stream; calls on a handle are asynchronous on that stream and
 * ordered; ddrl_pop_sync() waits.  A handle is thread-compatible, not thread-safe.
 *
 * Reference interface each entry point replaces is cited as path:line under /root/reference.
 *
 * Flat parameter order everywhere = the reference's learnable_params() order
 * [W1 (out x in, row-major), b1, W2, b2, ...] (common/include/nn/MLP.hpp:623-676, pinned by
 * common/src/test/CaffeUtest.cpp:24-55), ignore_null_lr=true view (gate weights excluded).
 * Host-side arrays are double (the reference's type, MLP.hpp:623,651); device state is fp32.
 */
#ifndef DDRL_POP_H
#define DDRL_POP_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDRL_MAX_HIDDEN 3

#define DDRL_OK            0
#define DDRL_EINVAL       -1   /* bad argument / unsupported configuration */
#define DDRL_EHIP         -2   /* HIP runtime error */
#define DDRL_ENOMEM       -3
#define DDRL_ESTATE       -4   /* call sequence error (e.g. update before enough samples) */
#define DDRL_EIO          -5   /* ini file problems */

/* nets of a DDPG population (DeepQNAg.hpp:209-234): ann, qnn, ann_target, qnn_target.
 * ann_testing (DeepQNAg.hpp:221,469-472) is a full copy of ann after every update: for nets without BatchNorm towers it is
 * ann itself (aliased); with towers it is a TEST-phase copy of its own, refreshed after every update round and after every
 * call that writes the actor. */
#define DDRL_NET_ACTOR          0
#define DDRL_NET_CRITIC         1
#define DDRL_NET_ACTOR_TARGET   2
#define DDRL_NET_CRITIC_TARGET  3
/* TD3 populations only (TD3Ag.hpp:221-229): qnn2, qnn_target2 */
#define DDRL_NET_CRITIC2         4
#define DDRL_NET_CRITIC2_TARGET  5

typedef struct ddrl_pop ddrl_pop_t;

/* The [agent] ini keys DeepQNAg::_unique_invoke reads (DeepQNAg.hpp:167-187) + ARLAgent's
 * (common/include/arch/ARLAgent.hpp:139-179), plus population-level fields.  Scalars here are the
 * population defaults; per-agent values (a sweep) are set with ddrl_pop_set_hyper(). */
typedef struct {
  int    n_agents;                 /* G: population size on this GPU */
  int    nb_sensors, nb_motors;    /* DeepQNAg ctor (DeepQNAg.hpp:95-98) */
  int    n_hidden_a; int hidden_unit_a[DDRL_MAX_HIDDEN];   /* agent.hidden_unit_a */
  int    n_hidden_q; int hidden_unit_q[DDRL_MAX_HIDDEN];   /* agent.hidden_unit_q */
  int    mini_batch_size;          /* agent.mini_batch_size (1..4096; up to 64 rows run as one fused launch) */
  int    replay_memory;            /* agent.replay_memory */
  int    inverting_grad;           /* agent.inverting_grad */
  int    force_more_update;        /* agent.force_more_update */
  int    hidden_layer_type;        /* 1 leaky-ReLU(0.01), 2 tanh, 3 ReLU (MLP.hpp:798-803) */
  int    actor_output_layer_type;  /* 0 linear, 1 leaky, 2 tanh, 3 ReLU (MLP.hpp:223-237) */
  int    momentum;                 /* 0..3 -> Adam betas (MLP.hpp:160-173) */
  int    nstep_return;             /* agent.nstep_return (DeepQNAg.hpp:360-373, 394-402) */
  int    batch_norm_actor, batch_norm_critic;  /* agent.batch_norm_actor / _critic: the bit-field of MLP.hpp:53-79 (bit 0 Scale bias,
                                                * bit 1 Scale, bits 2-4 where the towers sit); != 0: mini_batch_size <= 64,
                                                * widths <= 128 (kernels_ddpg_bn.hip); served for DDPG and TD3 handles; the developmental gates (both modes) and EWC are served: its vectors leave the towers'
                                                * running statistics out (lr_mult 0: DODevMLP.hpp:616, 641, 679) */
  double gamma;                    /* agent.gamma */
  double tau_soft_update, alpha_a, alpha_v, decay_v;
  /* [devnn] EWC keys (common/include/nn/DODevMLP.hpp:172-190); ewc < 0 disables */
  double ewc, ewc_decay, ewc_fisher_beta;
  int    ewc_best_param_method, ewc_for_critic;
  int    device;                   /* HIP device ordinal; -1 = current */
  /* TD3Ag (agent/deepqn/include/TD3Ag.hpp:183-186); td3 == 0: a DDPG population, the rest unused */
  int    td3;
  int    policy_freq;              /* agent.policy_freq: the actor and the targets move every policy_freq-th step */
  double policy_noise, noise_clip; /* agent.policy_noise, agent.noise_clip: target-policy smoothing */
  /* BatchNorm nets only (DeepQNAg.hpp:184-185, 216-234, 415-420) */
  int    test_net;                 /* agent.test_net: the target nets are TEST-phase copies (running statistics) */
  int    bn_adapt;                 /* agent.bn_adapt: the critic scores the actions of ann_testing, the TEST-phase actor copy */
} ddrl_ddpg_config;

const char *ddrl_last_error(void);
const char *ddrl_version(void);

/* fills *cfg with the defaults of BASELINE.md section 3 */
int ddrl_ddpg_config_default(ddrl_ddpg_config *cfg);
/* reads [agent] / [devnn] of a reference-format .ini (boost ini_parser syntax, list values a:b:c or
 * a,b,c as bib::to_array, common/include/bib/IniParser.hpp:11-24) into *cfg.  Replaces
 * DDPGAg_unique_invoke's config handling (agent/deepqn/src/pybinding/ddpg.cpp:17-45). Missing
 * mandatory keys are an error (the reference throws from ptree::get). */
int ddrl_ddpg_config_from_ini(const char *path, ddrl_ddpg_config *cfg);

/* Replaces DDPGAg_new + _unique_invoke (ddpg.cpp:10-45; DeepQNAg.hpp:167-235): allocates the four
 * nets of all G agents, Adam state, replay rings.  Weights are initialised ~N(0, 0.01^2) for W and
 * b (MLP.hpp:936-944) from init_seed (not bit-matched to Caffe's RNG: parity runs inject weights). */
int ddrl_ddpg_create(const ddrl_ddpg_config *cfg, uint64_t init_seed, ddrl_pop_t **out);
int ddrl_pop_destroy(ddrl_pop_t *h);
int ddrl_pop_sync(ddrl_pop_t *h);
/* what a handle holds: kind 0 DDPG, 1 NFAC, 2 policies only, 3 TD3; minibatch = 0 where there is none.
 * Any pointer may be NULL. */
int ddrl_pop_shape(ddrl_pop_t *h, int *kind, int *n_agents, int *nb_sensors, int *nb_motors, int *mini_batch_size);

/* per-agent hyper-parameters: key in {"alpha_a","alpha_v","tau_soft_update","gamma","decay_v"} */
int ddrl_pop_set_hyper(ddrl_pop_t *h, const char *key, const double *per_agent /* [G] */);

/* MLP::number_of_parameters / copyWeightsFrom / copyWeightsTo (MLP.hpp:610-676), all agents:
 * flat [G x n_params] doubles. */
int ddrl_pop_num_params(ddrl_pop_t *h, int net);
int ddrl_pop_set_weights(ddrl_pop_t *h, int net, const double *flat);
int ddrl_pop_get_weights(ddrl_pop_t *h, int net, double *flat);
/* Adam state of a trainable net (caffe SGDSolver history + Solver::iter, MLP.hpp:481-482,1127-1141) */
int ddrl_pop_get_adam(ddrl_pop_t *h, int net, double *m, double *v, int32_t *iter /* [G] */);
int ddrl_pop_set_adam(ddrl_pop_t *h, int net, const double *m, const double *v, const int32_t *iter);

/* replay sampling engine: one std::mt19937 per agent (bib/Seed.hpp:47-57), seeded explicitly */
int ddrl_pop_seed_replay(ddrl_pop_t *h, const uint32_t *seeds /* [G] */);

/* DeepQNAg::insertSample's deque push (DeepQNAg.hpp:159-163): one transition per agent with
 * mask[g] != 0 (mask NULL = all).  s,s2 [G x S], a [G x A], r [G], goal [G]. */
int ddrl_pop_replay_append(ddrl_pop_t *h, const float *s, const float *a, const float *r,
                           const float *s2, const uint8_t *goal, const uint8_t *mask);
/* bulk load of n transitions into agent g's ring (benchmarks, resume): rows [n x (2S+A+2)] =
 * s, a, s2, r, goal */
int ddrl_pop_replay_load(ddrl_pop_t *h, int agent, int n, const float *rows);
/* trajectory.clear() ("reset learning catched", DeepQNAg.hpp:254-257) */
int ddrl_pop_replay_clear(ddrl_pop_t *h, const uint8_t *mask);
int ddrl_pop_replay_len(ddrl_pop_t *h, int32_t *len /* [G] */);

/* MLP::computeOut on ann_testing (DeepQNAg.hpp:126; MLP.hpp:551-565) for all agents:
 * states [G x S] -> actions [G x A].  Host buffers. */
int ddrl_pop_infer(ddrl_pop_t *h, const float *states, float *actions);
/* same with device-resident buffers (no PCIe in the loop) */
int ddrl_pop_infer_dev(ddrl_pop_t *h, const float *d_states, float *d_actions);

/* MLP::computeOutBatch / computeOutVFBatch (MLP.hpp:531-579) for all agents: n rows per agent through
 * one net (any of the four; the critic's input rows are [s | a], an NFAC critic's are s).
 * x [G x n x in_dim] -> out [G x n x out_dim], host buffers.  No ownership games: the reference
 * returns a new std::vector the caller deletes (MLP.hpp:542,571). */
int ddrl_pop_forward_batch(ddrl_pop_t *h, int net, int n, const float *x, float *out);

/* One pass of the `fupd` loop body of DeepQNAg::end_episode (DeepQNAg.hpp:343-466) for every agent
 * with mask[g] != 0: sample B indices, gather, target forward passes, TD targets, critic Adam step,
 * deterministic policy gradient through the critic, inverting gradients, EWC terms, actor Adam
 * step, both soft target updates.  idx == NULL: indices are drawn on the device from each agent's
 * mt19937 with the libstdc++-11 uniform_int_distribution algorithm (bit-exact with
 * DeepQNAg.hpp:277).  idx != NULL: host array [G x B] of logical replay indices (0 = oldest).
 * The guard of DeepQNAg.hpp:338 (len >= 250 && len >= B) is applied per agent on the device. */
int ddrl_pop_ddpg_update(ddrl_pop_t *h, const int32_t *idx, const uint8_t *mask);
/* n consecutive passes of that loop body (the reference runs 1 + force_more_update of them per stored
 * transition, DeepQNAg.hpp:341; its ddpg.ini has force_more_update=7), each with a freshly sampled
 * minibatch, in ONE launch for one-tile minibatches: an agent's updates depend only on each other,
 * so the population is not forced through a grid-wide barrier between them.  Same replay-index
 * stream and same results as n calls of ddrl_pop_ddpg_update.  idx: NULL or [G x n x B].
 * 1 <= n <= 64. */
int ddrl_pop_ddpg_update_n(ddrl_pop_t *h, int n, const int32_t *idx, const uint8_t *mask);
/* ---- TD3 (the reference ships it as libddrl-td3: ddpg.cpp built with WANT_TD3) ----
 * ddrl_td3_config_from_ini reads the keys TD3Ag's _unique_invoke reads (TD3Ag.hpp:165-186: DeepQNAg's
 * without inverting_grad / force_more_update / nstep_return, plus policy_noise, noise_clip,
 * policy_freq) and sets td3 = 1; ddrl_td3_create = ddrl_ddpg_create on such a config: six nets per
 * agent (ann, qnn, qnn2 and their targets, TD3Ag.hpp:208-230).  Same limits as DDPG (mini_batch_size
 * <= 4096, layers <= 1024 wide): one fused launch for one-tile minibatches of nets up to 128 wide, the
 * chunk + apply kernels above 64 rows, the layer-by-layer kernels for wider nets.  Every other ddrl_pop_* entry point applies
 * (alpha_v / decay_v set both critics; the EWC calls cover qnn2 like qnn). */
int ddrl_td3_config_from_ini(const char *path, ddrl_ddpg_config *cfg);
int ddrl_td3_create(const ddrl_ddpg_config *cfg, uint64_t init_seed, ddrl_pop_t **out);
/* seeds of the target-policy smoothing streams, one per agent (default: derived from init_seed) */
int ddrl_pop_td3_seed_noise(ddrl_pop_t *h, const uint64_t *seeds /* [G] */);
/* One call of TD3Ag's end_episode (TD3Ag.hpp:284-380) for every agent with mask[g] != 0: sample B
 * indices, a' = clip(ann_target(s') + eps, -1, 1), y = r + gamma min(Q1', Q2'), both critics' Adam
 * steps, and for agents with actor_step[g] != 0 (current_step_counter % policy_freq == 0, :333; NULL
 * = all) the actor's step through the first critic and the three soft target updates.
 * idx as in ddrl_pop_ddpg_update.  noise: NULL = drawn on the device (N(0, policy_noise) truncated to
 * +-noise_clip, own stream: the reference's comes from its clock-seeded engine), else host array
 * [G x B x A] of the smoothing draws (parity tests). */
int ddrl_pop_td3_update(ddrl_pop_t *h, const int32_t *idx, const float *noise, const uint8_t *actor_step,
                        const uint8_t *mask);
/* diagnostics of the last TD3 update beyond ddrl_pop_ddpg_last (which applies): the second critic's
 * loss [G] and the smoothing noise used [G x B x A].  Either may be NULL. */
int ddrl_pop_td3_last(ddrl_pop_t *h, float *loss2, float *noise);

/* diagnostics of the last update: indices used [G x B], critic loss [G], TD targets [G x B],
 * action gradients fed to the actor [G x B x A].  Any pointer may be NULL. */
int ddrl_pop_ddpg_last(ddrl_pop_t *h, int32_t *idx, float *loss, float *q_targets, float *dact);

/* Developmental gates (DODevMLP.hpp:376-400 inserts the fork's "Developmental" layers; semantics as
 * far as the tree pins them: SURVEY.md 8h): per agent, a multiplicative gate per state dimension
 * (applied to the state input of every net) and per action dimension (applied to the actor's output
 * and to the critic's action input), shared by the agent's nets, not learned (heuristic != 0).
 * state_gate [G x S], action_gate [G x A]; either may be NULL (= ones); both NULL = gates off.  The
 * schedule (DODevMLP::inform, :420-470) stays with the caller.  Applies to DDPG and NFAC handles, nets with BatchNorm towers
 * included (the state gate ahead of an input tower, the action gate behind the actor's output tower). */
int ddrl_pop_set_gates(ddrl_pop_t *h, const double *state_gate, const double *action_gate);
/* the gates in force, [G x nb_sensors] / [G x nb_motors] (all ones when off); either may be NULL */
int ddrl_pop_get_gates(ddrl_pop_t *h, double *state_gate, double *action_gate);
/* The `probabilist` / `scale` parameters of the fork's Developmental layer (DODevMLP.hpp:276-304: st_probabilistic,
 * ac_probabilistic, st_scale, ac_scale).  *_bernoulli != 0 (the reference's probabilist = 0): per sample and gated
 * dimension the value passes with probability w (the gate weight), else 0; *_scale divides a passed value by w
 * (CaffeUtest.cpp:1126-1131).  *_bernoulli = 0 (default; probabilist = 1): the stored gate multiplies.  Every forward
 * pass of every layer instance draws its own mask, a backward pass re-creates its forward pass's mask (counter-based
 * generator keyed on seed / pass / agent / layer instance / row / dimension: the fork's generator is unknown, only the
 * pass rates are pinned).  Served by every DDPG / TD3 update path (one-tile, chunked, layer-by-layer), NFAC, infer and
 * forward_batch.  On nets with BatchNorm towers infer and forward_batch serve it; their UPDATE answers DDRL_EINVAL while it is
 * set (the per-dimension gates of ddrl_pop_set_gates are served there). */
int ddrl_pop_set_gate_mode(ddrl_pop_t *h, int st_bernoulli, int st_scale, int ac_bernoulli, int ac_scale, uint32_t seed);
/* DODevMLP's view of a net's parameters: after DODevMLP::exploit (DODevMLP.hpp:216-401) the gate weights are blobs of the
 * net, so number_of_parameters / copyWeightsTo / copyWeightsFrom (MLP.hpp:610-676) count and carry them:
 *   - an actor lists the state-gate weights first and the action-gate weights last (CaffeUtest.cpp:652-656, 964-965), a
 *     critic the action-gate weights, then the state-gate weights, then its InnerProducts (learnable_params()[1] is the
 *     state gate, :964; its copy of the flat vector holds the state gate at [len(ac_control) ..), :1001-1003);
 *   - a gate blob has one weight per controlled dimension, or per dimension when the control list is empty (+S / +A
 *     parameters, :629-636, :702-708), none when the list is "None" (n = -1 here);
 *   - heuristic 0 (the reference's default, DODevMLP.hpp:84: `learnable` != 0) leaves the gate weights lr_mult 1, so they
 *     are part of the view with ignore_null_lr as well; heuristic 1 / 2 freeze them (lr_mult 0, :381-385, :395-399): they
 *     are counted only with ignore_null_lr = 0 (CaffeUtest.cpp:1313-1314).
 * ddrl_pop_set_gate_layout records the control lists (an agent's nets share ONE set of gate weights, as the reference's
 * actor and critic share the blobs, :343-355; the targets here share them too) and, when no gates are set yet, installs
 * all-ones gates.  What heuristic 0 does NOT do here: the fork's Developmental layer backward for its own weights is not
 * in the tree, so the update kernels leave gate weights where the caller put them (parity unpinned; the forward, the
 * counts and the order are the reference's). */
int ddrl_pop_set_gate_layout(ddrl_pop_t *h, const int *st_control, int n_st, const int *ac_control, int n_ac, int learnable);
int ddrl_pop_dev_num_params(ddrl_pop_t *h, int net, int ignore_null_lr, int *n);
int ddrl_pop_dev_get_weights(ddrl_pop_t *h, int net, int ignore_null_lr, double *flat);        /* [G x n] */
int ddrl_pop_dev_set_weights(ddrl_pop_t *h, int net, int ignore_null_lr, const double *flat);  /* [G x n] */

/* EWC life-cycle (DODevMLP.hpp:604-758), per agent with mask[g] != 0 */
int ddrl_pop_update_best_param(ddrl_pop_t *h, const double *score /* [G] */, const uint8_t *mask);
int ddrl_pop_ewc_setup(ddrl_pop_t *h, const uint8_t *mask);
/* the same for one net (DDRL_NET_ACTOR / DDRL_NET_CRITIC): the agents call ann->ewc_setup() at a
 * development switch and qnn->ewc_setup() only with ewc_for_critic = 2 (DeepQNAg.hpp:253-261) */
int ddrl_pop_ewc_setup_net(ddrl_pop_t *h, int net, const uint8_t *mask);
int ddrl_pop_ewc_decay_update(ddrl_pop_t *h, const uint8_t *mask);
int ddrl_pop_ewc_cost(ddrl_pop_t *h, int net, double *cost /* [G] */);
/* fisher, fisher_hat (previous_fisher), wstar (best_param_previous_task): [G x n_params] */
int ddrl_pop_get_ewc(ddrl_pop_t *h, int net, double *fisher, double *fisher_hat, double *wstar,
                     double *fisher_nbr /* [G] */, double *decay_mult /* [G] */);

/* ------------------------------------------------------------------------------------------------
 * Agent-level front end over a DDPG population: ARLAgent::runf / start_episode
 * (common/include/arch/ARLAgent.hpp:37-137: decision_each, reward max-pooling, reward_scale,
 * discounted-return bookkeeping) around DeepQNAg::_run / insertSample / end_episode / end_instance
 * (agent/deepqn/include/DeepQNAg.hpp:122-165, 290-339), for G agents per call.  Exploration noise
 * (agent.gaussian_policy 0 e-greedy / 1 truncated Gaussian / 2 Ornstein-Uhlenbeck, agent.noise*)
 * runs on the host per agent (same distributions as the reference; its clock-seeded stream is not
 * reproducible).  Every array is per agent; mask NULL = all agents.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ddrl_agents ddrl_agents_t;
/* reads [agent] (+ [devnn]) of a reference .ini: the ddrl_ddpg_config keys plus agent.noise,
 * gaussian_policy, decision_each, reward_scale, noise2, noise3 */
int ddrl_agents_create_ddpg(const char *ini_path, int n_agents, int nb_sensors, int nb_motors, int device,
                            uint64_t seed, ddrl_agents_t **out);
/* same front end over an NFAC(lambda)-V population: OfflineCaclaAg::_run (OfflineCaclaAg.hpp:79-119:
 * trajectory push, gaussian_policy 1-4 or e-greedy) and end_episode / end_instance (:403-600:
 * best-param bookkeeping, trajectory_end_points, the update every agent.update_each_episode episodes).
 * max_trajectory <= 0: sized from environment.max_step_per_instance when the ini has it. */
int ddrl_agents_create_nfac(const char *ini_path, int n_agents, int nb_sensors, int nb_motors, int max_trajectory,
                            int device, uint64_t seed, ddrl_agents_t **out);
/* TD3Ag (agent/deepqn/include/TD3Ag.hpp:95-98, 165-231; start/run/end as DeepQNAg's): one update per
 * stored transition (:156-163), the actor and targets every policy_freq-th step of the episode (:333),
 * one more update at the end of a learning episode, end_instance counts the episode (:278-282). */
int ddrl_agents_create_td3(const char *ini_path, int n_agents, int nb_sensors, int nb_motors, int device,
                           uint64_t seed, ddrl_agents_t **out);
int ddrl_agents_destroy(ddrl_agents_t *h);
/* DODevMLP's developmental schedule: when the ini has a [devnn] section with st_control, the DDPG and
 * NFAC front ends read DODevMLP::exploit's keys (DODevMLP.hpp:76-214: st/ac_control, heuristic = 1 with
 * heuristic_devpoints, intrasec_motivation + im_smooth / im_window, reset_learning_algo,
 * informed_sensorimotor_space + pd_controller_factor; the reference's consistency checks are errors
 * here), start with every controlled dimension closed and, at each learning start_episode, run
 * inform(episode, last_sum_weighted_reward) per agent (:403-574): ewc_setup of the actor before a
 * switch, the new gates on the device, the replay cleared with reset_learning_algo, the critic's
 * ewc_setup with ewc_for_critic = 2 (DeepQNAg.hpp:250-272, OfflineCaclaAg.hpp:225-234); closed action
 * dimensions return the PD set point (neutral_action, :764-775).  heuristic = 2 (st/ac_probabilistic = 0, optional
 * st/ac_scale, heuristic_linearcoef): Bernoulli gates whose pass probability is episode x coefficient, re-set (after
 * ewc_setup) at every learning episode (:444-467).  heuristic 0 (learned gates) is refused. */
/* the population underneath (weights, Adam state, EWC life-cycle); owned by the front end */
ddrl_pop_t *ddrl_agents_population(ddrl_agents_t *h);
int ddrl_agents_start_episode(ddrl_agents_t *h, const float *states /* [G x S] */, const uint8_t *learning,
                              const uint8_t *mask);
/* one environment step: returns the action to apply [G x A].  On a decision step of a learning
 * agent: store the transition, run 1 + force_more_update update passes (DeepQNAg.hpp:159-165,341),
 * then perturb the policy's action. */
int ddrl_agents_run(ddrl_agents_t *h, const float *reward, const float *states, const uint8_t *learning,
                    const uint8_t *goal, const uint8_t *last, const uint8_t *mask, float *actions_out);
/* end_episode(learning) + end_instance(learning) as DDPGAg_end_episode does (ddpg.cpp:52-55) */
int ddrl_agents_end_episode(ddrl_agents_t *h, const uint8_t *learning, const uint8_t *mask);
int ddrl_agents_stats(ddrl_agents_t *h, double *sum_weighted_reward, double *sum_reward, int32_t *episode);

/* The reference's own C flavour for one agent, same names and argument types
 * (agent/deepqn/src/pybinding/ddpg.cpp:9-88; ctypes caller gym/agent.py:112-157): an existing
 * driver can load libddrl_pop.so in place of libddrl-ddpg.so.  Handles are int64 as gym/agent.py
 * declares them.  DDPGAg_unique_invoke understands "--config <ini>" (default config.ini) and, in
 * addition, "--seed <n>".  DDPGAg_run returns a buffer of nb_motors doubles valid until the next
 * call.  DDPGAg_delete has no counterpart (the reference's binding never frees the agent). */
int64_t DDPGAg_new(unsigned nb_motors, unsigned nb_sensors);
void DDPGAg_unique_invoke(int64_t ag, int argc, char **argv);
void DDPGAg_start_episode(int64_t ag, const double *state, bool learning);
const double *DDPGAg_run(int64_t ag, double reward, const double *sensors, bool learning, bool goal_reached, bool last);
void DDPGAg_end_episode(int64_t ag, bool learning);
void DDPGAg_dump(int64_t ag, bool learning, int episode, int step, double treward);
void DDPGAg_display(int64_t ag, bool learning, int episode, int step, double treward);
void DDPGAg_save(int64_t ag, int episode);
void DDPGAg_load(int64_t ag, int episode);
void DDPGAg_delete(int64_t ag);
/* and the NFAC(lambda)-V twin (agent/cacla/src/pybinding/nfac.cpp:17-97; gym/agent.py:17-62) */
int64_t OfflineCaclaAg_new(unsigned nb_motors, unsigned nb_sensors);
void OfflineCaclaAg_unique_invoke(int64_t ag, int argc, char **argv);
void OfflineCaclaAg_start_episode(int64_t ag, const double *state, bool learning);
const double *OfflineCaclaAg_run(int64_t ag, double reward, const double *sensors, bool learning, bool goal_reached, bool last);
void OfflineCaclaAg_end_episode(int64_t ag, bool learning);
void OfflineCaclaAg_dump(int64_t ag, bool learning, int episode, int step, double treward);
void OfflineCaclaAg_display(int64_t ag, bool learning, int episode, int step, double treward);
void OfflineCaclaAg_save(int64_t ag, int episode);
void OfflineCaclaAg_load(int64_t ag, int episode);
void OfflineCaclaAg_delete(int64_t ag);
/* TD3: the reference compiles ddpg.cpp with WANT_TD3 into libddrl-td3, which exports the DDPGAg_* names
 * over TD3Ag (ddpg.cpp:3-5; agent/deepqn/CMakeLists.txt:84-88; caller gym/agent.py:112-157).  One
 * library cannot export a name twice: the same calls under the TD3Ag_ prefix. */
int64_t TD3Ag_new(unsigned nb_motors, unsigned nb_sensors);
void TD3Ag_unique_invoke(int64_t ag, int argc, char **argv);
void TD3Ag_start_episode(int64_t ag, const double *state, bool learning);
const double *TD3Ag_run(int64_t ag, double reward, const double *sensors, bool learning, bool goal_reached, bool last);
void TD3Ag_end_episode(int64_t ag, bool learning);
void TD3Ag_dump(int64_t ag, bool learning, int episode, int step, double treward);
void TD3Ag_display(int64_t ag, bool learning, int episode, int step, double treward);
void TD3Ag_save(int64_t ag, int episode);
void TD3Ag_load(int64_t ag, int episode);
void TD3Ag_delete(int64_t ag);

/* ------------------------------------------------------------------------------------------------
 * NFAC(lambda)-V (agent/cacla/include/OfflineCaclaAg.hpp): population of G independent agents,
 * each with an actor and a V critic (no target nets).  An NFAC handle is a ddrl_pop_t: the weight /
 * Adam / EWC / inference / set_hyper("alpha_a","alpha_v","gamma") entry points above apply to it with
 * net = DDRL_NET_ACTOR or DDRL_NET_CRITIC (= vnn); the replay and ddpg entry points do not.
 * ---------------------------------------------------------------------------------------------- */
/* the [agent] keys OfflineCaclaAg::_unique_invoke reads (OfflineCaclaAg.hpp:124-179) + [devnn] */
typedef struct {
  int    n_agents, nb_sensors, nb_motors;
  int    n_hidden_a; int hidden_unit_a[DDRL_MAX_HIDDEN];
  int    n_hidden_v; int hidden_unit_v[DDRL_MAX_HIDDEN];
  int    update_delta_neg, update_critic_first;
  int    number_fitted_iteration, stoch_iter_actor, stoch_iter_critic;
  int    batch_norm_actor, batch_norm_critic;      /* must be 0 */
  int    actor_output_layer_type, hidden_layer_type, momentum;
  int    gae, inverting_gradient;
  int    max_trajectory;           /* capacity (transitions) of each agent's trajectory buffer */
  double gamma, lambda, alpha_a, alpha_v;
  double ewc, ewc_decay, ewc_fisher_beta;
  int    ewc_best_param_method, ewc_for_critic;
  int    device;
  /* agent.vnn_from_scratch (mandatory key, OfflineCaclaAg.hpp:129): every fitted iteration of update_critic starts
   * from a NEW critic -- weights ~N(0, 0.01^2), fresh Adam state, iter = 0, no EWC (:281-284); refused together
   * with EWC on the critic.  agent.corrected_update_ac (+ _factor, :164-173): the actor's output diff is
   * -(target - pi) * factor, without the inverting-gradient scaling (:550-563). */
  int    vnn_from_scratch, corrected_update_ac;
  double corrected_update_ac_factor;
} ddrl_nfac_config;

int ddrl_nfac_config_default(ddrl_nfac_config *cfg);
int ddrl_nfac_config_from_ini(const char *path, ddrl_nfac_config *cfg);
/* OfflineCaclaAg ctor + _unique_invoke (OfflineCaclaAg.hpp:59-62,122-209; C flavour
 * agent/cacla/src/pybinding/nfac.cpp:17-50) */
int ddrl_nfac_create(const ddrl_nfac_config *cfg, uint64_t init_seed, ddrl_pop_t **out);
/* trajectory.push_back({s, pure_a, a, next_s, r, goal}) (OfflineCaclaAg.hpp:84-85), one transition
 * per agent with mask[g] != 0 */
int ddrl_nfac_traj_append(ddrl_pop_t *h, const float *s, const float *pure_a, const float *a,
                          const float *s2, const float *r, const uint8_t *goal, const uint8_t *mask);
/* bulk form: replaces agent's trajectory by n rows [n x (2S+2A+2)] = s, pure_a, a, s2, r, goal and its
 * trajectory_end_points by ends[n_ends] */
int ddrl_nfac_traj_load(ddrl_pop_t *h, int agent, int n, const float *rows, int n_ends, const int32_t *ends);
/* trajectory_end_points.push_back(trajectory.size()) (OfflineCaclaAg.hpp:419) */
int ddrl_nfac_episode_end(ddrl_pop_t *h, const uint8_t *mask);
int ddrl_nfac_traj_len(ddrl_pop_t *h, int32_t *len /* [G] */);
/* capacity of every agent's trajectory buffer in transitions (> 0), or a negative DDRL_E* code */
int ddrl_nfac_max_trajectory(ddrl_pop_t *h);
/* the learning branch of OfflineCaclaAg::end_episode after the update_each_episode gate
 * (OfflineCaclaAg.hpp:423-594) for every agent with mask[g] != 0: critic fits (update_critic,
 * :242-401: TD(0) or lambda-return targets, number_fitted_iteration x stoch_iter_critic full-batch
 * Adam steps), advantages (optional GAE), CACLA target selection, stoch_iter_actor full-batch actor
 * steps, then trajectory.clear() and ewc_decay_update().  The batch is the whole trajectory
 * (increase_batchsize, MLP.hpp:403-435): any length up to max_trajectory. */
int ddrl_nfac_update(ddrl_pop_t *h, const uint8_t *mask);
/* MLP::learn_batch_lw (common/include/nn/MLP.hpp:487-506; learn_batch, :468-485, with lw == NULL) on the critic (vnn) of every
 * agent with mask[g] != 0: `iters` full-batch Adam steps on n rows x [G x n x S] -> y [G x n], each row's loss and
 * gradient scaled by its label weight lw [G x n]: loss = sum_i lw_i (v(x_i) - y_i)^2 / (2 n), d v_i = lw_i (v(x_i) - y_i) / n
 * (the fork's EuclideanWSLoss, MLP.hpp:958-965, is not in the tree: this is the form that reproduces the behaviour its tests pin,
 * CaffeUtest.cpp:153-244 -- the fit goes to the weighted mean of the labels, rows of weight 0 are ignored -- and is otherwise
 * parity unpinned).  EWC terms as in the reference's explicit loop when EWC is live.  The rows pass through the agents'
 * trajectory buffers: 1 <= n <= max_trajectory, and agents holding gathered transitions are refused (DDRL_ESTATE).
 * Not together with vnn_from_scratch. */
int ddrl_nfac_learn_batch_lw(ddrl_pop_t *h, const float *x, const float *y, const float *lw, int n, int iters, const uint8_t *mask);
/* diagnostics of the last update: advantages [G x max_trajectory] (row-padded), n = number of
 * positively advantaged transitions [G], last critic loss [G] */
int ddrl_nfac_last(ddrl_pop_t *h, float *deltas, int32_t *n_pos, float *critic_loss);


/* ------------------------------------------------------------------------------------------------
 * CMA-ES policy search (agent/cmaes): the vendored c-cmaes (agent/cmaes/src/cmaes.c) as driven by
 * CMAESAg (agent/cmaes/include/CMAESAg.hpp).  Population X [lambda x N], B, C live on the GPU in
 * fp32; sampling x = m + sigma * B (D o z) and the rank-mu covariance update are fp32-MFMA GEMMs,
 * ranking is an exact sort with c-cmaes' tie rule; the N-vectors (mean, ps, pc, D) and scalars are
 * fp64 on the device; the lazily scheduled eigendecomposition (cmaes.c:1885-1947) runs on the host
 * in fp64 by default (the reference's own tred2 / tql2 sequence: parity mode) or, option "device_eigen", on the GPU by
 * this library's one-sided Jacobi kernels.  z comes from the reference's own generator (cmaes.c:2421-2482, sequential by
 * construction) evaluated on the host by default, so runs are reproducible against c-cmaes seed for seed; option
 * "device_rng" draws it on the GPU instead (throughput mode: another stream).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ddrl_cmaes ddrl_cmaes_t;

/* cmaes_init(evo, N, xstart, stds, seed, lambda, "non") (cmaes.c:404-415; CMAESAg.hpp:97-98).
 * seed must be >= 1 (the reference's seed 0 means "seed from the clock", cmaes.c:2406-2410);
 * lambda < 2 selects the default 4 + floor(3 ln N) (cmaes.c:2795-2796). */
int ddrl_cmaes_create(int N, const double *xstart, const double *stds, long seed, int lambda, int device,
                      ddrl_cmaes_t **out);
int ddrl_cmaes_destroy(ddrl_cmaes_t *h);
/* cmaes_SamplePopulation (cmaes.c:612-659) */
int ddrl_cmaes_sample(ddrl_cmaes_t *h);
/* cmaes_ReSampleSingle (cmaes.c:685-708) */
int ddrl_cmaes_resample(ddrl_cmaes_t *h, int k);
/* CMAESAg::new_population's feasibility loop (CMAESAg.hpp:135-171): individuals with any
 * |x_i| >= bound are resampled one at a time in index order; *n_resampled (may be NULL) counts */
int ddrl_cmaes_make_feasible(ddrl_cmaes_t *h, double bound, int *n_resampled);
/* the sampled population, [lambda x N] row-major: host copy (double) / device pointer (fp32, row
 * stride *ld floats) for loading individuals into a policy population without leaving the GPU */
int ddrl_cmaes_get_population(ddrl_cmaes_t *h, double *X);
int ddrl_cmaes_population_dev(ddrl_cmaes_t *h, const float **d_X, int *ld);
/* cmaes_UpdateDistribution (cmaes.c:797-958) + Adapt_C2 (cmaes.c:963-1001).  fitness [lambda] is
 * double like c-cmaes' rgFunVal (CMAESAg.hpp:239,354) and is ranked in fp64.  _dev: device pointer
 * (e.g. the result of the RCCL fitness all-gather). */
int ddrl_cmaes_update(ddrl_cmaes_t *h, const double *fitness);
int ddrl_cmaes_update_dev(ddrl_cmaes_t *h, const double *d_fitness);
/* Sorted_index (cmaes.c:2989-3000) on the GPU: ascending, equal values => later index first */
int ddrl_cmaes_rank(ddrl_cmaes_t *h, const double *fitness, int n, int32_t *index_out);
/* No counterpart in the reference: ddrl_cmaes_rank for callers that cannot allocate device memory (tight serving
 * loops, harnesses that forbid hipMalloc).  It allocates nothing but it does WAIT for its stream before returning, so
 * it is not for use inside a HIP graph capture.  Same ranking, same tie rule, no
 * handle: the device staging of the fitness values and of the index lives in a fixed 48-KiB array inside the
 * library (one per GPU, held until the process ends), so n <= 4096 (larger n: DDRL_EINVAL).  fitness [n] and
 * index_out [n] may be host or device pointers; `stream` is a hipStream_t (NULL = the default stream) of the
 * CURRENT device; the call returns after the result has landed.  Calls no hipMalloc / hipFree.  All calls on
 * one GPU share that GPU's array: calls that overlap on one GPU race, the caller serialises them. */
int ddrl_cmaes_rank_noalloc(const double *fitness, int n, int32_t *index_out, void *stream);
/* cmaes_Get (scalars: "sigma","gen","lambda","mu","N","mueff","cs","damps","ccumcov","ccov","mucov",
 * "chiN","modulo","eigen_uptodate","countevals","fbestever") and cmaes_GetPtr-like vectors ("xmean",
 * "xbestever" (CMAESAg::getBestSolution, CMAESAg.hpp:328-335),"ps","pc","D",
 * "weights" [mu],"diagC","C" [N x N],"B" [N x N]) */
int ddrl_cmaes_get(ddrl_cmaes_t *h, const char *key, double *out);
int ddrl_cmaes_get_vec(ddrl_cmaes_t *h, const char *key, double *out);
int ddrl_cmaes_get_index(ddrl_cmaes_t *h, int32_t *index /* [lambda] */);
/* cmaes_TestForTermination (cmaes.c:1536-1667) with c-cmaes' default stopping parameters; CMAESAg
 * only logs it (CMAESAg.hpp:145-154).  msg receives the '|'-separated names of the criteria met
 * (TolFun, TolFunHist, TolX, TolUpX, ConditionNumber, NoEffectAxis, NoEffectCoordinate, MaxFunEvals,
 * MaxIter); returns their number, 0 = keep going, < 0 = error. */
int ddrl_cmaes_test_for_termination(ddrl_cmaes_t *h, char *msg, int msg_len);
/* options: "device_rng" (0/1) -- throughput mode: z ~ N(0, I) is drawn on the GPU from a counter-based
 * generator (Philox4x32-10) instead of c-cmaes' sequential LCG stream evaluated on the host
 * (cmaes.c:2421-2482); "device_eigen" (0/1): the eigendecomposition (cmaes.c:1885-1947, O(N^3) on
 * the host otherwise) by this library's one-sided Jacobi kernels in fp64 on the GPU; "eigen_block" (1/0, default 1): their rounds
 * over blocks of 8 rows (78 per sweep at N = 631) instead of single rows (630); "eigen_persist" (1/0, default 1; N <= 640): the
 * whole decomposition in one launch whose workgroups hand their rows to each other in memory (a bounded wait: if a workgroup is
 * not resident within 2 s -- "eigen_persist_wait_ms" -- the launch gives up, the decomposition is finished with one launch per
 * round, and the handle's "eigen_persist_timeouts" counter -- ddrl_cmaes_get -- says so; after two such launches the handle
 * stays with one launch per round) instead of one launch per round.  Same algorithm, different
 * stream / eigenbasis: no counterpart in the reference. */
int ddrl_cmaes_set_option(ddrl_cmaes_t *h, const char *key, int value);

/* CMAESAg (agent/cmaes/include/CMAESAg.hpp) around the two: a policies-only population -- G actor
 * nets without optimiser state, targets or replay, CMAESAg's `ann` (:81-82) once per individual --
 * and MLP::copyWeightsFrom (MLP.hpp:651-676, CMAESAg::loadPolicyParameters :324-326) from device
 * memory: row g of d_flat [G x ld] (fp32, learnable_params() order; ld == 0: one row for all) becomes
 * agent g's net, so a generation turns into lambda policies without leaving the GPU. */
int ddrl_policy_create(int n_agents, int nb_sensors, int nb_motors, const int *hidden, int n_hidden,
                       int hidden_layer_type, int output_layer_type, int device, uint64_t init_seed,
                       ddrl_pop_t **out);
int ddrl_pop_load_flat_dev(ddrl_pop_t *h, int net, const float *d_flat, int ld, const uint8_t *mask);
/* The ddrl_agents_* front end as CMAESAg: _unique_invoke (:50-133: [agent] hidden_unit_a,
 * actor_hidden_layer_type, actor_output_layer_type, batch_norm = 0, population, initial_deviation,
 * optional check_feasible / racing; cmaes_init at the net's initial weights + new_population),
 * ddrl_agents_start_episode = start_instance (:174-226: individual g, or the best-ever solution when
 * testing after the first update, goes into policy g), ddrl_agents_run = ARLAgent::runf + _run
 * (:38-47, no action noise), ddrl_agents_end_episode = end_episode + end_instance (:233-253:
 * arFunvals[g] = -sum_weighted_reward through the reference's float accumulator; once all lambda
 * individuals are in: cmaes_UpdateDistribution + new_population).  One agent slot per individual:
 * the generation is evaluated side by side; a mask with one bit set gives the reference's
 * one-individual-per-episode order.  One episode per instance, as in the reference's gym flavour. */
int ddrl_agents_create_cmaes(const char *ini_path, int nb_sensors, int nb_motors, int device, uint64_t seed,
                             ddrl_agents_t **out);
ddrl_cmaes_t *ddrl_agents_cmaes(ddrl_agents_t *h);            /* borrowed */
int ddrl_agents_cmaes_state(ddrl_agents_t *h, int *lambda, int *n_evaluated, int *generations, int *error_count);

/* ---- run state: save_run / load_previous_run of the reference (DeepQNAg.hpp:504-537:
 * the four nets, the trajectory deque, the episode counter; CMAESAg.hpp:279-308: scores, counters and
 * c-cmaes' "resume" file, cmaes.c:426-545; driven by --continue, Simulator.hpp:76-84,196-205) for a
 * whole population in ONE file, and complete enough that the continuation is bit-identical to an
 * uninterrupted run: weights, targets, Adam m / v / iter, per-agent hyper-parameters, EWC arrays and
 * bookkeeping, developmental gates, the replay rings with their sampling engines (DDPG / TD3), the
 * gathered trajectories (NFAC), the smoothing-noise streams (TD3); for CMA-ES the distribution (mean,
 * sigma, C, B, D, ps, pc), the current generation's samples, the random stream, the eigensystem
 * schedule and the termination history; for the agent front ends additionally every agent's episode /
 * step counters, returns, pending action and exploration-noise engine.
 * load_previous_run fills a handle CREATED WITH THE SAME SHAPES (algorithm, G, nets, minibatch, replay
 * capacity; lambda and N for CMA-ES): a file of another shape or a foreign file is refused with DDRL_EIO
 * before the handle is touched; a file that turns out truncated or damaged further in also fails with
 * DDRL_EIO, and the handle then holds a mix until a load succeeds.  Host byte order and layout: a file
 * resumes on the same kind of machine. */
int ddrl_pop_save_run(ddrl_pop_t *h, const char *path);
int ddrl_pop_load_previous_run(ddrl_pop_t *h, const char *path);
int ddrl_cmaes_save_run(ddrl_cmaes_t *h, const char *path);
int ddrl_cmaes_load_previous_run(ddrl_cmaes_t *h, const char *path);
/* the front end's file = its own state + its population's (+ the CMA-ES optimiser's) */
int ddrl_agents_save_run(ddrl_agents_t *h, const char *path);
int ddrl_agents_load_previous_run(ddrl_agents_t *h, const char *path);

/* what a front end drives: kind 0 DDPG, 1 NFAC, 2 CMA-ES, 3 TD3; any pointer may be NULL */
int ddrl_agents_shape(ddrl_agents_t *h, int *kind, int *n_agents, int *nb_sensors, int *nb_motors);

/* ---- batched environment front end: arch::Simulator (Simulator.hpp:76-248) for a
 * population.  One environment instance per agent, supplied as a table of C callbacks (the reference's
 * Environment interface: reset_episode, perceptions, performance, apply, running, final_state,
 * next_instance; common/include/arch/AEnvironment.hpp), stepped by a pool of n_threads host threads in
 * lock step; every tick is ONE ddrl_agents_run over [G x ...] batches.  Environments may finish at
 * different steps.  `env` is the environment / agent index; a given env is called from one thread at
 * a time, different envs concurrently.  dump (optional) writes the environment's log fields. */
typedef struct {
  void *user;
  void (*reset_episode)(void *user, int env, int learning);
  void (*perceptions)(void *user, int env, double *out /* [nb_sensors] */);
  double (*performance)(void *user, int env);
  void (*apply)(void *user, int env, const double *actuators /* [nb_motors] */);
  int (*running)(void *user, int env);
  int (*final_state)(void *user, int env);
  void (*next_instance)(void *user, int env, int learning);          /* may be NULL */
  void (*dump)(void *user, int env, char *buf, int buf_len);         /* may be NULL */
} ddrl_env_vtable;
typedef struct ddrl_sim ddrl_sim_t;
/* reads the [simulation] keys of the ini (max_episode, test_episode_per_episode, test_episode_at_end,
 * dump_log_each, display_log_each, save_agent_each; optional display_learning, continue_save_each;
 * Simulator.hpp:300-328).  Log lines go to <out_dir>/<agent>/0.learning.data and
 * <out_dir>/<agent>/0.<test episode>.testing.data in the reference's format (:236-245: episode, steps,
 * the agent's dump fields, the environment's), elapsed minutes to <out_dir>/time_elapsed (:127).
 * can_continue != 0 is --continue (:76-84, 196-205): every continue_save_each learning episodes the
 * run state goes to <out_dir>/continue.ddrl_pop (ddrl_agents_save_run) + continue.simu.data, and a
 * create that finds them resumes from there.  The agents handle is borrowed. */
int ddrl_sim_create(ddrl_agents_t *agents, const ddrl_env_vtable *env, const char *ini_path, const char *out_dir,
                    int n_threads, int can_continue, ddrl_sim_t **out);
int ddrl_sim_destroy(ddrl_sim_t *s);
/* Simulator::run (:86-128): the learning / testing episode schedule */
int ddrl_sim_run(ddrl_sim_t *s);
/* Simulator::run_episode (:139-208) for all G (agent, environment) pairs */
int ddrl_sim_run_episode(ddrl_sim_t *s, int learning, long lepisode, long tepisode);

/* Measurement hooks (no counterpart in the reference, which only records total wall minutes,
 * common/include/arch/Simulator.hpp:127): with profiling on, every ddrl_pop_ddpg_update() brackets
 * its update kernel with HIP events on the handle's stream; ddrl_pop_profile_read() synchronises and
 * returns the summed kernel time and the number of launches since the last read. */
int ddrl_pop_set_profiling(ddrl_pop_t *h, int on);
int ddrl_pop_profile_read(ddrl_pop_t *h, double *kernel_ms, int *launches);
/* diagnostic: enable != 0 makes the following updates record s_memtime at their phase boundaries;
 * out (may be NULL) receives the stamps of the last update, [G x 16] */
int ddrl_pop_debug_stamps(ddrl_pop_t *h, int enable, long long *out);
/* "force_generic" = 1: use the run-time-shape kernel even where a compile-time instantiation of the
 * update kernel matches the configured nets (both are held to the same parity tests) */
int ddrl_pop_set_option(ddrl_pop_t *h, const char *key, int value);

#ifdef __cplusplus
}
#endif
#endif

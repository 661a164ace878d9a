// f64_stubs.cpp -- only linked into the fp64 debug library (make f64): the paths that build leaves out (minibatches
// above 64 rows, layers wider than 128, BatchNorm towers, TD3, NFAC, the pair / resident kernels) answer "not supported".
#include <cstddef>
#define STUB_ERR(name) extern "C" int name() { return 801; /* hipErrorNotSupported */ }
#define STUB_ZERO(name) extern "C" std::size_t name() { return 0; }
STUB_ERR(ddpg_update_big_launch)
STUB_ERR(ddpg_update_wide_launch)
STUB_ERR(ddpg_update_pair_launch)
STUB_ERR(ddpg_update_resident_launch)
STUB_ERR(ddpg_update_loop8_launch)
STUB_ERR(td3_update_launch)
STUB_ERR(td3_update_big_launch)
STUB_ERR(td3_update_wide_launch)
STUB_ERR(td3_prep_launch)
STUB_ERR(ddpg_update_bn_launch)
STUB_ERR(td3_update_bn_launch)
STUB_ERR(net_forward_bn_launch)
STUB_ERR(net_forward_launch)
STUB_ERR(net_forward_wide_launch)
STUB_ZERO(ddpg_update_pair_ok)
STUB_ZERO(ddpg_update_loop8_ok)
STUB_ZERO(ddpg_update_pair_pays)
STUB_ZERO(ddpg_update_resident_ok)
STUB_ZERO(ddpg_update_bn_lds_bytes)
STUB_ZERO(ddpg_wide_block_floats)
STUB_ZERO(net_forward_wide_scratch_floats)
// the two C++-linkage hooks of nfac.hip that pop.hip calls (never reached: the debug library creates no NFAC handle)
struct NfacHost;
struct ddrl_pop;
struct Ckpt;
void ddrl_nfac_host_free(NfacHost *) {}
int ddrl_nfac_checkpoint(ddrl_pop *, Ckpt &) { return -4; }

// lqmpc_spec.hip -- register-resident specialisations (stub; filled in below in a later commit)
#include "lqmpc_common.h"
namespace lqmpc {
bool spec_available(int, int, int) { return false; }
bool launch_spec(const KParams &, hipStream_t, const char **) { return false; }
}  // namespace lqmpc

// KF6 for wide frames (32 < V <= 64 joints, the two-hand graph): stem_bf16_v6.hip compiled with the joint axis split into two
// halves (see the WIDE notes at the head of that file).  A translation unit of its own so that the wide instantiation and
// the headline kernel do not share a register-allocation context.
#define STGCN_V6_WIDE 1
#include "stem_bf16_v6.hip"

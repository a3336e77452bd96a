// SRU state-only scan over [u0 | forget] pairs, k = 2 instantiations (see sru_fwd.inl)
#define DMN_SRU_K 2
#define DMN_SRU_LAUNCH_NAME launch_sru_fwd_k2
#include "sru_fwd.inl"

// SRU forward scan, k = 3 instantiations (see sru_fwd.inl)
#define DMN_SRU_K 3
#define DMN_SRU_LAUNCH_NAME launch_sru_fwd_k3
#define DMN_SRU_LAUNCH_IMG_NAME launch_sru_fwd_img_k3
#include "sru_fwd.inl"

// SRU forward scan, k = 4 instantiations (see sru_fwd.inl)
#define DMN_SRU_K 4
#define DMN_SRU_LAUNCH_NAME launch_sru_fwd_k4
#define DMN_SRU_LAUNCH_IMG_NAME launch_sru_fwd_img_k4
#include "sru_fwd.inl"

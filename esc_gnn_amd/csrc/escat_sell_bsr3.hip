// escat_sell_bsr3.hip -- k_chain_sell instantiated for the BSR-3 operator (Q chain)
#include "escat_sell_kernel.h"

template <int DMAX, bool BWD>
static int launch_bsr3_sell_d(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                              float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.ent) {           // direct: no value pack (escat_sell.ent_of_slot)
        if (max_rows <= 256) return launch_sell_k<OpBsr3<DMAX, true>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
        return launch_sell_k<OpBsr3<DMAX, true>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    }
    if (max_rows <= 256) return launch_sell_k<OpBsr3<DMAX>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k<OpBsr3<DMAX>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}
template <bool BWD>
static int launch_bsr3_sell(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                            float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    switch (sd.dmax) {
        case 8:  return launch_bsr3_sell_d<8, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 12: return launch_bsr3_sell_d<12, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 16: return launch_bsr3_sell_d<16, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 20: return launch_bsr3_sell_d<20, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 22: return launch_bsr3_sell_d<22, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: BSR-3 packs must use 8, 12, 16, 20 or 22", sd.dmax);
}


int escat_launch_bsr3_sell(bool bwd, ESCAT_SELL_ARGS)
{
    return bwd ? launch_bsr3_sell<true>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st)
               : launch_bsr3_sell<false>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
}
#ifdef ESCAT_STAMPS
ESCAT_STAMP_SETTER(escat_debug_set_stamps_bsr3)
#endif

// escat_sell_csrv3.hip -- k_chain_sell instantiated for the per-channel CSR operator (factored vector operator)
#include "escat_sell_kernel.h"

template <int DMAX, bool BWD>
static int launch_csrv3_sell_d(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                               float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.planes == 4) {
        if (max_rows <= 256) return launch_sell_k<OpCsrV3<DMAX, true>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
        return launch_sell_k<OpCsrV3<DMAX, true>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    }
    if (max_rows <= 256) return launch_sell_k<OpCsrV3<DMAX>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k<OpCsrV3<DMAX>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}

int escat_launch_csrv3_sell(bool bwd, ESCAT_SELL_ARGS)
{
    switch (sd.dmax) {
#define CASE(d) case d: return bwd ? launch_csrv3_sell_d<d, true>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st) \
                                   : launch_csrv3_sell_d<d, false>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        CASE(8) CASE(16) CASE(20) CASE(24)
#undef CASE
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: per-channel CSR packs must use 8, 16, 20 or 24", sd.dmax);
}
#ifdef ESCAT_STAMPS
ESCAT_STAMP_SETTER(escat_debug_set_stamps_csrv3)
#endif

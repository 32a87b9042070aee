#define ESCAT_CSR_CA 1
#define ESCAT_CSR_CB 2
#define ESCAT_CSR_SUFFIX 12
#include "escat_sell_csr.inc"

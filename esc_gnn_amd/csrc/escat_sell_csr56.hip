#define ESCAT_CSR_CA 5
#define ESCAT_CSR_CB 6
#define ESCAT_CSR_SUFFIX 56
#include "escat_sell_csr.inc"

#define ESCAT_CSR_CA 7
#define ESCAT_CSR_CB 8
#define ESCAT_CSR_SUFFIX 78
#include "escat_sell_csr.inc"

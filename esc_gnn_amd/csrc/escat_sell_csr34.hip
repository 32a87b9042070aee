#define ESCAT_CSR_CA 3
#define ESCAT_CSR_CB 4
#define ESCAT_CSR_SUFFIX 34
#include "escat_sell_csr.inc"

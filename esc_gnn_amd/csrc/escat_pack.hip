// escat_pack.hip -- operator ingest on the GPU (placeholder until the pack kernels land)
#include "escat_dev.h"

// filter_sp_e0.hip — the k_filter instantiations of the SPECIAL launch for max_err = 0 (see filter_tu.hip.h)
#define SCT_FILTER_E 0
#define SCT_FILTER_SPECIAL 1
#include "filter_tu.hip.h"

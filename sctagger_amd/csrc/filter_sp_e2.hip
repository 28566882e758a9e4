// filter_sp_e2.hip — the k_filter instantiations of the SPECIAL launch for max_err = 2 (see filter_tu.hip.h)
#define SCT_FILTER_E 2
#define SCT_FILTER_SPECIAL 1
#include "filter_tu.hip.h"

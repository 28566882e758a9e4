// filter_sp_e1.hip — the k_filter instantiations of the SPECIAL launch for max_err = 1 (see filter_tu.hip.h)
#define SCT_FILTER_E 1
#define SCT_FILTER_SPECIAL 1
#include "filter_tu.hip.h"

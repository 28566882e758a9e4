// filter_e2.hip — the k_filter instantiations for max_err = 2 (see filter_tu.hip.h)
#define SCT_FILTER_E 2
#include "filter_tu.hip.h"

// filter_e0.hip — the k_filter instantiations for max_err = 0 (see filter_tu.hip.h)
#define SCT_FILTER_E 0
#include "filter_tu.hip.h"

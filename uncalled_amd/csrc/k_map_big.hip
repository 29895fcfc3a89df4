// Second translation unit of k_map.hip: the variant with the larger seed-cluster buffers (DevBig), see the note there.
#define UNC_BIG 1
#include "k_map.hip"

// tower.hip -- the device-only translation unit of the persistent tower launch (conv_tower.h).  sayuri_amd/_build.py
// compiles it to assembly, closes the layer loop in that assembly (tower_seam.py), assembles and links the result into
// a code object and embeds it in libsayuri_hip.so (tower_hsaco.inc), which loads it with hipModuleLoadData.
#include "conv_tower.h"

namespace sayuri {
template __global__ void conv_tower_kernel<4>(const TowerLayer*);
template __global__ void tower_se_fc_kernel<4>(const TowerLayer*);
template __global__ void conv_tower_kernel<2>(const TowerLayer*);
template __global__ void tower_se_fc_kernel<2>(const TowerLayer*);
}  // namespace sayuri

// placeholder until the scoring kernel lands (next commit)
#include "uvc_device.h"
extern "C" int uvc_launch_score(const RegionDev *, const UvcParams *, const UvcScoreRequest *, const UvcIndelAllele *, int32_t *, int64_t, int64_t *, hipStream_t) { return UVCGPU_EUNSUPPORTED; }

// cn_pose.hip -- multi_pose decode (models/decode.py:497-571).  Placeholder until
// the keypoint-matching kernel lands: reports CN_ERR_UNSUPPORTED (never silently
// computes on the CPU).
#include "cn_common.h"

extern "C" size_t cn_multi_pose_decode_workspace_bytes(int, int, int, int, int, int) { return 0; }

extern "C" int cn_multi_pose_decode_f32(const float *, const float *, const float *, const float *,
                                        const float *, const float *, int, int, int, int, int, int,
                                        int, float *, void *, size_t, void *)
{
    return CN_ERR_UNSUPPORTED;
}

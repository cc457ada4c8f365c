// explicit instantiation of the fused EKF-SLAM step kernel: n <= 43, 1 wavefronts per filter,
// 4 detections per group, 8 register pairs in flight per lane
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<43, 1, 4, 8, double>(const EkfStepParams&, hipStream_t);
}

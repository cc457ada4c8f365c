// explicit instantiation of the fused EKF-SLAM step kernel: n <= 103, 2 wavefronts per filter,
// 3 detections per group, 8 register pairs in flight per lane
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<103, 2, 3, 8, double>(const EkfStepParams&, hipStream_t);
}

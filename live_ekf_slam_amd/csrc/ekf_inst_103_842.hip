// explicit instantiation of the fused EKF-SLAM step kernel: n <= 103, 8 wavefronts per filter,
// 4 detections per group, 2 register pairs in flight per lane
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<103, 8, 4, 2, double>(const EkfStepParams&, hipStream_t);
}

// explicit instantiation of the fused EKF-SLAM step kernel: n <= 43, 1 wavefront per filter,
// 2 detections per group, 4 register pairs in flight per lane
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<43, 1, 2, 4, double>(const EkfStepParams&, hipStream_t);
}

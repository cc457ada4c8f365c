// explicit instantiation of the fused EKF-SLAM step kernel: n <= 43, 1 wavefront(s) per filter
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<43, 1>(const EkfStepParams&, hipStream_t);
}

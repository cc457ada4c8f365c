// explicit instantiation of the fused EKF-SLAM step kernel: n <= 203 (up to 100 landmarks), 4 wavefronts per filter
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<203, 4, 4, 4, double>(const EkfStepParams&, hipStream_t);
}
